#!/bin/bash
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "bn3 or masked" > gpurun_out/r4/t_bn3.log 2>&1 || { tail -40 gpurun_out/r4/t_bn3.log; exit 1; }
tail -1 gpurun_out/r4/t_bn3.log
python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "bn3 or reproducible" > gpurun_out/r4/t_bn3b.log 2>&1 || { tail -40 gpurun_out/r4/t_bn3b.log; exit 1; }
tail -1 gpurun_out/r4/t_bn3b.log
bash scripts/ab_prev.sh 3
bash scripts/kt_quick.sh kt6 bn3
