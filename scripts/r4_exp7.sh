#!/bin/bash
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r4/t_conv.log 2>&1 || { tail -40 gpurun_out/r4/t_conv.log; exit 1; }
tail -1 gpurun_out/r4/t_conv.log
python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "reproducible or fixture or fused" > gpurun_out/r4/t_res.log 2>&1 || { tail -40 gpurun_out/r4/t_res.log; exit 1; }
tail -1 gpurun_out/r4/t_res.log
python scripts/bm_stream1x1.py 2>&1 | grep -v amdgpu > gpurun_out/r4/bm1x1_b.log; cut -c1-52 gpurun_out/r4/bm1x1_b.log
bash scripts/ab_prev.sh 3
