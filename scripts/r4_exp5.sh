#!/bin/bash
# round 4, experiment 5: BN3 algebra prep in one ticketed launch; pure-threshold re-tune
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "bn3 or masked" > gpurun_out/r4/t_bn3.log 2>&1 || { tail -40 gpurun_out/r4/t_bn3.log; exit 1; }
tail -2 gpurun_out/r4/t_bn3.log
python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "bn3 or reproducible" > gpurun_out/r4/t_bn3b.log 2>&1 || { tail -40 gpurun_out/r4/t_bn3b.log; exit 1; }
tail -2 gpurun_out/r4/t_bn3b.log
for v in default 1e30 0 default 1e30; do
  if [ $v = default ]; then unset IIF_BN3_ALGEBRA_PURE_MIN_ELEMS; else export IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=$v; fi
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events 2>gpurun_out/r4/bench_err.log | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/pure_min=$v: /"
done
