#!/bin/bash
# round 4, experiment 3: 1x1 weight-gradient fast path + uniform soffsets in the tile kernels
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r4/t_conv.log 2>&1 || { tail -30 gpurun_out/r4/t_conv.log; exit 1; }
tail -2 gpurun_out/r4/t_conv.log
IIF_WGRAD_NO_1X1=1 python scripts/bm_wgrad1x1.py --only 1x1 --check > gpurun_out/r4/wg3_old.log 2>&1; tail -1 gpurun_out/r4/wg3_old.log
python scripts/bm_wgrad1x1.py --only 1x1 --check > gpurun_out/r4/wg3_new.log 2>&1; tail -1 gpurun_out/r4/wg3_new.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export IIF_WGRAD_NO_1X1=1; else unset IIF_WGRAD_NO_1X1; fi
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events 2>gpurun_out/r4/bench_err.log | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/no_1x1=$v: /"
done
