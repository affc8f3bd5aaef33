#!/bin/bash
# round 4, experiment 2: ring depth of the tap-per-tile weight gradient
set -e
mkdir -p gpurun_out/r4
run() { name=$1; shift; env "$@" python scripts/bm_wgrad1x1.py > gpurun_out/r4/wg2_$name.log 2>&1; tail -1 gpurun_out/r4/wg2_$name.log; }
run n3 IIF_WGRAD_NST_256=3 IIF_WGRAD_NST_128=3 IIF_WGRAD_NST_64=3
run n4 IIF_WGRAD_NST_256=4 IIF_WGRAD_NST_128=3 IIF_WGRAD_NST_64=4
run n6 IIF_WGRAD_NST_256=6 IIF_WGRAD_NST_128=5 IIF_WGRAD_NST_64=6
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "wgrad" > gpurun_out/r4/t_wgrad.log 2>&1 || { tail -30 gpurun_out/r4/t_wgrad.log; exit 1; }
tail -2 gpurun_out/r4/t_wgrad.log
for v in "3 3 3" "6 3 3" "6 5 6" "3 3 3" "6 3 3" "6 5 6" "4 3 4"; do
  set -- $v
  IIF_WGRAD_NST_256=$1 IIF_WGRAD_NST_128=$2 IIF_WGRAD_NST_64=$3 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events 2>gpurun_out/r4/bench_err.log | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/nst $1 $2 $3: /"
done
