#!/bin/bash
mkdir -p gpurun_out/r5
F="--steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
timeout -k 10 600 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "stem or pool" > gpurun_out/r5/gpu_tests_l.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5/gpu_tests_l.log
run() { label=$1; shift; env "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s|^|$label: |"; }
for i in 1 2 3; do
  run "pool backward + BN pass (two launches)" IIF_NO_POOL_BWD_FUSED=1 python bench.py $F
  run "gathered inside the BN pass           " X=1 python bench.py $F
done 2>&1 | tee gpurun_out/r5/ab_y.txt
