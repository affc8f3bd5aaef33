#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
for i in 1 2 3; do
  (cd _prev && $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/start of round: /')
  IIF_AMD_LIB=$PWD/_ab/v3/libiif_amd.so $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/before pool bwd grid: /'
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/now: /'
done 2>&1 | tee gpurun_out/r5/ab_e.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests_c.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r5/gpu_tests_c.log
