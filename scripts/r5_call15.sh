#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests_d.log 2>&1; echo "gpu tests rc=$?"; grep -v amdgpu gpurun_out/r5/gpu_tests_d.log | tail -4 | cut -c1-250
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
for i in 1 2 3; do
  (cd _prev && $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/start of round: /')
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/now: /'
done 2>&1 | tee gpurun_out/r5/ab_i.txt
python scripts/dbg_route_curve.py default standard noalg 2>&1 | grep -v amdgpu | tee gpurun_out/r5/route_curves2.txt
