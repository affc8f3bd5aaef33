#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "algebra or reproducible or default_routes" > gpurun_out/r5/t_alg.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r5/t_alg.log | grep -B40 "short test summary" | cut -c1-220 | tail -60; tail -3 gpurun_out/r5/t_alg.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2 3; do
  run "shortcut on the standard passes (Gram behind the branch)" IIF_NO_DS_ALGEBRA=1
  run "shortcut by algebra (Gram behind the branch)" X=1
done 2>&1 | tee gpurun_out/r5/ab_g.txt
