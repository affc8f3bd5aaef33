#!/bin/bash
mkdir -p gpurun_out/r5
python scripts/dbg_ds_alg.py 1e30 2>&1 | grep -v amdgpu
python -m pytest tests/test_conv_gpu.py tests/test_resnet_gpu.py -x -q -m gpu -k "algebra or reproducible or default_routes or shortcut" > gpurun_out/r5/t_alg.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r5/t_alg.log | grep -B30 "short test summary" | cut -c1-220 | tail -45; tail -3 gpurun_out/r5/t_alg.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2; do
  run "shortcut on the standard passes" IIF_NO_DS_ALGEBRA=1
  run "shortcut by algebra" X=1
done 2>&1 | tee gpurun_out/r5/ab_h.txt
