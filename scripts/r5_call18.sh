#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2 3; do
  run "default" X=1
  run "small-output weight-gradient GEMMs: half the splits" IIF_WGRAD_SMALL_DIV=2
  run "a quarter" IIF_WGRAD_SMALL_DIV=4
  run "an eighth" IIF_WGRAD_SMALL_DIV=8
done 2>&1 | tee gpurun_out/r5/ab_k.txt
