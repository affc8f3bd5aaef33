#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2 3; do
  run "default" X=1
  run "Gram half the splits" IIF_WGRAD_GRAM_DIV=2
  run "Gram a quarter" IIF_WGRAD_GRAM_DIV=4
  run "Gram an eighth" IIF_WGRAD_GRAM_DIV=8
  run "Gram a quarter and small P half" IIF_WGRAD_GRAM_DIV=4 IIF_WGRAD_SMALL_DIV=2
done 2>&1 | tee gpurun_out/r5/ab_l.txt
