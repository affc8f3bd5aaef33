#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s|^|$label: |"; }
for i in 1 2 3; do
  run "default" X=1
  run "gram4 small2" IIF_WGRAD_GRAM_DIV=4 IIF_WGRAD_SMALL_DIV=2
  run "gram8 small2" IIF_WGRAD_GRAM_DIV=8 IIF_WGRAD_SMALL_DIV=2
  run "gram4 small2 all2" IIF_WGRAD_GRAM_DIV=4 IIF_WGRAD_SMALL_DIV=2 IIF_WGRAD_ALL_DIV=2
  run "gram4 small2 halo2" IIF_WGRAD_GRAM_DIV=4 IIF_WGRAD_SMALL_DIV=2 IIF_WGRAD_HALO_DIV=2
  run "gram4 small2 all2 halo2" IIF_WGRAD_GRAM_DIV=4 IIF_WGRAD_SMALL_DIV=2 IIF_WGRAD_ALL_DIV=2 IIF_WGRAD_HALO_DIV=2
done 2>&1 | tee gpurun_out/r5/ab_m.txt
