#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2 3; do
  run "default" X=1
  run "dgrad2 on the 128-row tile" IIF_CONV_NO_BM256_2SRC=1
done 2>&1 | tee gpurun_out/r5/ab_c.txt
