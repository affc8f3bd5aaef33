#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s|^|$label: |"; }
for i in 1 2 3; do
  run "default" X=1
  run "sums from the producer everywhere" IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=1e12
  run "sums from P down to 28x28" IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=9e7
  run "BN passes with nt stores" IIF_BN_NT_STORES=1
  run "small P a quarter" IIF_WGRAD_SMALL_DIV=4
done 2>&1 | tee gpurun_out/r5/ab_n.txt
