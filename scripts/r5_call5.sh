#!/bin/bash
mkdir -p gpurun_out/r5
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label: /"; }
for i in 1 2; do
  (cd _prev && $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev: /')
  run "new default (full grid, U<=4, nt stores)" X=1
  run "full grid U<=4 plain stores" IIF_BN_PLAIN_STORES=1
  run "full grid U=1 nt" IIF_BN_UNROLL=1
  run "full grid U=2 nt" IIF_BN_UNROLL=2
  run "cap 4096 U<=4 nt" IIF_BN_GRID_CAP=4096
  run "cap 2048 U<=4 nt" IIF_BN_GRID_CAP=2048
  run "cap 4096 U=1 plain (= round 4 form)" IIF_BN_GRID_CAP=4096 IIF_BN_UNROLL=1 IIF_BN_PLAIN_STORES=1
  run "cap 1024 U<=4 nt" IIF_BN_GRID_CAP=1024
  run "producer sums everywhere" IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=1e12
done 2>&1 | tee gpurun_out/r5/ab_b.txt
