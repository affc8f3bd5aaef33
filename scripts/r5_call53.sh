#!/bin/bash
mkdir -p gpurun_out/r5
F="--steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
run() { label=$1; shift; env "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s|^|$label: |"; }
for i in 1 2 3; do
  run "HEAD                                   " IIF_AMD_LIB=$PWD/_ab/v1/libiif_amd.so python bench.py $F
  run "one-vector BN-backward pass <= 64 VGPRs" X=1 python bench.py $F
done 2>&1 | tee gpurun_out/r5/ab_ah.txt
