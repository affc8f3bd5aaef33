#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_conv_gpu.py tests/test_resnet_gpu.py -x -q -m gpu -k "algebra or reproducible or default_routes or shortcut" 2>&1 | tail -2
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
mkdir -p _ab
for i in 1 2 3; do
  IIF_AMD_LIB=$PWD/_ab/v4/libiif_amd.so $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/correction by wave sums: /'
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/correction in the bias loop: /'
done 2>&1 | tee gpurun_out/r5/ab_j.txt
