#!/bin/bash
mkdir -p gpurun_out/r5
{
python scripts/dbg_route_curve.py default standard noalg
IIF_AMD_LIB=$PWD/_ab/v1/libiif_amd.so python scripts/dbg_route_curve.py default
IIF_AMD_LIB=$PWD/_ab/v2/libiif_amd.so python scripts/dbg_route_curve.py default standard
(cd _prev && python ../scripts/dbg_route_curve.py default standard)
} 2>&1 | grep -v amdgpu | tee gpurun_out/r5/route_curves.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
for i in 1 2; do
  (cd _prev && $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev: /')
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new:  /'
  IIF_BN_GRID_CAP=1000000000 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new, bn grid uncapped:  /'
  IIF_BN_GRID_CAP=16384 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new, bn grid 16384:  /'
done 2>&1 | tee gpurun_out/r5/ab_a.txt
