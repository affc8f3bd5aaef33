#!/bin/bash
set -e
mkdir -p gpurun_out/r4
for a in "fwd 256 14 256 1024 1 1" "fwd 256 14 1024 256 1 1" "fwd 256 28 128 512 1 1" "fwd 256 28 512 128 1 1" "fwd 256 56 256 64 1 1" "dgrad 256 14 256 1024 1 1" "dgrad 256 14 1024 256 1 1" "fwd 256 7 512 2048 1 1"; do
  python scripts/conv_stamps.py $a 2>&1 | grep -v amdgpu
done > gpurun_out/r4/stamps1.log 2>&1
cat gpurun_out/r4/stamps1.log
python scripts/bm_stream1x1.py > gpurun_out/r4/bm1x1_a.log 2>&1; tail -30 gpurun_out/r4/bm1x1_a.log
