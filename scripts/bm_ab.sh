#!/bin/bash
# 128- vs 256-pixel tiles on the MFMA-bound shapes (isolated)
for a in "fwd 256 14 256 256 3 1" "fwd 256 14 1024 256 1 1" "fwd 256 14 256 1024 1 1" "fwd 256 28 128 128 3 1" "fwd 256 28 512 128 1 1" "fwd 256 28 128 512 1 1" "fwd 256 7 512 512 3 1" "fwd 256 7 512 2048 1 1" "fwd 256 7 2048 512 1 1" "fwd 256 56 64 256 1 1" "fwd 256 56 256 128 1 1" "dgrad 256 14 256 256 3 1" "dgrad 256 28 512 128 1 1" "dgrad 256 56 256 64 1 1"; do
  for bm in 128 256; do echo -n "BM=$bm "; IIF_CONV_BM=$bm python scripts/prof_conv.py $a 20 2>&1 | grep -v amdgpu; done
done
