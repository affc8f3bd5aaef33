#!/bin/bash
# 128- vs 256-channel weight-gradient tiles (isolated)
for a in "wgrad 256 14 256 256 3 1" "wgrad 256 14 256 1024 1 1" "wgrad 256 14 1024 256 1 1" "wgrad 256 7 512 512 3 1" "wgrad 256 7 512 2048 1 1" "wgrad 256 7 2048 512 1 1" "wgrad 256 28 128 512 1 1" "wgrad 256 28 512 256 1 1" "wgrad 256 56 64 256 1 1" "wgrad 256 28 256 256 3 2"; do
  for bc in 128 256; do echo -n "BC=$bc "; IIF_WGRAD_BC=$bc python scripts/prof_conv.py $a 20 2>&1 | grep -v amdgpu; done
done
