#!/bin/bash
for a in "fwd 256 14 256 256 3 1" "fwd 256 28 128 128 3 1" "fwd 256 7 512 512 3 1" "dgrad 256 14 256 256 3 1" "dgrad 256 28 128 128 3 1" "dgrad 256 7 512 512 3 1"; do
  echo -n "taps128 "; IIF_CONV_NO_HALO=1 IIF_CONV_BM=128 python scripts/prof_conv.py $a 20 2>&1 | grep -v amdgpu
  echo -n "taps256 "; IIF_CONV_NO_HALO=1 IIF_CONV_BM=256 python scripts/prof_conv.py $a 20 2>&1 | grep -v amdgpu
  echo -n "halo    "; IIF_CONV_HALO_FORCE=1 python scripts/prof_conv.py $a 20 2>&1 | grep -v amdgpu
done
