#!/bin/bash
# isolated timings of the bandwidth-bound convolution shapes of ResNet50 @ bs 256
for a in "fwd 256 56 64 256 1 1" "fwd 256 56 256 64 1 1" "fwd 256 28 128 512 1 1" "fwd 256 28 512 128 1 1" "fwd 256 56 64 64 3 1" \
         "dgrad 256 56 64 256 1 1" "dgrad 256 56 256 64 1 1" "dgrad 256 28 128 512 1 1" "dgrad 256 28 512 128 1 1" "dgrad 256 56 64 64 3 1" \
         "wgrad 256 56 64 256 1 1" "wgrad 256 56 256 64 1 1" "wgrad 256 56 64 64 3 1" "wgrad 256 14 256 256 3 1" "wgrad 256 14 256 1024 1 1" "wgrad 256 28 128 128 3 1"; do
  python scripts/prof_conv.py $a 20
done
