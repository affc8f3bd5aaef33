#!/bin/bash
# isolated weight-gradient timings of every distinct ResNet50 shape at bs 256 (count per step in the comment)
while read cnt a; do echo -n "x$cnt "; python scripts/prof_conv.py wgrad $a 20 2>&1 | grep -v amdgpu; done <<'LIST'
1 256 56 64 64 1 1
3 256 56 64 64 3 1
4 256 56 64 256 1 1
2 256 56 256 64 1 1
1 256 56 256 128 1 1
1 256 56 128 128 3 2
4 256 28 128 512 1 1
1 256 56 256 512 1 2
3 256 28 512 128 1 1
3 256 28 128 128 3 1
1 256 28 512 256 1 1
1 256 28 256 256 3 2
6 256 14 256 1024 1 1
1 256 28 512 1024 1 2
5 256 14 1024 256 1 1
5 256 14 256 256 3 1
1 256 14 1024 512 1 1
1 256 14 512 512 3 2
3 256 7 512 2048 1 1
1 256 14 1024 2048 1 2
2 256 7 2048 512 1 1
2 256 7 512 512 3 1
LIST
