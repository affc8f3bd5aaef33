#!/bin/bash
# nine-taps-per-block 3x3 weight gradient: tap-per-tile kernel (HALO=0) vs the halo-window kernel
for a in "wgrad 256 56 64 64 3 1" "wgrad 256 28 128 128 3 1" "wgrad 256 14 256 256 3 1" "wgrad 256 7 512 512 3 1"; do
  IIF_WGRAD_HALO=0 python scripts/prof_conv.py $a 20
  python scripts/prof_conv.py $a 20
done
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "IIF_WGRAD_HALO=0"
run "X=0"
run "IIF_WGRAD_HALO=0"
run "X=0"
