#!/bin/bash
# sixteen-taps-per-block stem weight gradient vs the tap-per-tile kernel
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "IIF_WGRAD_STEM=0"
run "X=0"
run "IIF_WGRAD_STEM=0"
run "X=0"
run "IIF_WGRAD_STEM_BLOCKS=4"
run "IIF_WGRAD_STEM_BLOCKS=1"
