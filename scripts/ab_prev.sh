#!/bin/bash
# A/B of the working tree against the copy of an earlier commit under _prev/ (built here, shipped with the snapshot), in ONE
# gpurun call: boxes differ by ~1 %, so only same-call pairs count.   scripts/ab_prev.sh [rounds] [bench args...]
rounds=${1:-2}; shift || true
root=$(pwd)
for i in $(seq $rounds); do
  (cd $root/_prev && timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev: /')
  (cd $root && timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new:  /')
done
