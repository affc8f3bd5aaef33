#!/bin/bash
# A/B of environment settings on the working tree in ONE gpurun call:  scripts/ab_env.sh rounds "ENV1=.. ENV2=.." "ENV3=.." ...
# ("-" = no extra environment)
rounds=$1; shift
for i in $(seq $rounds); do
  for e in "$@"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    echo -n "[$e] "
    env $ee timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
