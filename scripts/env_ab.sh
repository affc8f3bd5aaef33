#!/bin/bash
# A/B of runtime environment knobs on one box
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HSA_ENABLE_INTERRUPT=0"
run "GPU_MAX_HW_QUEUES=4"
run "X=0"
