#!/bin/bash
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/prof_r5_g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_kt -- python3 $root/bench.py --no-cpu-baseline --no-kernel-events --no-fp32-step --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(find /tmp/p_kt -name "*.db" | head -1)
python3 $root/scripts/rocpd_timeline.py $db 0 $out/r5_g_step_listing.txt > $out/r5_g_step_timeline.txt
head -4 $out/r5_g_step_timeline.txt
tail -22 $out/r5_g_step_listing.txt | cut -c1-120
