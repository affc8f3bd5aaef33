#!/bin/bash
# kernel trace + timeline of BASELINE configs[3] (ResNeXt-101-32x4d, bs 128, C=365) on the GPU box
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/prof_r5_x101; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_x101
A="--model resnext101_32x4d --batch 128 --classes 365 --no-cpu-baseline --no-kernel-events --no-fp32-step"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_x101 -- python3 $root/bench.py $A --steps 8 --warmup 3 > $out/kt.log 2>&1
db=$(find /tmp/p_x101 -name "*.db" | head -1)
python3 $root/scripts/rocpd_stats.py $db $out/r5_x101_kernel_stats.csv
python3 $root/scripts/rocpd_timeline.py $db 2 $out/r5_x101_step_listing.txt > $out/r5_x101_step_timeline.txt
head -6 $out/r5_x101_step_timeline.txt
cd $root && python3 bench.py $A --steps 20 --warmup 5 | cut -c1-300
