#!/bin/bash
root=$(pwd)
mkdir -p gpurun_out/r5
bash scripts/collect_profiles.sh r5_c > gpurun_out/r5/collect_c.log 2>&1; tail -4 gpurun_out/r5/collect_c.log
db=$(find /tmp/p_kt -name "*.db" | head -1)
python3 scripts/rocpd_timeline.py $db 2 gpurun_out/r5/r5_c_listing.txt > /dev/null 2>&1 || true
ls -la gpurun_out/r5/r5_c_listing.txt
