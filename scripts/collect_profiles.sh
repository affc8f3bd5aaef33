#!/bin/bash
# The rocprofv3 passes behind profiles/r<round>_<tag>_*: run ON the GPU box from the repo root (gpurun).
#   scripts/collect_profiles.sh r4_a
# 1 kernel trace (stats + timeline), 2 HBM counter passes (FETCH_SIZE / WRITE_SIZE, one counter each), 1 MFMA pass.
# Counters are collected in passes of their own (--kernel-trace + --pmc only), as the pool requires.
set -e -o pipefail
tag=${1:-r3_a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_m      # a leased box keeps /tmp between calls: never read a stale database
BENCH="$root/bench.py --no-cpu-baseline --no-kernel-events --no-fp32-step"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_kt -- python3 $BENCH --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(find /tmp/p_kt -name "*.db" | head -1)
python3 $root/scripts/rocpd_stats.py $db $out/${tag}_kernel_stats.csv
python3 $root/scripts/rocpd_timeline.py $db 0 $out/${tag}_step_listing.txt > $out/${tag}_step_timeline.txt
echo "trace done" > $out/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/p_f -- python3 $BENCH --steps 4 --warmup 2 > $out/f.log 2>&1
echo "fetch done" >> $out/progress.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/p_w -- python3 $BENCH --steps 4 --warmup 2 > $out/w.log 2>&1
echo "write done" >> $out/progress.txt
python3 $root/scripts/rocpd_hbm.py $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) 6 $out/${tag}_pmc_hbm_traffic.csv
python3 $root/scripts/rocpd_hbm_listing.py $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) $out/${tag}_hbm_per_launch.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format rocpd -d /tmp/p_m -- python3 $BENCH --steps 4 --warmup 2 > $out/m.log 2>&1
python3 $root/scripts/rocpd_mfma.py $(find /tmp/p_m -name "*.db" | head -1) 6 $out/${tag}_pmc_mfma_util.csv
echo "all done" >> $out/progress.txt
tail -2 $out/${tag}_pmc_hbm_traffic.csv
head -3 $out/${tag}_step_timeline.txt
