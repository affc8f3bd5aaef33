#!/bin/bash
# kernel trace of another bench configuration: scripts/kt_cfg.sh <tag> <bench args...>   (on the GPU box)
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_$tag -- python3 $root/bench.py --no-cpu-baseline --no-kernel-events --no-fp32-step --steps 10 --warmup 3 "$@" > $out/$tag.log 2>&1
db=$(find /tmp/p_$tag -name "*.db" | head -1)
python3 $root/scripts/rocpd_stats.py $db $out/${tag}_stats.csv
python3 $root/scripts/rocpd_timeline.py $db 2 $out/${tag}_listing.txt > $out/${tag}_timeline.txt
head -5 $out/${tag}_timeline.txt
tail -1 $out/$tag.log | cut -c1-200
