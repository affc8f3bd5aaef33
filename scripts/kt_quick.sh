#!/bin/bash
# quick kernel trace of the default bench step: scripts/kt_quick.sh <tag> [name-filter]  (on the GPU box; KT_ARGS = extra bench arguments)
set -e -o pipefail
tag=${1:-kt}; filt=${2:-.}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${KT_DIR:-r6}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_$tag -- python3 $root/bench.py --no-cpu-baseline --no-kernel-events --no-fp32-step --steps 10 --warmup 3 $KT_ARGS > $out/$tag.log 2>&1
db=$(find /tmp/p_$tag -name "*.db" | head -1)
python3 $root/scripts/rocpd_stats.py $db $out/${tag}_stats.csv
python3 $root/scripts/rocpd_timeline.py $db 2 $out/${tag}_listing.txt > $out/${tag}_timeline.txt
python3 - $out/${tag}_stats.csv "$filt" <<'PY'
import csv, sys, re
rows = list(csv.reader(open(sys.argv[1])))[1:]
for r in rows:
    if re.search(sys.argv[2], r[0]):
        print("%-100s n/step %6.1f ms/step %7.3f avg %8.1f us" % (r[0][:100], float(r[1]) / 13, float(r[2]) / 13e6, float(r[3]) / 1e3))
PY
head -4 $out/${tag}_timeline.txt
