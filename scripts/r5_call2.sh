#!/bin/bash
set -e -o pipefail
root=$(pwd)
mkdir -p gpurun_out/r5
scripts/micro/stream_rw.bin > gpurun_out/r5/stream_rw.txt 2>&1 || true
cat gpurun_out/r5/stream_rw.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests_a.log 2>&1 || { tail -40 gpurun_out/r5/gpu_tests_a.log; exit 1; }
tail -3 gpurun_out/r5/gpu_tests_a.log
bash scripts/ab_prev.sh 2 > gpurun_out/r5/ab_a.txt 2>&1; cat gpurun_out/r5/ab_a.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/cont_kt -- python3 $root/scripts/bm_contention.py --once > $root/gpurun_out/r5/cont_kt.log 2>&1
f=$(find /tmp/cont_kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $root/gpurun_out/r5/contention_resources.txt <<'PY'
import csv, sys
seen = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if k in seen: continue
    seen[k] = r
    print("%-110s lds %7s  vgpr %4s agpr %4s sgpr %4s  wg %5s grid %9s" % (k[:110], r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Workgroup_Size"), r.get("Grid_Size")))
PY
cat $root/gpurun_out/r5/contention_resources.txt
