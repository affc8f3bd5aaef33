#!/bin/bash
set -e -o pipefail
root=$(pwd)
mkdir -p gpurun_out/r5
scripts/micro/stream_rw.bin 2>&1 | grep coef > gpurun_out/r5/stream_rw_coef.txt || true
cat gpurun_out/r5/stream_rw_coef.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests_a.log 2>&1 || { tail -40 gpurun_out/r5/gpu_tests_a.log; exit 1; }
tail -3 gpurun_out/r5/gpu_tests_a.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
for i in 1 2; do
  (cd _prev && $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev: /')
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new:  /'
  IIF_BN_GRID_CAP=1000000000 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new, bn grid uncapped:  /'
  IIF_BN_GRID_CAP=16384 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new, bn grid 16384:  /'
done 2>&1 | tee gpurun_out/r5/ab_a.txt
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
