#!/bin/bash
# round-5 opening call: baseline bench of the start-of-round build, the contention table, the IIF head evidence
set -e -o pipefail
root=$(pwd)
mkdir -p gpurun_out/r5
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/r5/base.txt
cat gpurun_out/r5/base.txt
python scripts/bm_contention.py 2>&1 | grep -v amdgpu > gpurun_out/r5/contention.txt
cat gpurun_out/r5/contention.txt
python scripts/bench_iif_head.py 2>&1 | grep -v amdgpu > gpurun_out/r5/head_bw.txt
cat gpurun_out/r5/head_bw.txt
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
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/head_kt -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_kt.log 2>&1
cp $(find /tmp/head_kt -name "*kernel_stats.csv" | head -1) $root/gpurun_out/r5/head_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/head_f -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_f.log 2>&1
cp $(find /tmp/head_f -name "*counter_collection.csv" | head -1) $root/gpurun_out/r5/head_fetch.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/head_w -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_w.log 2>&1
cp $(find /tmp/head_w -name "*counter_collection.csv" | head -1) $root/gpurun_out/r5/head_write.csv
head -5 $root/gpurun_out/r5/head_kernel_stats.csv
