#!/bin/bash
root=$(pwd)
mkdir -p gpurun_out/r5
bash scripts/collect_profiles.sh r5_a > gpurun_out/r5/collect_a.log 2>&1; tail -5 gpurun_out/r5/collect_a.log
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
cd $root && python scripts/bm_contention.py 2>&1 | grep -v amdgpu > gpurun_out/r5/contention2.txt; cat gpurun_out/r5/contention2.txt
python scripts/bench_iif_head.py 2>&1 | grep -v amdgpu > gpurun_out/r5/head_bw2.txt; cat gpurun_out/r5/head_bw2.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/head_kt -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_kt.log 2>&1
cp $(find /tmp/head_kt -name "*kernel_stats.csv" | head -1) $root/gpurun_out/r5/head_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/head_f -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_f.log 2>&1
cp $(find /tmp/head_f -name "*counter_collection.csv" | head -1) $root/gpurun_out/r5/head_fetch.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/head_w -- python3 $root/scripts/prof_iif_head.py > $root/gpurun_out/r5/head_w.log 2>&1
cp $(find /tmp/head_w -name "*counter_collection.csv" | head -1) $root/gpurun_out/r5/head_write.csv
head -6 $root/gpurun_out/r5/head_kernel_stats.csv
cd $root && python -m pytest tests/test_iif_head_gpu.py -x -q -m gpu 2>&1 | tail -2
