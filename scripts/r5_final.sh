#!/bin/bash
# end-of-round evidence on the GPU box: the r5_g profile set, the default bench line, BASELINE configs 1 and 4 (1-indexed: ResNet32
# CIFAR / ResNeXt-101), the per-shape table
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r5; mkdir -p $out
bash scripts/collect_profiles.sh r5_g > $out/collect_e.log 2>&1 && echo "profiles ok"
cp gpurun_out/prof_r5_g/r5_g_pmc_hbm_traffic.csv profiles/ 2>/dev/null      # bench.py reads the newest traffic table
python bench.py > $out/r5_g_bench_default.json 2> $out/bench_default_e.err && tail -1 $out/r5_g_bench_default.json | cut -c1-330
python bench.py --no-cpu-baseline --no-fp32-step --per-shape --event-every 5 > $out/bench_pershape.json 2> $out/r5_g_pershape.txt; grep -c "conv\]" $out/r5_g_pershape.txt
python bench.py --no-fp32-step --model resnet32 --batch 128 --image 32 --classes 100 > $out/r5_g_cfg1_bench.json 2> $out/cfg1.err; cut -c1-200 $out/r5_g_cfg1_bench.json
python bench.py --no-cpu-baseline --no-fp32-step --model resnext101_32x4d --classes 365 --batch 128 > $out/r5_g_cfg4_bench.json 2> $out/cfg4.err; cut -c1-200 $out/r5_g_cfg4_bench.json
