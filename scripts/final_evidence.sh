#!/bin/bash
# end-of-round evidence on the GPU box:  scripts/final_evidence.sh <tag, e.g. r6_b>
# the profile set (scripts/collect_profiles.sh: kernel trace + timeline, two HBM counter passes, one MFMA pass), the default bench
# line, BASELINE configs 1 and 4 (1-indexed: ResNet32 CIFAR / ResNeXt-101), the per-shape table.  Everything lands under gpurun_out/<tag>/.
tag=${1:-r6_a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
bash scripts/collect_profiles.sh $tag > $out/collect.log 2>&1 && echo "profiles ok"
cp gpurun_out/prof_$tag/${tag}_* $out/ 2>/dev/null
cp gpurun_out/prof_$tag/${tag}_pmc_hbm_traffic.csv profiles/ 2>/dev/null      # bench.py reads the newest traffic table
python bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err && tail -1 $out/${tag}_bench_default.json | cut -c1-330
python bench.py --no-cpu-baseline --no-fp32-step --per-shape --event-every 5 > $out/bench_pershape.json 2> $out/${tag}_pershape.txt; grep -c "conv\]" $out/${tag}_pershape.txt
python bench.py --no-fp32-step --model resnet32 --batch 128 --image 32 --classes 100 > $out/${tag}_cfg1_bench.json 2> $out/cfg1.err; cut -c1-200 $out/${tag}_cfg1_bench.json
python bench.py --no-cpu-baseline --no-fp32-step --model resnext101_32x4d --classes 365 --batch 128 > $out/${tag}_cfg4_bench.json 2> $out/cfg4.err; cut -c1-200 $out/${tag}_cfg4_bench.json
