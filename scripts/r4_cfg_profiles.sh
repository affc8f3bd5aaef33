#!/bin/bash
# bench lines + kernel traces of BASELINE configs 1 and 4 for profiles/r4_cfg*  (on the GPU box)
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r4; mkdir -p $out
python bench.py --no-fp32-step --model resnet32 --batch 128 --image 32 --classes 100 > $out/r4_cfg1_bench.json 2> $out/cfg1.err
python bench.py --no-cpu-baseline --no-fp32-step --model resnext101_32x4d --classes 365 --batch 128 > $out/r4_cfg4_bench.json 2> $out/cfg4.err
KT_ARGS="--model resnet32 --batch 128 --image 32 --classes 100" bash scripts/kt_quick.sh cfg1 zzzz > /dev/null
KT_ARGS="--model resnext101_32x4d --classes 365 --batch 128" bash scripts/kt_quick.sh cfg4 zzzz > /dev/null
head -c 300 $out/r4_cfg1_bench.json; echo; head -c 300 $out/r4_cfg4_bench.json; echo
head -3 $out/cfg1_timeline.txt; head -3 $out/cfg4_timeline.txt
