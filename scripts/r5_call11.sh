#!/bin/bash
root=$(pwd)
mkdir -p gpurun_out/r5
bash scripts/collect_profiles.sh r5_b > gpurun_out/r5/collect_b.log 2>&1; tail -4 gpurun_out/r5/collect_b.log
python bench.py --steps 20 --warmup 5 --per-shape > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/pershape.txt; head -c 400 gpurun_out/r5/bench_default.json; echo
python bench.py --no-fp32-step --model resnet32 --batch 128 --image 32 --classes 100 > gpurun_out/r5/r5_cfg1_bench.json 2> gpurun_out/r5/cfg1.err; head -c 300 gpurun_out/r5/r5_cfg1_bench.json; echo
python bench.py --no-cpu-baseline --no-fp32-step --model resnext101_32x4d --classes 365 --batch 128 > gpurun_out/r5/r5_cfg4_bench.json 2> gpurun_out/r5/cfg4.err; head -c 300 gpurun_out/r5/r5_cfg4_bench.json; echo
bash scripts/bench_configs.sh > gpurun_out/r5/bench_configs.txt 2>&1; tail -25 gpurun_out/r5/bench_configs.txt
