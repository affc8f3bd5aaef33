#!/bin/bash
mkdir -p gpurun_out/r5
bash scripts/collect_profiles.sh r5_d > gpurun_out/r5/collect_d.log 2>&1 && echo "profiles ok"
python bench.py > gpurun_out/r5/bench_default_d.json 2> gpurun_out/r5/bench_default_d.err && tail -1 gpurun_out/r5/bench_default_d.json | cut -c1-400
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gpu_tests_d.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r5/gpu_tests_d.log
