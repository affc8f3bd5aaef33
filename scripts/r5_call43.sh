#!/bin/bash
mkdir -p gpurun_out/r5
python bench.py > gpurun_out/r5/r5_f_bench_default.json 2> gpurun_out/r5/bench_default_f.err && tail -1 gpurun_out/r5/r5_f_bench_default.json | cut -c1-200
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gpu_tests_n.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r5/gpu_tests_n.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
