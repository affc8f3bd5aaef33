#!/bin/bash
# final verification on the GPU box: the whole GPU suite, the smoke entry, one default bench line
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gpu_tests_final.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r5/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | cut -c1-220
