#!/bin/bash
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_iif_head_gpu.py tests/test_mmdet_golden.py tests/test_mmdet_fasa_gpu.py -x -q -m gpu > gpurun_out/r4/t_head.log 2>&1 || { tail -40 gpurun_out/r4/t_head.log; exit 1; }
tail -1 gpurun_out/r4/t_head.log
python scripts/bench_iif_head.py > gpurun_out/r4/head_bw.log 2>&1; grep -v amdgpu gpurun_out/r4/head_bw.log | tail -20
(cd _prev && python scripts/bench_iif_head.py 2>&1 | grep -v amdgpu | tail -20 > ../gpurun_out/r4/head_bw_prev.log); echo PREV; cat gpurun_out/r4/head_bw_prev.log
