#!/bin/bash
# the whole GPU suite + smoke + an A/B against _prev (one gpurun call)
set -e
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t_all.log 2>&1 || { tail -40 gpurun_out/r4/t_all.log; exit 1; }
tail -2 gpurun_out/r4/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.log 2>&1 || { tail -20 gpurun_out/r4/smoke.log; exit 1; }
tail -2 gpurun_out/r4/smoke.log
[ -d _prev ] && bash scripts/ab_prev.sh 2
