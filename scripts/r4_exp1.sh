#!/bin/bash
# round 4, experiment 1: do contiguous split-K ranges camp on memory channels?
set -e
mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gpurun_out/r4/camping scripts/micro/camping.hip
gpurun_out/r4/camping > gpurun_out/r4/camping.log 2>&1
echo camping done
python scripts/bm_wgrad1x1.py --check > gpurun_out/r4/wg_base.log 2>&1
echo base done
for c in 1 4 16; do
  IIF_WGRAD_CHUNK=$c python scripts/bm_wgrad1x1.py --check > gpurun_out/r4/wg_c$c.log 2>&1
  echo chunk $c done
done
for v in 0 4 0 4 1 16; do
  IIF_WGRAD_CHUNK=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events > gpurun_out/r4/bench_c$v.$RANDOM.log 2>gpurun_out/r4/bench_err.log
  echo bench $v done
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r4/bench_c*.log
