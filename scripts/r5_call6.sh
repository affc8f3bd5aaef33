#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests_b.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r5/gpu_tests_b.log
B="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-step --no-kernel-events"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
{
echo "# bench.py --force-reducer (one RCCL rank: the bucket / stream machinery runs, the collectives move no data) by CU budget of the"
echo "# persistent kernels during backward; ms per step, two rounds, same call.  What the reservation COSTS; what it buys needs >= 2 GPUs."
for i in 1 2; do
  python $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/no reducer:            /'
  for cu in 0 248 240 224 192; do
    $TR --master-port $((29600 + cu % 97)) $B --force-reducer --reducer-cus $cu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/force-reducer, budget $cu: /"
  done
done
} 2>&1 | tee gpurun_out/r5/reducer_cu_budget.txt
python bench.py --gpus 2 --backend gloo --device-index 0 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-step 2>gpurun_out/r5/gloo2.err | tail -1 > gpurun_out/r5/bench_gloo2.json; head -c 600 gpurun_out/r5/bench_gloo2.json; echo
$TR --master-port 29711 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-step --force-reducer 2>/dev/null | tail -1 > gpurun_out/r5/bench_force_reducer.json; head -c 300 gpurun_out/r5/bench_force_reducer.json; echo
{ python scripts/graph_probe.py resnet50 256 224 1000 --reducer; python scripts/graph_probe.py resnet50 256 224 1000; } 2>&1 | grep -v amdgpu | tee gpurun_out/r5/graph_reducer.txt
