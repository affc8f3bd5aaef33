#!/bin/bash
# 1-rank RCCL run of the bucketed reducer under different stream / hardware-queue settings
run() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --force-reducer 2>&1 | tail -1 | cut -c60-100; }
echo "default:"; run 29601
echo "GPU_MAX_HW_QUEUES=8:"; GPU_MAX_HW_QUEUES=8 run 29602
echo "wgrad stream priority -1:"; IIF_WGRAD_STREAM_PRIORITY=-1 run 29603
echo "no wgrad stream:"; IIF_NO_WGRAD_STREAM=1 run 29604
echo "no reducer, plain:"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c60-100
echo "no reducer, GPU_MAX_HW_QUEUES=8:"; GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c60-100
