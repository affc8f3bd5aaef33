#!/bin/bash
# side streams on / off for small networks (one gpurun call):  where does the weight-gradient stream start to pay?
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 5 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for bs in 128 256 512 1024 2048; do
  echo -n "resnet32 bs $bs  side: "; run --model resnet32 --batch $bs --image 32 --classes 100
  echo -n "resnet32 bs $bs  none: "; IIF_NO_WGRAD_STREAM=1 run --model resnet32 --batch $bs --image 32 --classes 100
done
for cfg in "resnet18 64 224" "resnet50 16 224" "resnet50 32 128" "resnet50 64 224"; do
  set -- $cfg
  echo -n "$1 bs $2 img $3 side: "; run --model $1 --batch $2 --image $3
  echo -n "$1 bs $2 img $3 none: "; IIF_NO_WGRAD_STREAM=1 run --model $1 --batch $2 --image $3
done
