#!/bin/bash
# odd batch / image sizes and the other architectures through the whole training step (bf16): finite loss, no fault
run() { echo "== $*"; timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events "$@" 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('   ', d['value'], 'img/s  loss', d['config']['final_loss'])
except Exception as e:
    print('    FAILED', e)"; }
run --batch 50 --image 160
run --batch 33 --image 97
run --batch 7 --image 225 --classes 13
run --model resnet18 --batch 64
run --model resnet34 --batch 48 --image 192
run --model resnet101 --batch 64
run --model resnet152 --batch 32
run --model wide_resnet50_2 --batch 64
run --model resnext50_32x4d --batch 64 --classes 365
run --model resnext101_32x8d --batch 32 --classes 365
run --model se_resnext50_32x4d --batch 64
run --model se_resnet152 --batch 32
run --model resnet110 --batch 128 --image 32 --classes 10
run --model se_resnet32 --batch 100 --image 32 --classes 100
run --model resnet50 --batch 64 --dtype f32
