run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read().strip(); 
try:
    d=json.loads(l); print(d['metric'], d['value'], d['ms_per_step'])
except Exception as e:
    print('FAILED', l[:300])"; }
run --batch 512
run --batch 96 --image 192
run --batch 40 --image 160
run --model se_resnet50
run --model resnet101 --batch 128
run --model resnet152 --batch 64
run --model resnext50_32x4d --batch 128
run --model resnet18 --batch 256
run --model resnet32 --batch 128 --image 32 --classes 100
run --model se_resnet32 --batch 128 --image 32 --classes 100
run --model wide_resnet50_2 --batch 128
run --model wide_resnet101_2 --batch 64
run --model resnext101_32x8d --batch 64
run --model resnet34 --batch 256
run --model se_resnext50_32x4d --batch 128
run --model se_resnet152 --batch 64
