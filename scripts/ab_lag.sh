# experiment: how many blocks the weight-gradient stream may lag behind the compute stream
for i in 1 2; do for v in 2 3 4; do
  export IIF_WGRAD_LAG=$v
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad lag $v', d['ms_per_step'])" || exit 1
done; done
