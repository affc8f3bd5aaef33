for i in 1 2 3; do for v in 1.5e8 1e8; do
  export IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=$v
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pure_min $v', d['ms_per_step'])" || exit 1
done; done
