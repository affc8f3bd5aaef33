# experiment: blocks per CU of the split-K weight gradients (fewer blocks = fewer slab bytes, less latency hiding)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])" || exit 1; }
for i in 1 2; do
unset IIF_WGRAD_SLOT_PCT IIF_WGRAD_SLOT_PCT_256 IIF_WGRAD_SLOT_PCT_128 IIF_WGRAD_SLOT_PCT_64
IIF_WGRAD_SLOT_PCT_256=200 run w256_two_per_cu
IIF_WGRAD_SLOT_PCT_256=150 run w256_1.5
IIF_WGRAD_SLOT_PCT_256=125 run w256_1.25
IIF_WGRAD_SLOT_PCT_256=100 run w256_one_per_cu
IIF_WGRAD_SLOT_PCT_256=75 run w256_0.75
done
