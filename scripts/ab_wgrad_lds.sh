# experiment: weight-gradient blocks padded to N KB of LDS (fewer per CU: room for the compute stream's kernels)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])" || exit 1; }
for i in 1 2; do
unset IIF_WGRAD_LDS_KB; run base
IIF_WGRAD_LDS_KB=54 run lds54_two_per_cu
IIF_WGRAD_LDS_KB=81 run lds81_one_per_cu
IIF_WGRAD_LDS_KB=110 run lds110
done
