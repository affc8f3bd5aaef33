run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])" || exit 1; }
for i in 1 2; do
unset IIF_CONV_HALO_BM IIF_CONV_NO_HALO; run base
IIF_CONV_HALO_BM=128 run halo_bm128
IIF_CONV_HALO_BM=256 run halo_bm256
IIF_CONV_NO_HALO=1 run no_halo
done
