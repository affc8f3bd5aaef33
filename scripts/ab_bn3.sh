# in-step A/B of the algebraic BN3 backward
mkdir -p gpurun_out/r3
for v in off default all_hybrid all_pure s12 off default; do
  unset IIF_NO_BN3_ALGEBRA IIF_BN3_ALGEBRA_MAXC IIF_BN3_ALGEBRA_MIN_ELEMS IIF_BN3_ALGEBRA_PURE_MIN_ELEMS
  case $v in off) export IIF_NO_BN3_ALGEBRA=1;; default) ;; all_hybrid) export IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=1e30;;
    all_pure) export IIF_BN3_ALGEBRA_PURE_MIN_ELEMS=0;; s12) export IIF_BN3_ALGEBRA_MAXC=128;; esac
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn3 algebra $v', d['ms_per_step'])" || exit 1
done
