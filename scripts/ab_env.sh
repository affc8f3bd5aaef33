# generic in-step A/B: scripts/ab_env.sh VAR [runs]   (default bench step with VAR unset, then VAR=1, alternating)
var=$1; runs=${2:-2}
for i in $(seq $runs); do for v in 0 1; do
  if [ $v = 1 ]; then export $var=1; else unset $var; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['ms_per_step'])" || exit 1
done; done
