# in-step A/B of the two-pass conv3 forward: off / tile kernels / streaming kernel
for v in off tile stream off tile stream; do
  unset IIF_TWOPASS IIF_CONV_STREAM_TWOPASS
  case $v in off) ;; tile) export IIF_TWOPASS=1;; stream) export IIF_TWOPASS=1 IIF_CONV_STREAM_TWOPASS=1;; esac
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32-step --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-pass $v', d['ms_per_step'], d['config']['final_loss'])" || exit 1
done
