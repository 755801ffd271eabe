#!/bin/bash
# PPG_XCD_MAP group sizes / PPG_XCD_BAND on one box, alternating, in one arithmetic mode: ab_xcd.sh PREC "map:band map:band ..." (- = unset)
prec=$1
for i in 1 2 3; do
  for v in $2; do
    if [ "$v" = "-" ]; then unset PPG_XCD_MAP PPG_XCD_BAND; else export PPG_XCD_MAP="${v%%:*}" PPG_XCD_BAND="${v##*:}"; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-split-pass --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prec $prec map:band=$v', j['value'], j['ms_per_step'], 'extract only', j['extract_only_frames_per_s'], 'frac', j['roofline']['frac'], 'other conv ms', j['device_ms_per_step']['conv3x3'])" || exit 1
  done
done
