#!/bin/bash
# as ab_envs.sh, at a reduced-precision mode: ab_envs_prec.sh PREC VAR v1 v2 ...
prec=$1; var=$2; shift; shift
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var="$v"; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-split-pass --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prec $prec $var=$v', j['value'], j['ms_per_step'], j['extract_only_frames_per_s'], j['roofline']['frac'])" || exit 1
  done
done
