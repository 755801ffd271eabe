#!/bin/bash
# A/B of several values of one environment variable on one box: ab_envs.sh VAR v1 v2 ... ("-" = unset), three rounds
var=$1; shift
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var="$v"; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-split-pass 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', j['value'], j['ms_per_step'], j['extract_only_frames_per_s'], j['roofline']['frac'])" || exit 1
  done
done
