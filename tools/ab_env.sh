#!/bin/bash
# A/B of an environment switch on one box: alternating runs without / with VAR=1, three each
var=$1
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-split-pass 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', j['value'], j['ms_per_step'], 'frac', j['roofline']['frac'], 'conv1b ms', j['roofline']['avg_launch_ms'], 'other conv ms', j['device_ms_per_step']['conv3x3'])" || exit 1
  done
done
