#!/bin/bash
# A/B of two builds of the library on one box: ab/lib_old.so and ab/lib_new.so, alternating, three runs each
for i in 1 2 3; do
  for v in old new; do
    cp ab/lib_$v.so ppg-slam_amd/libppgfront.so
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-split-pass 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], j['extract_only_frames_per_s'])" || exit 1
  done
done
