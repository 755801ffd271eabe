#!/bin/bash
# step time with stages skipped (measurement aid; results are wrong by construction)
# usage: skip_sweep.sh VAR v1 v2 ...   (VAR = PPG_DEBUG_SKIP or PPG_DEBUG_MATCH_SKIP)
set -o pipefail
var=$1; shift
for m in "$@"; do
  env $var=$m timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-pass --no-sweep > gpurun_out/skip_$m.log 2>&1 || { tail -n 20 gpurun_out/skip_$m.log; exit 1; }
  python - <<PY
import json
j=json.loads(open('gpurun_out/skip_$m.log').read().strip().splitlines()[-1])
print('$var',$m,'value',j['value'],'ms',j['ms_per_step'],'extract_only',j.get('extract_only_frames_per_s'))
PY
done
