#!/bin/bash
# reduced-precision modes: conv-only floor and isolated kernel times
set -o pipefail
R=$GRAFT_REPO_ROOT
for p in 3 1; do
for m in 0 7; do
  PPG_DEBUG_SKIP=$m timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-pass --no-sweep --precision $p > gpurun_out/prec_${p}_$m.log 2>&1 || { tail -n 20 gpurun_out/prec_${p}_$m.log; exit 1; }
  python - <<PY
import json
j=json.loads(open('gpurun_out/prec_${p}_$m.log').read().strip().splitlines()[-1])
print('precision',$p,'skip',$m,'value',j['value'],'ms',j['ms_per_step'],'extract_only',j.get('extract_only_frames_per_s'), j.get('device_ms_per_step'))
PY
done
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p3 -- python3 $R/tools/prof_blocking.py 32 3 > $R/gpurun_out/prof_p3.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_p3.log; exit 1; }
python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_p3/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10))
PY
