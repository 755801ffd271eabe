#!/bin/bash
# one GPU iteration: line/match parity tests, a short bench, isolated kernel times; stops at the first failure
set -o pipefail
tag=${1:-x}
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "${2:-line or tail or match}" > gpurun_out/t_$tag.log 2>&1 || { tail -n 30 gpurun_out/t_$tag.log; exit 1; }
tail -n 2 gpurun_out/t_$tag.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 > gpurun_out/b_$tag.log 2>&1 || { tail -n 30 gpurun_out/b_$tag.log; exit 1; }
python - <<PY
import json
j=json.loads(open('gpurun_out/b_$tag.log').read().strip().splitlines()[-1])
print('value',j['value'],'ms',j['ms_per_step'],'upload',j['with_upload']['value'],'frac',j['roofline']['frac'],'f16',j['f16_split_mode']['value'],'bf16',j.get('bf16_split_mode',{}).get('value'))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/prof_blocking.py > $R/gpurun_out/prof_$tag.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_$tag.log; exit 1; }
python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_$tag/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:32]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10))
PY
