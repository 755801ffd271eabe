#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_b1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -- python3 $R/tools/prof_blocking.py ${1:-1} ${2:-0} > $R/gpurun_out/prof_b1.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_b1.log; exit 1; }
python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_b1/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last extract: find the last conv1b-fused kernel start
idx=[i for i,r in enumerate(rows) if '8, true' in r['Kernel_Name'] or 'conv_ws_kernel<true' in r['Kernel_Name'] or 'ILi3ELi2ELi8ELb1' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0-2:]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print('%8.1f %8.1f %7.1f  q%s  %s grid %s wg %s'%(s/1e3,e/1e3,(e-s)/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:70],r.get('Grid_Size_X','?'),r.get('Workgroup_Size_X','?')))
PY
