#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
B=${1:-1}
timeout -k 10 100 python3 $R/tools/prof_pipe.py $B 200 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/tools/prof_pipe.py $B 40 > $R/gpurun_out/prof_pipe.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_pipe.log; exit 1; }
python - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/prof_pipe/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steady state: take the middle third
n=len(rows); rows=rows[n//3:2*n//3]
t0=int(rows[0]['Start_Timestamp']); t1=int(rows[-1]['End_Timestamp'])
busy=collections.defaultdict(int); cnt=collections.defaultdict(int)
for r in rows:
    q=r['Queue_Id']; busy[q]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); cnt[q]+=1
print('window %.1f us'%((t1-t0)/1e3))
for q in busy: print('queue',q,'busy %.1f%%'%(100*busy[q]/(t1-t0)),'kernels',cnt[q])
convs=[r for r in rows if 'Lb1ELi32' in r['Kernel_Name'] or 'true, 32' in r['Kernel_Name']]
if len(convs)>2:
    d=[int(convs[i+1]['Start_Timestamp'])-int(convs[i]['Start_Timestamp']) for i in range(len(convs)-1)]
    print('conv1b start-to-start us:',[round(x/1e3) for x in d[:12]])
# one frame timeline of the conv queue
i0=rows.index(convs[1]) if len(convs)>1 else 0
tt=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+70]:
    s=int(r['Start_Timestamp'])-tt; e=int(r['End_Timestamp'])-tt
    print('%8.1f %8.1f q%s %s'%(s/1e3,e/1e3,r['Queue_Id'],r['Kernel_Name'][:60]))
PY
