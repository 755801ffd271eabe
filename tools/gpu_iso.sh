#!/bin/bash
# Isolated (blocking, un-pipelined) kernel statistics of one arithmetic mode: usage gpu_iso.sh PRECISION
set -o pipefail
R=$GRAFT_REPO_ROOT
P=${1:-1}
O=$R/gpurun_out/iso_p$P
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/prof_blocking.py 32 $P > $O/log.txt 2>&1 || { tail -n 20 $O/log.txt; exit 1; }
cp $O/*/*_kernel_stats.csv $R/gpurun_out/iso_stats_p$P.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/iso_stats_p$P.csv')))
steps=6.0
conv=post=0.0
for r in rows:
    t=int(r['TotalDurationNs'])/1e3
    per=t/steps if 'match' not in r['Name'] and 'row_norm' not in r['Name'] and 'transpose' not in r['Name'] else t/5.0
    if 'conv_' in r['Name']: conv+=per
    elif 'rocclr' not in r['Name'] and 'upload' not in r['Name']: post+=per
    print("%-64s calls %4s avg %8.1f us  per step %8.1f us" % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, per))
print("conv per step %.1f us, post+match per step %.1f us" % (conv, post))
PY
