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
python3 $R/tools/iso_report.py $R/gpurun_out/iso_stats_p$P.csv $O/*/*_kernel_trace.csv
