#!/bin/bash
# Kernel statistics of the default bench at a reduced-precision mode with and without the wave-specialised convolutions.
# usage: prof_ws.sh PREC [extra env assignments...]   -> gpurun_out/prof_ws/
set -o pipefail
prec=${1:-1}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_ws
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ws in 0 1; do
  export PPG_CONV_WS=$ws
  for kv in "$@"; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ws$ws -- python3 $R/bench.py --no-cpu-baseline --no-split-pass --no-sweep --steps 12 --warmup 3 --precision $prec > $O/ws$ws.log 2>&1 || { tail -n 20 $O/ws$ws.log; exit 1; }
  cp $O/ws$ws/*/*_kernel_stats.csv $O/kernel_stats_ws$ws.csv
  rm -rf $O/ws$ws
  echo "== PPG_CONV_WS=$ws"; grep '^{' $O/ws$ws.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'])"
  head -n 14 $O/kernel_stats_ws$ws.csv | cut -c1-200
done
