#!/bin/bash
# Counters behind the power/clock statement: the wave-specialised convolutions of tools/ubench/conv_bench (32 frames, mode 1) on
# all-zero and on random operands - the same instruction stream - once under --kernel-trace (durations) and once under --pmc
# (busy cycles).  Cycles that do not change with the data against durations that do = a clock that depends on the data.
# usage: prof_power_clock.sh   -> gpurun_out/power_clock/
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power_clock
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 0 1; do
  timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_d$d -- $R/tools/ubench/conv_bench $d 32 1 1 > $O/trace_d$d.log 2>&1 || { tail -n 20 $O/trace_d$d.log; exit 1; }
  cp $O/trace_d$d/*/*_kernel_stats.csv $O/kernel_stats_data$d.csv
  timeout -s KILL 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d$d -- $R/tools/ubench/conv_bench $d 32 1 1 > $O/pmc_d$d.log 2>&1 || { tail -n 20 $O/pmc_d$d.log; exit 1; }
  python3 $R/tools/summarise_counters.py $O/counters_data$d.csv $O/pmc_d$d
  rm -rf $O/trace_d$d $O/pmc_d$d
  echo "data $d done"
done
for d in 0 1; do echo "== data $d"; grep conv_ws $O/kernel_stats_data$d.csv | cut -c1-160; grep conv_ws $O/counters_data$d.csv | cut -c1-260; done
