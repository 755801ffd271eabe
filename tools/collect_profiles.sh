#!/bin/bash
# Round profile collection on the GPU box: kernel statistics of the default bench in the three arithmetic modes, the
# isolated (blocking) kernel times, and the counter passes (each in a run of its own, kernel trace only).
# usage: collect_profiles.sh TAG    -> gpurun_out/profiles_TAG/
set -o pipefail
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --no-cpu-baseline --no-split-pass --no-sweep"
for p in 0 1 3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p$p -- python3 $BENCH --steps 20 --warmup 5 --precision $p > $O/bench_p$p.log 2>&1 || { tail -n 20 $O/bench_p$p.log; exit 1; }
  cp $O/stats_p$p/*/*_kernel_stats.csv $O/kernel_stats_p$p.csv
  grep '^{' $O/bench_p$p.log | tail -n 1 > $O/bench_p$p.json
  echo "stats precision $p done"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iso -- python3 $R/tools/prof_blocking.py 32 0 > $O/iso.log 2>&1 || { tail -n 20 $O/iso.log; exit 1; }
cp $O/stats_iso/*/*_kernel_stats.csv $O/kernel_stats_isolated.csv
echo "isolated stats done"
SQ0="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
SQ3="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
for p in 0 1 3; do
  eval sq=\$SQ$p
  for pass in FETCH_SIZE WRITE_SIZE "$sq"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_p${p}_$name -- python3 $BENCH --steps 3 --warmup 1 --precision $p > $O/pmc_p${p}_$name.log 2>&1 || { tail -n 20 $O/pmc_p${p}_$name.log; exit 1; }
    echo "pmc precision $p $name done"
  done
  python3 $R/tools/summarise_counters.py $O/hbm_counters_p$p.csv $O/pmc_p${p}_FETCH_SIZE $O/pmc_p${p}_WRITE_SIZE
  python3 $R/tools/summarise_counters.py $O/mfma_counters_p$p.csv $O/pmc_p${p}_SQ_VALU_MFMA_BUSY_CYCLES
done
# keep only the summaries (the raw per-dispatch files are tens of MiB)
rm -rf $O/stats_p* $O/stats_iso $O/pmc_p*_FETCH_SIZE $O/pmc_p*_WRITE_SIZE $O/pmc_p*_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
