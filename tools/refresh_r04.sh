#!/bin/bash
# Round-4 evidence in one gpurun call (everything lands in gpurun_out/; copy what is judged to profiles/r04_*):
#   PMC traffic (batch-1 frame; B = 32 step and R-row frames differenced) FIRST, installed as profiles/r04_* on the box so the bench line
#   that follows cites them; full bench line; rocprofv3 kernel stats of the same command; B = 32 step kernel stats at two frame counts;
#   request-row kernel stats; configs[4] long-form run; roofline re-check over all of it.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
bash tools/pmc_batch.sh > $O/pmc_batch.log 2>&1; tail -3 $O/pmc_batch.log
[ -s $O/pmc_hbm_traffic.json ] && cp $O/pmc_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json
[ -s $O/pmc_batch_traffic.json ] && cp $O/pmc_batch_traffic.json profiles/r04_pmc_batch_traffic.json
bash tools/run_bench_prof.sh > $O/run_bench_prof.log 2>&1
for F in 32 96; do bash tools/prof_batch.sh 32 $F > $O/prof_batch_F$F.txt 2>&1; cp $O/batch_kernel_stats.csv $O/batch_kernel_stats_F$F.csv; done
export TMPDIR=/tmp; cd /tmp
for R in 4 8; do
  rm -rf /tmp/profr$R
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profr$R -o r -- python $GRAFT_REPO_ROOT/tools/pmc_rows_run.py $R 256 > $O/prof_rows_R$R.log 2>&1
  F=$(find /tmp/profr$R -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/rows_kernel_stats_R$R.csv && head -4 $F
done
cd $GRAFT_REPO_ROOT
timeout 600 python tools/longform_bench.py 4096 fp8 > $O/longform_fp8.txt 2>&1; tail -4 $O/longform_fp8.txt
python tools/check_roofline.py $O/bench_line.json $O/bench_kernel_stats.csv $O/pmc_hbm_traffic.json $O/batch_kernel_stats_F32.csv $O/batch_kernel_stats_F96.csv \
  $O/pmc_batch_traffic.json > $O/check_roofline.txt 2>&1; tail -40 $O/check_roofline.txt
