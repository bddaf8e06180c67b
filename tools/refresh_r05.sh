#!/bin/bash
# Round-5 evidence in one gpurun call (everything lands in gpurun_out/; tools/install_r05.sh copies what is judged to profiles/r05_*):
#   PMC traffic (batch-1 frame; B = 32 step and R-row frames differenced) FIRST, installed as profiles/r05_* on the box so the bench line
#   that follows cites them; full bench line; rocprofv3 kernel stats of the same command; B = 32 step kernel stats at two frame counts;
#   request-row kernel stats; configs[4] long-form run + 4 concurrent fp8 streams; per-stage profile; parity log; roofline re-check.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
git rev-parse --short HEAD > $O/r05_commit.txt 2>/dev/null || true
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
PMC_BATCH_PREV=r04_pmc_batch_traffic.json bash tools/pmc_batch.sh > $O/pmc_batch.log 2>&1; tail -3 $O/pmc_batch.log
[ -s $O/pmc_hbm_traffic.json ] && cp $O/pmc_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json
[ -s $O/pmc_batch_traffic.json ] && cp $O/pmc_batch_traffic.json profiles/r05_pmc_batch_traffic.json
bash tools/run_bench_prof.sh > $O/run_bench_prof.log 2>&1
for F in 32 96; do bash tools/prof_batch.sh 32 $F > $O/prof_batch_F$F.txt 2>&1; cp $O/batch_kernel_stats.csv $O/batch_kernel_stats_F$F.csv; done
export TMPDIR=/tmp; cd /tmp
for R in 4 8; do
  rm -rf /tmp/profr$R
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profr$R -o r -- python $GRAFT_REPO_ROOT/tools/pmc_rows_run.py $R 256 > $O/prof_rows_R$R.log 2>&1
  F=$(find /tmp/profr$R -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/rows_kernel_stats_R$R.csv && head -4 $F
done
cd $GRAFT_REPO_ROOT
timeout 900 python tools/longform_bench.py 4096 fp8 f16 x 4 1024 2>&1 | tail -3 > $O/longform_fp8.txt; tail -4 $O/longform_fp8.txt
for wg in 0 255; do FISHRT_PERSIST_PROF_WG=$wg FISHRT_PERSIST_PROF=1 python tools/p2_quick.py bf16 2>&1 | tail -3; done > $O/stage_prof_final.txt; cat $O/stage_prof_final.txt
rm -f $O/r05_rows_parity_raw.txt
FISHRT_PARITY_LOG=$O/r05_rows_parity_raw.txt python -m pytest tests/test_kv_forced_gpu.py -q --timeout 900 2>&1 | tail -2
python tools/check_roofline.py $O/bench_line.json $O/bench_kernel_stats.csv $O/pmc_hbm_traffic.json $O/batch_kernel_stats_F32.csv $O/batch_kernel_stats_F96.csv \
  $O/pmc_batch_traffic.json > $O/check_roofline.txt 2>&1; tail -45 $O/check_roofline.txt
