#!/bin/bash
# One round's evidence in one gpurun call:   tools/refresh.sh NN          (on the GPU box, e.g. `gpurun -- 'bash tools/refresh.sh 06'`)
# then, back in the repository:              tools/refresh.sh NN install  (copies what is judged from gpurun_out/ to profiles/rNN_*)
# (replaces the per-round refresh_r03 / r04 / r05 + install_r05 scripts.)  Everything lands in gpurun_out/ (scratch).  Order: PMC traffic
# (batch-1 frame; B = 32 step and R-row frames differenced) FIRST, installed as profiles/rNN_* on the box so that the bench line that
# follows cites them; full bench line; rocprofv3 kernel stats of the same command; B = 32 step kernel stats at two frame counts;
# request-row kernel stats; configs[4] long-form run + 4 concurrent fp8 streams; per-stage profile; parity log; roofline re-check.
# The commit: .git does not travel to the GPU box, so the caller writes `git rev-parse --short HEAD` to tools/_ab/commit.txt before the
# gpurun call (tools/_ab/ is git-ignored scratch that does travel); it is exported as FISHRT_COMMIT, which tools/pmc_traffic.py and
# tools/pmc_batch.py record inside every PMC summary.
set -u
NN=${1:?round number, e.g. 06}; R=r$NN
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if [ "${2:-}" = install ]; then
  cd "$HERE"; O=gpurun_out; P=profiles
  for f in bench_line.json bench_kernel_stats.csv check_roofline.txt pmc_hbm_traffic.json pmc_batch_traffic.json batch_kernel_stats_F32.csv \
           batch_kernel_stats_F96.csv rows_kernel_stats_R4.csv rows_kernel_stats_R8.csv longform_fp8.txt; do
    [ -s $O/$f ] && cp $O/$f $P/${R}_$f
  done
  [ -s $O/commit.txt ] && cp $O/commit.txt $P/${R}_commit.txt
  [ -s $O/rows_parity_raw.txt ] && ( echo "Round $NN: max |dlogit| of every shipped instantiation of the persistent kernels against the CPU oracle under the KV-forced protocol"
    echo "(tests/test_kv_forced_gpu.py with FISHRT_PARITY_LOG: the oracle teacher-forced on the GPU's tokens, attending over the GPU's own cached K/V rows --"
    echo "slow layers, the current step's row included, and the fast decoder's per-pass rows; tolerance 2e-4 slow / fast; 'units' = bf16 ulps floored at 2^-17)."
    echo; cat $O/rows_parity_raw.txt ) > $P/${R}_rows_parity.txt
  [ -s $O/stage_prof_final.txt ] && ( echo "== per-stage profile at the evidence commit (tools/refresh.sh $NN; FISHRT_PERSIST_PROF=1, workgroups 0 and 255) =="; cat $O/stage_prof_final.txt ) > $P/${R}_stage_profile.txt
  ls -la $P/${R}_*
  exit 0
fi
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export FISHRT_COMMIT=$(cat tools/_ab/commit.txt 2>/dev/null || echo unknown)
echo $FISHRT_COMMIT > $O/commit.txt
PREV=$(ls profiles/r*_pmc_batch_traffic.json 2>/dev/null | sort | tail -1 | xargs -r basename)
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
PMC_BATCH_PREV=${PREV:-r05_pmc_batch_traffic.json} bash tools/pmc_batch.sh > $O/pmc_batch.log 2>&1; tail -3 $O/pmc_batch.log
[ -s $O/pmc_hbm_traffic.json ] && cp $O/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
[ -s $O/pmc_batch_traffic.json ] && cp $O/pmc_batch_traffic.json profiles/${R}_pmc_batch_traffic.json
bash tools/run_bench_prof.sh > $O/run_bench_prof.log 2>&1
for F in 32 96; do bash tools/prof_batch.sh 32 $F > $O/prof_batch_F$F.txt 2>&1; cp $O/batch_kernel_stats.csv $O/batch_kernel_stats_F$F.csv; done
export TMPDIR=/tmp; cd /tmp
for RR in 4 8; do
  rm -rf /tmp/profr$RR
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profr$RR -o r -- python $GRAFT_REPO_ROOT/tools/pmc_rows_run.py $RR 256 > $O/prof_rows_R$RR.log 2>&1
  F=$(find /tmp/profr$RR -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/rows_kernel_stats_R$RR.csv && head -4 $F
done
cd $GRAFT_REPO_ROOT
timeout 900 python tools/longform_bench.py 4096 fp8 f16 x 4 1024 2>&1 | tail -3 > $O/longform_fp8.txt; tail -4 $O/longform_fp8.txt
for wg in 0 255; do FISHRT_PERSIST_PROF_WG=$wg FISHRT_PERSIST_PROF=1 python tools/p2_quick.py bf16 2>&1 | tail -3; done > $O/stage_prof_final.txt; cat $O/stage_prof_final.txt
rm -f $O/rows_parity_raw.txt
FISHRT_PARITY_LOG=$O/rows_parity_raw.txt python -m pytest tests/test_kv_forced_gpu.py -q --timeout 900 2>&1 | tail -2
python tools/check_roofline.py $O/bench_line.json $O/bench_kernel_stats.csv $O/pmc_hbm_traffic.json $O/batch_kernel_stats_F32.csv $O/batch_kernel_stats_F96.csv \
  $O/pmc_batch_traffic.json > $O/check_roofline.txt 2>&1; tail -45 $O/check_roofline.txt
