#!/bin/bash
# copy the judged summaries of the last tools/refresh_r05.sh run from gpurun_out/ (scratch) to profiles/r05_* (tracked)
cd "$(dirname "$0")/.."; O=gpurun_out; P=profiles
cp $O/bench_line.json $P/r05_bench_line.json
cp $O/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
cp $O/check_roofline.txt $P/r05_check_roofline.txt
cp $O/pmc_hbm_traffic.json $P/r05_pmc_hbm_traffic.json
cp $O/pmc_batch_traffic.json $P/r05_pmc_batch_traffic.json
cp $O/batch_kernel_stats_F32.csv $P/r05_batch_kernel_stats_F32.csv
cp $O/batch_kernel_stats_F96.csv $P/r05_batch_kernel_stats_F96.csv
cp $O/rows_kernel_stats_R4.csv $P/r05_rows_kernel_stats_R4.csv
cp $O/rows_kernel_stats_R8.csv $P/r05_rows_kernel_stats_R8.csv
cp $O/longform_fp8.txt $P/r05_longform_fp8.txt
( echo "Round 5: max |dlogit| of every shipped instantiation of the persistent kernels against the CPU oracle under the KV-forced protocol"
  echo "(tests/test_kv_forced_gpu.py with FISHRT_PARITY_LOG: the oracle teacher-forced on the GPU's tokens, attending over the GPU's own cached K/V rows --"
  echo "slow layers, the current step's row included, and the fast decoder's per-pass rows; tolerance 2e-4 slow / fast; 'units' = bf16 ulps floored at 2^-17)."
  echo; cat $O/r05_rows_parity_raw.txt ) > $P/r05_rows_parity.txt
( echo "== final per-stage profile at the evidence commit (tools/refresh_r05.sh) =="; cat $O/stage_prof_final.txt ) > $P/r05_stage_profile_final.txt
ls -la $P/r05_*
