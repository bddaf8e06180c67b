#!/bin/bash
# Round-3 evidence in one gpurun call: full bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, roofline re-check.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/run_bench_prof.sh > gpurun_out/run_bench_prof.log 2>&1
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
tail -3 gpurun_out/pmc_traffic.log
python tools/check_roofline.py gpurun_out/bench_line.json gpurun_out/bench_kernel_stats.csv > gpurun_out/check_roofline.txt 2>&1; tail -25 gpurun_out/check_roofline.txt
