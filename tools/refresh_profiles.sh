#!/bin/bash
# Runs on the MI355X box: GPU test suite, the bench line, the rocprofv3 kernel-trace summary of the bench command and the row-path
# micro-benchmarks; everything lands in gpurun_out/ (copied into profiles/ afterwards).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_all.log 2>&1; tail -2 $O/gpu_all.log
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
(timeout 100 tools/ubench_gemm.bin 32 4 0; timeout 100 tools/ubench_gemm.bin 256 4 0; timeout 100 tools/ubench_gemm.bin 382 4 1 382; timeout 100 tools/ubench_gemm.bin 1910 4 1 382) > $O/ubench_gemm.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
find /tmp/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -12 $O/bench_kernel_stats.csv | cut -c1-160
