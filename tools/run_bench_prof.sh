cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
F=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp $F $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.csv; head -8 $F; else tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof2.log; fi
tail -c 400 $GRAFT_REPO_ROOT/gpurun_out/bench_line.json
