#!/bin/bash
# per-kernel time of the B = 32 static-batch decode step (rocprofv3 --kernel-trace --stats on tools/batch_bench.py 32 64)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/profb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -o b -- python $GRAFT_REPO_ROOT/tools/batch_bench.py ${1:-32} ${2:-64} > $O/prof_batch.log 2>&1
F=$(find /tmp/profb -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp $F $O/batch_kernel_stats.csv; python3 - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-90s calls %6s avg %8.1f us  %5.1f%%" % (r["Name"].replace("void fs::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
else tail -5 $O/prof_batch.log; fi
tail -2 $O/prof_batch.log
