#!/bin/bash
# per-(kernel, grid) time of one FireflyCodec.decode (T = 64 and 256) under rocprofv3 --kernel-trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/profv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/profv -o voc -- python $GRAFT_REPO_ROOT/tools/vocoder_bench.py > $O/prof_voc.log 2>&1
F=$(find /tmp/profv -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/voc_calls.py $F $O/voc_calls.txt
head -40 $O/voc_calls.txt
