#!/bin/bash
# per-(kernel, grid) time of FireflyCodec.decode at T = 256 under rocprofv3 --kernel-trace: tools/prof_voc.sh [precision]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
P=${1:-bf16x3}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/profv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/profv -o voc -- python $GRAFT_REPO_ROOT/tools/vocoder_time.py 256 $P > $O/prof_voc.log 2>&1
F=$(find /tmp/profv -name "*kernel_trace.csv" | head -1)
if [ -n "$F" ]; then python3 $GRAFT_REPO_ROOT/tools/voc_calls.py $F $O/voc_calls_$P.txt; head -45 $O/voc_calls_$P.txt; else tail -20 $O/prof_voc.log; fi
