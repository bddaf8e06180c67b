#!/bin/bash
# A/B of the B = 32 static-batch decode step under environment knobs (interleaved repetitions of tools/batch_bench.py).
# usage: ab_batch.sh REPS "name1:K=V K2=V2" "name2:" ...
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
REPS=$1; shift
for r in $(seq 1 $REPS); do
  for spec in "$@"; do
    name="${spec%%:*}"; kv="${spec#*:}"
    us=$(env $kv python tools/batch_bench.py ${AB_B:-32} ${AB_FRAMES:-256} 2 2>&1 | tail -1 | sed -n 's/.* \([0-9]*\) us\/step.*/\1/p')
    echo "$name rep$r $us us/step"
  done
done
