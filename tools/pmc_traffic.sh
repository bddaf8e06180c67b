#!/bin/bash
# HBM bytes per decode frame (MI355X_MICROARCH.md HBM / rocprofv3 recipe): two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace
# only) over one configs[1] request of 64 frames (tools/pmc_run.py), for the persistent path (default) and the per-node path (--no-persistent); tools/pmc_traffic.py turns the
# counter CSVs into profiles/r02_pmc_hbm_traffic.json (FETCH_SIZE doubled: the gfx950 correction for wide coalesced reads).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_*
for mode in persistent per_node; do
  flag=1; [ $mode = per_node ] && flag=0
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${mode}_$ctr -o c -- \
      python $GRAFT_REPO_ROOT/tools/pmc_run.py 64 $flag > $O/pmc_${mode}_$ctr.log 2>&1
    echo "$mode $ctr rc=$?"
  done
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp $O/pmc_hbm_traffic.json 64
cat $O/pmc_hbm_traffic.json
