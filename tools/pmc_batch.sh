#!/bin/bash
# HBM bytes per B = 32 static-batch decode step and per R-row persistent frame (MI355X_MICROARCH.md HBM / rocprofv3 recipe): separate
# --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel-trace only) over the SAME command at two frame counts; the difference divided by the
# extra steps is the per-step traffic (prefill, weight load and packing kernels cancel).  tools/pmc_batch.py writes the summary.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmcb_*
for ctr in FETCH_SIZE WRITE_SIZE; do
  for F in 32 96; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_batch32_${F}_$ctr -o c -- \
      python $GRAFT_REPO_ROOT/tools/batch_bench.py 32 $F > $O/pmcb_batch32_${F}_$ctr.log 2>&1
    echo "batch32 $F $ctr rc=$?"
    for R in 4 8; do
      timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_rows${R}_${F}_$ctr -o c -- \
        python $GRAFT_REPO_ROOT/tools/pmc_rows_run.py $R $F > $O/pmcb_rows${R}_${F}_$ctr.log 2>&1
      echo "rows$R $F $ctr rc=$?"
    done
  done
done
python3 $GRAFT_REPO_ROOT/tools/pmc_batch.py /tmp $O/pmc_batch_traffic.json
cat $O/pmc_batch_traffic.json
