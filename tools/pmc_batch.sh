#!/bin/bash
# HBM bytes per B = 32 static-batch decode step and per R-row persistent frame (MI355X_MICROARCH.md HBM / rocprofv3 recipe): separate
# --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel-trace only) over the SAME command at two frame counts; the difference divided by the
# extra steps is the per-step traffic (prefill, weight load and packing kernels cancel).  tools/pmc_batch.py writes the summary.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmcb_*
TAGS=${1:-"batch32 rows4 rows8"}
run() {  # a counter pass that hangs (seen once in 12 passes) is killed after 150 s and repeated
  local out=$1; shift
  for try in 1 2 3; do rm -rf $out; timeout 150 rocprofv3 "$@" && return 0; echo "retry $try: $out"; done; return 1
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  for F in 32 96; do
    # (the batch step is 373 dispatches: rocprofv3's counter collection crashed repeatably past ~50k dispatches, so 16 / 48 frames, one job)
    [[ $TAGS == *batch32* ]] && run /tmp/pmcb_batch32_${F}_$ctr --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_batch32_${F}_$ctr -o c -- \
      python $GRAFT_REPO_ROOT/tools/batch_bench.py 32 $((F / 2)) 1 > $O/pmcb_batch32_${F}_$ctr.log 2>&1
    echo "batch32 $F $ctr rc=$?"
    for R in 4 8; do
      [[ $TAGS == *rows$R* ]] && run /tmp/pmcb_rows${R}_${F}_$ctr --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_rows${R}_${F}_$ctr -o c -- \
        python $GRAFT_REPO_ROOT/tools/pmc_rows_run.py $R $F > $O/pmcb_rows${R}_${F}_$ctr.log 2>&1
      echo "rows$R $F $ctr rc=$?"
    done
  done
done
python3 $GRAFT_REPO_ROOT/tools/pmc_batch.py /tmp $O/pmc_batch_traffic.json $GRAFT_REPO_ROOT/profiles/${PMC_BATCH_PREV:-r04_pmc_batch_traffic.json}
cat $O/pmc_batch_traffic.json
