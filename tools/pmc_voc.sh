#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*(LDS|MFMA|WAIT|BUSY_CU|WAVE_CYC)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/pmc_avail.txt
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf /tmp/pv_$n
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pv_$n -o v -- python $GRAFT_REPO_ROOT/tools/vocoder_bench.py > $O/pmc_voc_$n.log 2>&1
  python3 - "$n" <<'PY' >> $O/pmc_voc.txt
import csv, glob, sys, collections
n = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"/tmp/pv_{n}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void fs::", "")
        if "conv1d_mfma" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items(): print(k, dict(d))
PY
done
cat $O/pmc_voc.txt; echo; cut -c1-1500 $O/pmc_avail.txt
