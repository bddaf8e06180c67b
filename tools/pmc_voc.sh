#!/bin/bash
# matrix-core utilisation and HBM traffic of the vocoder's conv kernels (decode, T = 256, bf16x3 mode): separate --pmc passes, kernel-trace only
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -f $O/pmc_voc.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf /tmp/pv_$n
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pv_$n -o v -- python $GRAFT_REPO_ROOT/tools/vocoder_time.py 256 bf16x3 > $O/pmc_voc_$n.log 2>&1
  python3 - "$n" <<'PY' >> $O/pmc_voc.txt
import csv, glob, sys, collections
n = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"/tmp/pv_{n}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void fs::", "")
        if "conv1d" in k or "mean3" in k or "act_split" in k:
            key = k + " grid " + r.get("Grid_Size", "?")
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key + r["Counter_Name"]] += 1
for k, d in sorted(acc.items()):
    print(k, {c: v for c, v in d.items()}, "dispatches", max(cnt[k + c] for c in d))
PY
done
cat $O/pmc_voc.txt
