#!/bin/bash
# matrix-core utilisation and HBM traffic of the vocoder's conv kernels (decode, T = 256, VOC_PREC = f16 | bf16x3): separate --pmc passes,
# kernel-trace only; prints one table: kernel, grid threads, dispatches, mfma% = MFMA_BUSY / (4 SIMDs x BUSY_CU), fetch / write MB per dispatch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
P=${VOC_PREC:-f16}
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pv_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pv_$i -o v -- python $GRAFT_REPO_ROOT/tools/vocoder_time.py 256 $P > $O/pmc_voc_$i.log 2>&1
  i=$((i+1))
done
python3 - "$P" <<'PY' > $O/pmc_voc_$P.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for i in range(3):
    for f in glob.glob(f"/tmp/pv_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void fs::", "")
            if "conv1d" in k or "respair" in k or "mean3" in k or "act_split" in k or "dwconv" in k:
                key = (k, int(r.get("Grid_Size", "0")))
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
print(f"# tools/pmc_voc.sh (VOC_PREC={sys.argv[1]}): three separate passes of rocprofv3 --pmc <set> --kernel-trace -- python tools/vocoder_time.py 256 {sys.argv[1]}")
print("# sets: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES | FETCH_SIZE | WRITE_SIZE; FETCH_SIZE / WRITE_SIZE in KiB as counted (the gfx950 correction doubles FETCH_SIZE for wide coalesced reads)")
print("%-52s %9s %5s %7s %14s %14s" % ("kernel", "threads", "n", "mfma%", "fetch MB/disp", "write MB/disp"))
for key in sorted(acc):
    d, c = acc[key], cnt[key]
    n = max(c.values())
    mf = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(4.0 * d.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0)
    print("%-52s %9d %5d %6.1f%% %14.1f %14.1f" % (key[0][:52], key[1], n, mf, d.get("FETCH_SIZE", 0.0) / max(c.get("FETCH_SIZE", 1), 1) / 1024, d.get("WRITE_SIZE", 0.0) / max(c.get("WRITE_SIZE", 1), 1) / 1024))
PY
cat $O/pmc_voc_$P.txt
