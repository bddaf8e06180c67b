#!/bin/bash
# HBM bytes per launch of the batch-1 decode kernels (MI355X_MICROARCH.md HBM/rocprofv3 recipe): two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; kernel-trace only) over tools/ubench_lm.bin, medians per kernel, FETCH_SIZE doubled (gfx950 correction).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_f /tmp/pmc_w
UBENCH_QUICK=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- $GRAFT_REPO_ROOT/tools/ubench_lm.bin 495 > $O/pmc_f.log 2>&1
UBENCH_QUICK=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- $GRAFT_REPO_ROOT/tools/ubench_lm.bin 495 > $O/pmc_w.log 2>&1
python3 - <<'PY' > $O/pmc_hbm_traffic.csv
import csv, glob, statistics, re, collections
def load(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                out[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out
F, W = load("/tmp/pmc_f", "FETCH_SIZE"), load("/tmp/pmc_w", "WRITE_SIZE")
print("kernel,dispatches,FETCH_SIZE_KB_median,fetch_bytes_corrected_x2,WRITE_SIZE_KB_median")
for k in sorted(F):
    fm = statistics.median(F[k]); wm = statistics.median(W.get(k, [0.0]))
    print(f'"{k}",{len(F[k])},{fm:.1f},{int(fm * 1024 * 2)},{wm:.1f}')
PY
cat $O/pmc_hbm_traffic.csv
