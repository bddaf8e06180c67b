"""Summarise the rocprofv3 --pmc passes of tools/pmc_batch.sh: HBM bytes per decode step = (traffic of the run with F2 frames - traffic of
the run with F1 frames) / extra steps, traffic = sum over ALL kernels of FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md
HBM) + WRITE_SIZE, both KiB.  usage: pmc_batch.py <dir with pmcb_*> <out.json> [<earlier summary to merge>]"""
import csv, glob, json, os, subprocess, sys

root, out_path = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "fish-speech.rs_amd"))
import bench
from fishrt import config as fcfg
cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS


def total(d, name):
    t, n = 0.0, 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                t += float(r["Counter_Value"]); n += 1
    return t, n


def per_kernel(d, name):
    import collections, re
    t = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                t[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void fs::", "").replace("fs::", "")] += float(r["Counter_Value"])
    return t


def kernel_split(tag, F1, F2, steps):
    """per kernel: (fetch x 2, write) bytes per step, differenced between the two runs"""
    f1, f2 = per_kernel(f"{root}/pmcb_{tag}_{F1}_FETCH_SIZE", "FETCH_SIZE"), per_kernel(f"{root}/pmcb_{tag}_{F2}_FETCH_SIZE", "FETCH_SIZE")
    w1, w2 = per_kernel(f"{root}/pmcb_{tag}_{F1}_WRITE_SIZE", "WRITE_SIZE"), per_kernel(f"{root}/pmcb_{tag}_{F2}_WRITE_SIZE", "WRITE_SIZE")
    out = {}
    for k in f2:
        fb, wb = (f2[k] - f1.get(k, 0.0)) * 2048 / steps, (w2.get(k, 0.0) - w1.get(k, 0.0)) * 1024 / steps
        if fb + wb > 1e6:
            out[k] = {"fetch_bytes_x2": int(fb), "write_bytes": int(wb)}
    return dict(sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_x2"] + kv[1]["write_bytes"])))


def run_bytes(tag, F):
    f, nf = total(f"{root}/pmcb_{tag}_{F}_FETCH_SIZE", "FETCH_SIZE")
    w, nw = total(f"{root}/pmcb_{tag}_{F}_WRITE_SIZE", "WRITE_SIZE")
    return (f * 1024 * 2 + w * 1024, nf) if nf and nw else (None, 0)


res = {}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):   # keys of an earlier summary that this run did not re-measure are kept (with their commit)
    old = json.load(open(sys.argv[3]))
    for k, v in old.items():
        if isinstance(v, dict):
            v.setdefault("commit", old.get("commit", "?")); res[k] = v
try:
    res["commit"] = os.environ.get("FISHRT_COMMIT") or subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=sys.path[0]).decode().strip()
except Exception:
    res["commit"] = "unknown"
F1, F2 = 32, 96
corr = "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024, all kernels of the run; (longer run) - (shorter run)"
# B = 32 static batch: one job of F / 2 frames; prompts U{64..384} seed 77 padded to Lmax
import numpy as np
rng = np.random.RandomState(77)
Lmax = int(rng.randint(64, 385, 32).max())
b1, n1 = run_bytes("batch32", F1); b2, n2 = run_bytes("batch32", F2)
if b1 and b2:
    steps = (F2 - F1) // 2   # tools/pmc_batch.sh runs the batch job once at F / 2 frames
    Tavg = Lmax + (F1 + F2) / 4
    res["static_batch32"] = {"hbm_bytes_per_step": int((b2 - b1) / steps), "algorithmic_bytes_per_step": int(bench.frame_bytes(cfg, tok, 0) + 32 * 12288 * Tavg),
                             "commit": res["commit"], "kv_len_avg": Tavg, "steps_differenced": steps, "dispatches": [n1, n2],
                             "kernels_bytes_per_step": kernel_split("batch32", F1, F2, steps),
                             "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/batch_bench.py 32 {16,48} 1", "corrections": corr}
L = bench.default_voice_prompt(tok).shape[1]
for R in (4, 8):
    b1, n1 = run_bytes(f"rows{R}", F1); b2, n2 = run_bytes(f"rows{R}", F2)
    if b1 and b2:
        steps = F2 - F1
        Tavg = L + (F1 + F2) / 2
        # one R-row frame: the weights once + every row's KV
        res[f"rows_R{R}"] = {"hbm_bytes_per_frame": int((b2 - b1) / steps), "algorithmic_bytes_per_frame": int(bench.frame_bytes(cfg, tok, 0) + R * 12288 * Tavg),
                             "commit": res["commit"], "kv_len_avg": Tavg, "frames_differenced": steps, "dispatches": [n1, n2],
                             "kernels_bytes_per_frame": kernel_split(f"rows{R}", F1, F2, steps),
                             "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/pmc_rows_run.py {R} {{32,96}}", "corrections": corr}
json.dump(res, open(out_path, "w"), indent=1)
