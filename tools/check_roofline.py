"""Recompute every number of bench.py's `roofline` object from committed evidence and compare:
  usage: check_roofline.py <bench_line.json> <rocprof kernel_stats.csv> [<pmc_hbm_traffic.json>]
  * frame: algorithmic bytes / frame_us / 8 TB/s == roofline.frac; B_min likewise == frac_min
  * per kernel: rocprofv3 --kernel-trace --stats average duration of k_slow_persist / k_fast_persist vs the HIP-event averages
    bench.py measured live (roofline.kernels[*].avg_us), and the fractions recomputed from the rocprof figure
  * traffic: the PMC summary's bytes per frame == roofline.traffic
  * extras (when the line has them): every roofline_frac of static_batch32 / _fp8 / 256 and of persistent_rows.R* recomputed from the
    SURVEY.md 8d formula and the line's own step_us / frame_us
  usage (round 4): check_roofline.py <bench_line.json> <kernel_stats.csv> [<pmc_hbm_traffic.json> [<batch_stats_F32.csv> <batch_stats_F96.csv>
  [<pmc_batch_traffic.json>]]]
  * batch step: rocprofv3 kernel time of the B = 32 decode step = (total kernel time of tools/batch_bench.py 32 96) - (... 32 32), / the 128
    extra steps, vs extras.static_batch32.step_us (HIP events); the per-kernel split of the step is printed
  * batch traffic: extras.static_batch32.traffic / persistent_rows.R*.traffic == the tracked PMC summary"""
import csv, json, sys

sys.path.insert(0, __file__.rsplit("/", 2)[0]); sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/fish-speech.rs_amd")
import bench
from fishrt import config as fcfg

line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = line["roofline"]
cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
T = r["kv_len_avg"]
bf = bench.frame_bytes(cfg, tok, T)
b_slow, b_fast, b_fast_min = bench.frame_bytes_split(cfg, tok, T)
ok = True


def check(what, got, exp, tol):
    global ok
    good = abs(got - exp) <= tol * max(abs(exp), 1e-12)
    ok &= good
    print(f"{'ok ' if good else 'BAD'} {what}: bench line {got:.6g}  recomputed {exp:.6g}")


assert abs(bf - (b_slow + b_fast)) < 1, "the per-launch split must add up to the frame formula"
check("algorithmic bytes per frame", r["algorithmic_bytes_per_frame"], bf, 1e-9)
check("frame frac", r["frac"], bf / (r["frame_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
check("frame frac_min", r["frac_min"], (b_slow + b_fast_min) / (r["frame_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
check("frames/s from frame_us", line["decode_frames_per_s_per_gpu"], 1e6 / r["frame_us"], 1e-3)
stats = {}
for row in csv.DictReader(open(sys.argv[2])):
    stats[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
for name, nb, nbm in (("k_slow_persist", b_slow, b_slow), ("k_fast_persist", b_fast, b_fast_min)):
    k = r["kernels"][name]
    rp = [v for n, v in stats.items() if name in n]
    assert rp, f"{name} not in the rocprof stats"
    calls, avg = max(rp)
    print(f"   {name}: rocprofv3 {calls} calls, avg {avg:.1f} us; HIP events in bench.py {k['avg_us']:.1f} us")
    check(f"{name} avg_us (HIP events vs rocprofv3, 4 % band)", k["avg_us"], avg, 0.04)
    check(f"{name} algorithmic bytes", k["algorithmic_bytes_per_launch"], nb, 1e-9)
    check(f"{name} frac", k["frac"], nb / (k["avg_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
    check(f"{name} frac_min", k["frac_min"], nbm / (k["avg_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
ssum = sum(r["kernels"][n]["avg_us"] for n in r["kernels"])
print(f"   sum of the two kernels {ssum:.1f} us vs frame {r['frame_us']:.1f} us (per-kernel figures: direct launches with a HIP event in front of / between / behind them; frame: the event-free replay of 8-frame graphs)")
ok &= ssum <= r["frame_us"] * 1.02
if len(sys.argv) > 3:
    pmc = json.load(open(sys.argv[3]))
    check("traffic (PMC bytes per frame)", r["traffic"], pmc["persistent"]["hbm_bytes_per_frame"], 1e-9)
ex = line.get("extras", {})
if ex:
    import numpy as np
    prompts = bench.config2_prompts(tok, 256)
    for name, wb, B in (("static_batch32", 2, 32), ("static_batch32_fp8", 1, 32), ("static_batch256", 2, 256)):
        if name in ex:
            Lmax = max(p.shape[1] for p in prompts[:B])
            bs = bench.frame_bytes(cfg, tok, 0, wb) + B * 12288 * (Lmax + 256 / 2)
            check(f"extras.{name}.roofline_frac", ex[name]["roofline_frac"], bs / (ex[name]["step_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
            check(f"extras.{name}.decode_frames_per_s", ex[name]["decode_frames_per_s"], B * 1e6 / ex[name]["step_us"], 2e-3)
    pr = ex.get("persistent_rows", {})
    Lp = bench.default_voice_prompt(tok).shape[1]
    for key, v in pr.items():
        if key[0] == "R" and "frame_us" in v:
            R = int(key[1])
            bs = bench.frame_bytes(cfg, tok, 0, 1 if key.endswith("_fp8") else 2) + R * 12288 * (Lp + 256 / 2.0)
            check(f"extras.persistent_rows.{key}.roofline_frac", v["roofline_frac"], bs / (v["frame_us"] * 1e-6) / bench.HBM_PEAK, 2e-3)
            check(f"extras.persistent_rows.{key}.decode_frames_per_s", v["decode_frames_per_s"], R * 1e6 / v["frame_us"], 2e-3)
    for key in ("batch1_sampled", "batch1_fp8"):
        if key in ex and "roofline_frac" in ex[key] and "frame_us" in ex[key]:
            wb = 1 if "fp8" in key else 2
            check(f"extras.{key}.roofline_frac", ex[key]["roofline_frac"], bench.frame_bytes(cfg, tok, T, wb) / (ex[key]["frame_us"] * 1e-6) / bench.HBM_PEAK, 1e-2)
if len(sys.argv) > 5 and ex:
    def load(p):
        return {row["Name"]: float(row["TotalDurationNs"]) for row in csv.DictReader(open(p))}
    a, b = load(sys.argv[4]), load(sys.argv[5])
    steps = 2 * (96 - 32)
    per = {k: (b[k] - a.get(k, 0.0)) / steps / 1e3 for k in b}
    tot = sum(per.values())
    print(f"   B = 32 decode step, rocprofv3 kernel time differenced over {steps} steps (KV window Lmax + 32..96): {tot:.1f} us of kernels per step")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]:
        print(f"      {v:8.1f} us/step  {k.replace('void fs::', '')[:110]}")
    # ~300 dispatches per step (373 before round 5's folded step): the profiler's per-dispatch timestamping adds 0.3-0.9 us to every node's duration (box to box: the sum came out
    # 7 % and 17 % above the un-profiled HIP-event step time in two runs), so this is a sanity band, not a timing claim -- the step time of
    # record is the HIP-event one; what the table above is for is the SPLIT of the step over its kernels
    check("extras.static_batch32.step_us (HIP events) vs rocprofv3 kernel time per step (profiler-inflated; 25 % sanity band)", ex["static_batch32"]["step_us"], tot, 0.25)
if len(sys.argv) > 6 and ex:
    pb = json.load(open(sys.argv[6]))
    if "static_batch32" in pb:
        check("extras.static_batch32.traffic (PMC bytes per step)", ex["static_batch32"].get("traffic") or 0, pb["static_batch32"]["hbm_bytes_per_step"], 1e-9)
        print(f"      traffic / algorithmic bytes of the same KV window: {pb['static_batch32']['hbm_bytes_per_step'] / pb['static_batch32']['algorithmic_bytes_per_step']:.3f}")
    for R in (4, 8):
        k = f"rows_R{R}"
        if k in pb and f"R{R}" in ex.get("persistent_rows", {}):
            check(f"extras.persistent_rows.R{R}.traffic (PMC bytes per frame)", ex["persistent_rows"][f"R{R}"].get("traffic") or 0, pb[k]["hbm_bytes_per_frame"], 1e-9)
            print(f"      traffic / algorithmic bytes of the same KV window: {pb[k]['hbm_bytes_per_frame'] / pb[k]['algorithmic_bytes_per_frame']:.3f}")
print("ROOFLINE CHECK", "PASSED" if ok else "FAILED")
sys.exit(0 if ok else 1)
