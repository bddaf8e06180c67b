"""Recompute every number of bench.py's `roofline` object from committed evidence and compare:
  usage: check_roofline.py <bench_line.json> <rocprof kernel_stats.csv> [<pmc_hbm_traffic.json>]
  * frame: algorithmic bytes / frame_us / 8 TB/s == roofline.frac; B_min likewise == frac_min
  * per kernel: rocprofv3 --kernel-trace --stats average duration of k_slow_persist / k_fast_persist vs the HIP-event averages
    bench.py measured live (roofline.kernels[*].avg_us), and the fractions recomputed from the rocprof figure
  * traffic: the PMC summary's bytes per frame == roofline.traffic"""
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
print(f"   sum of the two kernels {ssum:.1f} us vs frame {r['frame_us']:.1f} us (the difference = the two launch boundaries of a replay)")
ok &= ssum <= r["frame_us"] * 1.02
if len(sys.argv) > 3:
    pmc = json.load(open(sys.argv[3]))
    check("traffic (PMC bytes per frame)", r["traffic"], pmc["persistent"]["hbm_bytes_per_frame"], 1e-9)
print("ROOFLINE CHECK", "PASSED" if ok else "FAILED")
sys.exit(0 if ok else 1)
