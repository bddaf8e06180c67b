"""A/B of library builds under tools/_ab/lib<name>.so (scratch, git-ignored) on ONE box: runs are interleaved (box-to-box spread is ~2 %, run-to-run
~0.3 %), each run a fresh process after copying the variant over fish-speech.rs_amd/libfishrt.so.  usage: ab_lib.py REPS name1 name2 ... [-- ENV=V ...]"""
import os, shutil, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
env_extra = {}
if "--" in args:
    k = args.index("--")
    env_extra = dict(a.split("=", 1) for a in args[k + 1:])
    args = args[:k]
reps, names = int(args[0]), args[1:]
dtype = os.environ.get("AB_DTYPE", "bf16")
live = os.path.join(ROOT, "fish-speech.rs_amd", "libfishrt.so")
keep = live + ".ab_keep"
shutil.copy(live, keep)
res = {n: [] for n in names}
try:
    for r in range(reps):
        for n in names:
            shutil.copy(os.path.join(ROOT, "tools", "_ab", f"lib{n}.so"), live)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2_quick.py"), dtype], capture_output=True, text=True, env=dict(os.environ, **env_extra)).stdout.strip().splitlines()[-1]
            res[n].append((float(out.split("greedy")[1].split("us/frame")[0]), out.split("crc")[1].split()[0]))
finally:
    shutil.move(keep, live)
for n in names:
    v = [u for u, _ in res[n]]
    print(f"[{dtype}] {n:20s} median {statistics.median(v):7.1f}  min {min(v):7.1f}  max {max(v):7.1f}  us/frame   crc {sorted(set(c for _, c in res[n]))}")
