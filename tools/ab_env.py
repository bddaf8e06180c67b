"""A/B of the batch-1 persistent path under environment knobs, one process per variant, several repetitions interleaved (box-to-box and
run-to-run noise is ~1 %: compare medians of interleaved runs).  usage: ab_env.py REPS name1:K=V,K2=V2 name2: ...   (empty = defaults)"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, _, kv = spec.partition(":")
    env = dict(kv_.split("=", 1) for kv_ in kv.split(",") if kv_)
    variants.append((name, env))
dtype = os.environ.get("AB_DTYPE", "bf16")
res = {n: [] for n, _ in variants}
for r in range(reps):
    for name, env in variants:
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2_quick.py"), dtype], capture_output=True, text=True, env=e).stdout.strip().splitlines()[-1]
        us = float(out.split("greedy")[1].split("us/frame")[0])
        res[name].append((us, out.split("crc")[1].split()[0]))
for name, _ in variants:
    v = [u for u, _ in res[name]]
    print(f"{name:28s} median {statistics.median(v):7.1f}  min {min(v):7.1f}  max {max(v):7.1f}  us/frame   crc {sorted(set(c for _, c in res[name]))}")
