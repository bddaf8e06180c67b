"""Coordinate descent over the pre-sweep naps of the request-row persistent kernels (FISHRT_NAPS_ROWS_FAST / FISHRT_NAPS_ROWS_SLOW, 64-clock
units per stage kind) on n concurrent configs[1]-like requests; prints the decode us per n-row frame after every improving move.
usage: tune_naps_rows.py [n_rows] [start naps, 12 comma-separated: fast S1 S2 S3 S4 head decision, slow S1 S2 S3 S4 S5 head]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(4321)
F = 96
prompts, mnt = [], []
for i in range(n):
    L = 360 + 7 * i
    p = np.zeros((9, L), np.uint32); p[0] = rng.randint(0, 100000, L)
    prompts.append(p); mnt.append(L + F - 2)


def measure(naps):
    os.environ["FISHRT_NAPS_ROWS_FAST"] = ",".join(str(v) for v in naps[:6])
    os.environ["FISHRT_NAPS_ROWS_SLOW"] = ",".join(str(v) for v in naps[6:])
    best = 1e9
    for _ in range(2):
        lm.generate_multi(prompts, mnt, temp=0.0, repetition_penalty=1.2, ignore_eos=True)
        best = min(best, lm.last_stats()["decode_ms"] * 1e3 / (F - 1))
    return best


naps = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else [16, 16, 20, 20, 20, 12, 24, 0, 8, 40, 32, 12])]
cur = measure(naps)
print(f"start {naps}: {cur:.1f} us per {n}-row frame", flush=True)
names = ["f.S1", "f.S2", "f.S3", "f.S4", "f.head", "f.dec", "s.S1", "s.S2", "s.S3", "s.S4", "s.S5", "s.head"]
for sweep in range(3):
    improved = False
    for i in range(12):
        for v in sorted(set([max(0, naps[i] - 8), max(0, naps[i] - 4), naps[i] + 4, naps[i] + 8, naps[i] + 16, 0, 2])):
            if v == naps[i]:
                continue
            t = measure(naps[:i] + [v] + naps[i + 1:])
            if t < cur - 0.5:
                cur, naps[i], improved = t, v, True
                print(f"  {names[i]} = {v}: {cur:.1f}  {naps}", flush=True)
    if not improved:
        break
print(f"final {naps}: {cur:.1f} us per {n}-row frame")
print("FAST", ",".join(str(v) for v in naps[:6]), "SLOW", ",".join(str(v) for v in naps[6:]))
