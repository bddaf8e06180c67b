"""Coordinate descent over the pre-sweep naps of the persistent decode kernels (FISHRT_NAPS_FAST / FISHRT_NAPS_SLOW: 64-clock units
before the first sweep of each stage kind) on the configs[1] workload; prints the decode us/frame after every improving move.
usage: tune_naps.py [greedy|sampled] [dtype] [start naps, 12 comma-separated] [KV length of a random text prompt: tunes the slow kernel only]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
mode = sys.argv[1] if len(sys.argv) > 1 else "greedy"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
kw = dict(temp=0.0, top_p=1.0, top_k=0) if mode == "greedy" else dict(temp=0.7, top_p=0.8, top_k=256)
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
F = 192
KV = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if KV:
    lm.close()
    lm = fishrt.DualARTransformer(dict(fcfg.FISH_1_5, max_seq_len=8192), fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
    p = np.zeros((9, KV), np.uint32); p[0] = np.random.RandomState(1).randint(0, 100000, KV)
    F = 96
M = F + p.shape[1] - 2


def measure(naps):
    os.environ["FISHRT_NAPS_FAST"] = ",".join(str(v) for v in naps[:6])
    os.environ["FISHRT_NAPS_SLOW"] = ",".join(str(v) for v in naps[6:])
    lm.debug_capture(0)  # drops the captured graphs: the next call re-captures with the new kernel arguments
    best = 1e9
    for _ in range(2):
        lm.clear_slow_layer_caches()
        lm.generate_blocking(p, M, repetition_penalty=1.2, seed=1, ignore_eos=True, **kw)
        best = min(best, lm.last_stats()["decode_ms"] * 1e3 / (F - 1))
    return best


naps = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else [16, 16, 20, 20, 20, 12, 24, 0, 8, 40, 32, 12])]
cur = measure(naps)
print(f"start {naps}: {cur:.1f} us/frame", flush=True)
names = ["f.S1", "f.S2", "f.S3", "f.S4", "f.head", "f.dec", "s.S1", "s.S2", "s.S3", "s.S4", "s.S5", "s.head"]
for sweep in range(3):
    improved = False
    for i in (range(6, 12) if KV else range(12)):
        for v in (sorted(set([max(0, naps[i] - 8), max(0, naps[i] - 4), naps[i] + 4, naps[i] + 8, naps[i] + 16, 0, 2]))):
            if v == naps[i]:
                continue
            t = measure(naps[:i] + [v] + naps[i + 1:])
            if t < cur - 0.3:
                cur, naps[i], improved = t, v, True
                print(f"  {names[i]} = {v}: {cur:.1f} us/frame  {naps}", flush=True)
    if not improved:
        break
print(f"final {naps}: {cur:.1f} us/frame")
print("FAST", ",".join(str(v) for v in naps[:6]), "SLOW", ",".join(str(v) for v in naps[6:]))
