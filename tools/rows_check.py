"""Request-row persistent kernels (fs_lm_generate_multi) vs one fs_lm_generate call per request: token agreement + time (Fish-1.5, bf16).
usage: rows_check.py [n_rows] [frames] [prompt_len]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 24
L0 = int(sys.argv[3]) if len(sys.argv) > 3 else 48
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(4321)
prompts, mnt = [], []
for i in range(n):
    L = L0 + 7 * i
    p = np.zeros((9, L), np.uint32)
    p[0] = rng.randint(0, 100000, L)
    prompts.append(p)
    mnt.append(L + F - 2 + (i % 2))   # F or F + 1 iterations: ragged budgets
ref = []
t0 = time.time()
us1 = []
for i in range(n):
    lm.clear_slow_layer_caches()
    ref.append(lm.generate_blocking(prompts[i], mnt[i], temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True))
    st = lm.last_stats()
    us1.append(st["decode_ms"] * 1e3 / max(1, st["frames"] - 1))
print(f"one by one: {time.time() - t0:.3f}s, decode us/frame {np.mean(us1):.1f}, frames {[r.shape[1] for r in ref]}")
for rep in range(2):
    t0 = time.time()
    got = lm.generate_multi(prompts, mnt, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    st = lm.last_stats()
    nfr = max(g.shape[1] for g in got)
    print(f"multi: {time.time() - t0:.3f}s decode {st['decode_ms']:.2f} ms = {st['decode_ms'] * 1e3 / max(1, nfr - 1):.1f} us per {n}-row frame "
          f"({st['frames']} frames, {st['kernels_per_frame']} launches/frame) -> {1e3 * (st['frames'] - n) / st['decode_ms']:.0f} frames/s aggregate")
bad = 0
for i in range(n):
    a, b = ref[i], got[i]
    if a.shape != b.shape:
        print(f"row {i}: shape {a.shape} vs {b.shape}"); bad += 1; continue
    neq = (a != b).any(axis=0)
    first = int(np.argmax(neq)) if neq.any() else -1
    print(f"row {i}: L {prompts[i].shape[1]} frames {a.shape[1]} identical {int((~neq).sum())}/{a.shape[1]} first differing frame {first}"
          + (f" ref {a[:, first].tolist()} got {b[:, first].tolist()}" if first >= 0 else ""))
    bad += first >= 0 and first < 4
print("RESULT", "FAIL" if bad else "ok")
