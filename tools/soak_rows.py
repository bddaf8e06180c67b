"""Soak of the request-row persistent kernels: REPS multi-request calls of N rows x F frames (Fish-1.5 shapes, bf16, distinct prompts and
ragged budgets); every repetition must reproduce the first one's tokens row by row (a missed / torn edge granule or a timed-out wait would
change them or raise), and rows that carry the SAME request must agree with each other.  usage: soak_rows.py [F] [N] [REPS] [sampled]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
kw = dict(temp=0.7, top_p=0.8, top_k=256) if len(sys.argv) > 4 and sys.argv[4] == "sampled" else dict(temp=0.0, top_p=1.0, top_k=0)
lm = fishrt.DualARTransformer(dict(fcfg.FISH_1_5, max_seq_len=8192), fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(9)
prompts, budgets = [], []
for i in range(N):
    L = 64 + 13 * (i // 2)           # rows 2k and 2k+1 carry the same request (same prompt, budget and seed)
    if i % 2 == 0:
        p = np.zeros((9, L), np.uint32); p[0] = rng.randint(0, 100000, L)
    prompts.append(p); budgets.append(L + F - 2 - 7 * (i // 2))
seeds = [11 + i // 2 for i in range(N)]
ref, t0, frames = None, time.time(), 0
for rep in range(REPS):
    outs = lm.generate_multi(prompts, budgets, repetition_penalty=1.2, seeds=seeds, ignore_eos=True, **kw)
    st = lm.last_stats()
    assert st["kernels_per_frame"] == 1 + (N + 3) // 4, st
    for i in range(0, N - 1, 2):
        assert np.array_equal(outs[i], outs[i + 1]), f"rep {rep}: rows {i} and {i + 1} carry the same request and differ at frame {int(np.argmax((outs[i] != outs[i + 1]).any(0)))}"
    if ref is None:
        ref = outs
    for i in range(N):
        assert outs[i].shape == ref[i].shape and np.array_equal(outs[i], ref[i]), f"rep {rep} row {i} differs from rep 0"
    frames += sum(o.shape[1] for o in outs)
nfr = max(o.shape[1] for o in outs)
print(f"rows soak ok [N={N}, {kw}]: {REPS} x {N} rows x ~{F} frames = {frames} request-frames, identical tokens, {time.time() - t0:.1f} s, "
      f"last decode {st['decode_ms'] / (nfr - 1) * 1e3:.1f} us per {N}-row frame (KV up to {max(budgets)})")
