"""Persistent fast decoder vs the per-node graph path on one GPU: token agreement + decode time per frame (Fish-1.5 shapes, bf16)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg

lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
rng = np.random.RandomState(1234)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prompt = np.zeros((9, L), np.uint32)
prompt[0] = rng.randint(0, 100000, L)
res = {}
for rp in (1.0, 1.2):
    for persistent in (False, True, True):
        lm.clear_slow_layer_caches()
        t0 = time.time()
        out = lm.generate_blocking(prompt, M, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=rp, ignore_eos=True, persistent=persistent)
        st = lm.last_stats()
        us = st["decode_ms"] * 1e3 / max(1, st["frames"] - 1)
        print(f"rep_pen {rp} persistent={persistent}: frames {out.shape[1]} decode {us:.1f} us/frame wall {time.time()-t0:.3f}s first codes {out[:, 1].tolist()}")
        res[(rp, persistent)] = out
    a, b = res[(rp, False)], res[(rp, True)]
    n = min(a.shape[1], b.shape[1])
    neq = (a[:, :n] != b[:, :n]).any(axis=0)
    first = int(np.argmax(neq)) if neq.any() else -1
    print(f"rep_pen {rp}: identical frames {int((~neq).sum())}/{n}, first differing frame {first}")
