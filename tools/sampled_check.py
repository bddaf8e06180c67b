"""Sampled decoding: persistent kernels (in-launch block-parallel sampler) vs the per-node path: token agreement + decode us/frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
M = 256 + p.shape[1] - 2
for kw in (dict(temp=0.7, top_p=0.8, top_k=256), dict(temp=0.7, top_p=0.9, top_k=50), dict(temp=1.0, top_p=0.3, top_k=256)):
    res = {}
    for persistent in (False, True, True):
        lm.clear_slow_layer_caches()
        out = lm.generate_blocking(p, M, repetition_penalty=1.2, seed=11, ignore_eos=True, persistent=persistent, **kw)
        st = lm.last_stats()
        res[persistent] = out
        print(f"{kw} persistent={persistent}: {st['decode_ms']*1e3/255:.1f} us/frame, kernels/frame {st['kernels_per_frame']}, frames {out.shape[1]}")
    a, b = res[False], res[True]
    neq = (a != b).any(axis=0)
    print(f"   identical frames {int((~neq).sum())}/{a.shape[1]}, first differing frame {int(np.argmax(neq)) if neq.any() else -1}")
