"""A/B: the batch-1 frame with k_fast_persist (resident W2 in LDS) vs k_fast_rows<1> (W2 streamed, row-pair weights streamed, no spills in the
sampled instantiation): decode us/frame on the configs[1] prompt.  FISHRT_B1_ROWS_FAST=1 selects the row kernel (experiment hook)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=2).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
F = 192
M = F + p.shape[1] - 2
for name, kw in (("greedy", dict(temp=0.0, top_p=1.0, top_k=0)), ("sampled", dict(temp=0.7, top_p=0.8, top_k=256))):
    res = {}
    for hook in (False, True, False, True):
        if hook: os.environ["FISHRT_B1_ROWS_FAST"] = "1"
        else: os.environ.pop("FISHRT_B1_ROWS_FAST", None)
        best = 1e9
        for _ in range(2):
            lm.clear_slow_layer_caches()
            out = lm.generate_blocking(p, M, repetition_penalty=1.2, seed=1, ignore_eos=True, **kw)
            best = min(best, lm.last_stats()["decode_ms"] * 1e3 / (F - 1))
        res.setdefault(hook, []).append((best, out))
    a, b = res[False][-1], res[True][-1]
    same = int(np.argmax((a[1] != b[1]).any(0))) if (a[1] != b[1]).any() else F
    print(f"{name}: k_fast_persist {res[False][0][0]:.1f} / {a[0]:.1f} us/frame   k_fast_rows<1> {res[True][0][0]:.1f} / {b[0]:.1f} us/frame   identical frames {same}/{F}")
