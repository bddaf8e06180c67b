"""Pre-sweep naps (defaults tuned on configs[1]) vs no naps on other operating points of the persistent path: decode us/frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
pv = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
def text(L, seed=1):
    p = np.zeros((9, L), np.uint32); p[0] = np.random.RandomState(seed).randint(0, 100000, L); return p
cases = [("bf16 greedy, KV 367..", "bf16", pv, 128, dict(temp=0.0, top_p=1.0, top_k=0)),
         ("bf16 sampled k256, KV 367..", "bf16", pv, 128, dict(temp=0.7, top_p=0.8, top_k=256)),
         ("bf16 greedy, KV 16..", "bf16", text(16), 96, dict(temp=0.0, top_p=1.0, top_k=0)),
         ("bf16 greedy, KV 1500..", "bf16", text(1500), 96, dict(temp=0.0, top_p=1.0, top_k=0)),
         ("bf16 greedy, KV 4100..", "bf16", text(4100), 96, dict(temp=0.0, top_p=1.0, top_k=0)),
         ("fp8 greedy, KV 367..", "fp8", pv, 128, dict(temp=0.0, top_p=1.0, top_k=0))]
lms = {}
for name, dtype, p, F, kw in cases:
    if dtype not in lms:
        lms[dtype] = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
    lm = lms[dtype]
    res = []
    for naps in (None, "0,0,0,0,0,0"):
        for k in ("FISHRT_NAPS_FAST", "FISHRT_NAPS_SLOW"):
            if naps is None: os.environ.pop(k, None)
            else: os.environ[k] = naps
        lm.debug_capture(0)
        best = 1e9
        for _ in range(2):
            lm.clear_slow_layer_caches()
            lm.generate_blocking(p, F + p.shape[1] - 2, repetition_penalty=1.2, seed=1, ignore_eos=True, **kw)
            best = min(best, lm.last_stats()["decode_ms"] * 1e3 / (F - 1))
        res.append(best)
    print(f"{name:30s}: tuned naps {res[0]:7.1f} us/frame   no naps {res[1]:7.1f}")
