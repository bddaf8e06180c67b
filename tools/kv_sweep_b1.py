"""batch-1 persistent path: decode us/frame at several KV lengths (the slow kernel's attention slice count n_sl changes at 128 x 2^k cached tokens)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
def text(L, seed=1):
    p = np.zeros((9, L), np.uint32); p[0] = np.random.RandomState(seed).randint(0, 100000, L); return p
for name, p, F in (("KV 367..622 (configs[1], 256 frames)", bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS), 256), ("KV 300..395", text(300), 96), ("KV 600..695", text(600), 96),
                   ("KV 1500..1595", text(1500), 96), ("KV 4100..4195", text(4100), 96)):
    best = 1e9
    for _ in range(3):
        lm.clear_slow_layer_caches()
        out = lm.generate_blocking(p, F + p.shape[1] - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
        best = min(best, lm.last_stats()["decode_ms"] * 1e3 / (F - 1))
    print(f"{name:40s} {best:7.1f} us/frame   first codes {out[:, 1].tolist()[:4]}")
