import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fishrt
from fishrt import config as fcfg
TOK = fcfg.FISH_1_5_TOKENS
lm8 = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16", max_batch=8).load_synthetic(0xF15E5EED)
def tp(L, seed):
    rng = np.random.RandomState(seed); p = np.zeros((9, L), np.uint32); p[0] = rng.randint(0, TOK["im_end_id"], L); return p
prompts = [tp(12, 1000 + s) for s in range(24)]
M = 70
ref = []
for p in prompts:
    lm8.clear_slow_layer_caches()
    ref.append(lm8.generate_blocking(p, 12 + M, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2))
for g0 in range(0, 24, 4):
    got = lm8.generate_multi(prompts[g0:g0 + 4], 12 + M, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2)
    for i in range(4):
        a, b = got[i], ref[g0 + i]
        n = min(a.shape[1], b.shape[1])
        neq = (a[:, :n] != b[:, :n]).any(0)
        print(g0 + i, "multi", a.shape[1], "single", b.shape[1], "first diff", int(neq.argmax()) if neq.any() else -1)
