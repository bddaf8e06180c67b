"""R concurrent SAMPLED configs[1]-shaped requests on the request-row kernels (temp 0.7 / top-p 0.8 / top-k 256, rep-pen 1.2): frame time,
and every row's codes against its own fs_lm_generate call with the same seed.  usage: rows_sampled_bench.py [R] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import numpy as np, bench, fishrt
from fishrt import config as fcfg
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(bench.SEED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, ignore_eos=True)
best = 1e9
for _ in range(3):
    outs = lm.generate_multi([p] * R, F + p.shape[1] - 2, seeds=list(range(1, R + 1)), **kw)
    st = lm.last_stats()
    best = min(best, st["decode_ms"] * 1e3 / (F - 1))
print(f"R={R} sampled: {best:.1f} us per frame -> {R * 1e6 / best:.0f} frames/s, launches/frame {st['kernels_per_frame']}")
same = []
for i in range(R):
    lm.clear_slow_layer_caches()
    o = lm.generate_blocking(p, F + p.shape[1] - 2, seed=i + 1, **kw)
    neq = (o != outs[i]).any(0) if o.shape == outs[i].shape else np.array([True])
    same.append(int(np.argmax(neq)) if neq.any() else o.shape[1])
print("frames identical to the row's own sampled generate call:", same)
