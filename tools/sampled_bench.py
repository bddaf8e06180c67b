"""Decode time per frame at configs[1] shapes under different sampling settings.  usage: sampled_bench.py [bf16|fp8|f32]"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt, bench
from fishrt import config as fcfg
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
M = 256 + p.shape[1] - 2
for name, kw in (("greedy", dict(temp=0.0, top_p=1.0, top_k=0)), ("temp0.7 top_p0.8 top_k256", dict(temp=0.7, top_p=0.8, top_k=256)),
                 ("temp0.7 top_p0.9 top_k50", dict(temp=0.7, top_p=0.9, top_k=50)),
                 ("temp0.05 top_p0.8 top_k256 (sharper rows)", dict(temp=0.05, top_p=0.8, top_k=256)), ("temp0.02 top_p0.8 top_k256 (peaked rows)", dict(temp=0.02, top_p=0.8, top_k=256)), ("temp0.7 top_p0.8 top_k0", dict(temp=0.7, top_p=0.8, top_k=0))):
    for _ in range(2):
        lm.clear_slow_layer_caches()
        out = lm.generate_blocking(p, M, repetition_penalty=1.2, seed=1, ignore_eos=True, **kw)
    st = lm.last_stats()
    print(f"[{dtype}] {name:42s}: {st['decode_ms']*1e3/255:.1f} us/frame")
