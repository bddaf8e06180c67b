"""Where does a request's wall time go outside the HIP-event-timed GPU region?  (configs[1] request, bf16)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
from fishrt import config as fcfg
import bench
cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
lm = fishrt.DualARTransformer(cfg, tok, 0, "bf16").load_synthetic(bench.SEED)
p = bench.default_voice_prompt(tok)
M = 256 + p.shape[1] - 2
kw = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
for i in range(4):
    t0 = time.perf_counter(); lm.clear_slow_layer_caches(); t1 = time.perf_counter()
    out = lm.generate_blocking(p, M, **kw); t2 = time.perf_counter()
    st = lm.last_stats()
    print(f"clear {1e3*(t1-t0):.3f} ms | generate wall {1e3*(t2-t1):.3f} ms | gpu prefill {st['prefill_ms']:.3f} + decode {st['decode_ms']:.3f} = {st['prefill_ms']+st['decode_ms']:.3f} ms"
          f" | outside {1e3*(t2-t1)-st['prefill_ms']-st['decode_ms']:.3f} ms")
