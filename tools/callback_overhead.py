import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS, 0, "fp8").load_synthetic(0xF15E5EED)
rng = np.random.RandomState(4); L = 64
p = np.zeros((9, L), np.uint32); p[0] = rng.randint(6, 32000, L)
kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=1, ignore_eos=True)
M = 2048 + L - 2
for rep in range(2):
    lm.clear_slow_layer_caches(); t = time.perf_counter(); lm.generate_blocking(p, M, **kw); t0 = time.perf_counter() - t; d0 = lm.last_stats()["decode_ms"]
    lm.clear_slow_layer_caches(); t = time.perf_counter(); lm.generate_blocking(p, M, on_frame=lambda i, c: False, **kw); t1 = time.perf_counter() - t; d1 = lm.last_stats()["decode_ms"]
    class Null:
        def decode(s, codes): return np.zeros((1, 1, 2048 * codes.shape[2]), np.float32)
    lm.clear_slow_layer_caches(); syn = fishrt.StreamingSynth(lm, Null()); t = time.perf_counter(); syn(p, M, **kw); t2 = time.perf_counter() - t; d2 = lm.last_stats()["decode_ms"]
    print(f"2048 frames: no callback {t0:.3f}s (decode {d0:.0f} ms) | no-op callback {t1:.3f}s (decode {d1:.0f} ms) | StreamingSynth with a null vocoder {t2:.3f}s (decode {d2:.0f} ms)")
