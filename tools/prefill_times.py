"""prefill wall time vs prompt length (bf16, one sequence; forward_generate over L tokens, best of 3)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
rng = np.random.RandomState(0)
for L in (64, 128, 256, 367, 512, 1024, 2048, 4096, 8000):
    p = np.zeros((9, L), np.uint32); p[0] = rng.randint(0, 100000, L)
    best = 1e9
    for _ in range(3):
        lm.clear_slow_layer_caches()
        t = time.perf_counter(); lm.forward_generate(p, 0); best = min(best, time.perf_counter() - t)
    print(f"L={L}: {best*1e3:.2f} ms  ({L/best/1e3:.0f} k tokens/s)")
