import sys, os
import numpy as np
sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import fishrt
a = fishrt.FireflyCodec(0, precision="f16").load_synthetic(0xC0DEC)
b = fishrt.FireflyCodec(0, precision="f32").load_synthetic(0xC0DEC)
T = 64
cases = {"zeros": np.zeros((1, 8, T), np.uint32), "all999": np.full((1, 8, T), 999, np.uint32),
         "alt": np.tile(np.array([0, 999], np.uint32), (1, 8, T // 2)).reshape(1, 8, T),
         "ramp": (np.arange(8 * T, dtype=np.uint32).reshape(1, 8, T) * 37) % 1000}
for k, c in cases.items():
    c = np.ascontiguousarray(c)
    x, y = a.decode(c)[0, 0].astype(np.float64), b.decode(c)[0, 0].astype(np.float64)
    print(f"{k:8s}: finite {np.isfinite(x).all()}  rms diff {np.sqrt(np.mean((x - y) ** 2)):.2e}  signal rms {np.sqrt(np.mean(y ** 2)):.3f}  max|pcm| {np.abs(y).max():.3f}")
