"""decode time of the vocoder at T frames in both precision modes (HIP path only; prints ms per call)"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fish-speech.rs_amd"))
import torch  # noqa: F401  (rocprofv3 needs torch's HIP runtime loaded first)
import fishrt

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
codes = np.random.RandomState(0).randint(0, 1000, (1, 8, T)).astype(np.uint32)
for prec in (sys.argv[2:] or ["bf16x3", "f32"]):
    c = fishrt.FireflyCodec(0, precision=prec).load_synthetic(0xC0DEC)
    for _ in range(2):
        c.decode(codes)
    t = time.perf_counter()
    n = 5
    for _ in range(n):
        pcm = c.decode(codes)
    dt = (time.perf_counter() - t) / n
    print(f"{prec}: T={T} {dt * 1e3:.2f} ms/decode (incl. {pcm.nbytes / 1e6:.1f} MB D2H copy), rms {float(np.sqrt((pcm.astype(np.float64) ** 2).mean())):.4f}")
    c.close()
