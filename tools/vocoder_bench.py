import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
c = fishrt.FireflyCodec(0).load_synthetic(1)
rng = np.random.RandomState(0)
for T in (64, 256):
    codes = rng.randint(0, 1000, (1, 8, T)).astype(np.uint32)
    c.decode(codes)
    t = time.perf_counter(); c.decode(codes); dt = time.perf_counter() - t
    print(f"T={T}: {dt*1e3:.1f} ms -> {2.65e9*T/dt/1e12:.2f} TFLOP/s, RTF {(T/21.535)/dt:.1f}")
