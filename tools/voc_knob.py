"""Vocoder decode time (256 frames, f16 mode, best of 5) + PCM checksum under the current tuning-knob environment."""
import sys, time, os, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
c = fishrt.FireflyCodec(0, precision=os.environ.get("VOC_PREC", "f16")).load_synthetic(0xC0DEC)
codes = np.random.RandomState(1).randint(0, 1000, (1, 8, 256)).astype(np.uint32)
c.decode(codes)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); pcm = c.decode(codes); best = min(best, time.perf_counter() - t)
print(f"{best*1e3:.3f} ms  crc {zlib.crc32(pcm.tobytes()):08x}  knobs " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FISHRT_")))
