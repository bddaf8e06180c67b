"""Vocoder alone (no LM on the device): 4096 frames decoded at once, as 16 stateful chunks of 256 (fs_codec_stream_*), and as 16 stateless chunks
with a 24-frame halo (the round-2 streaming); wall ms incl. H2D / D2H, best of 3, and PCM equality."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
T, CH = 4096, 256
codes = np.random.RandomState(3).randint(0, 1000, (8, T)).astype(np.uint32)
c = fishrt.FireflyCodec(0, precision=prec).load_synthetic(0xC0DEC)
def best(f):
    b, out = 1e9, None
    for _ in range(3):
        t = time.perf_counter(); out = f(); b = min(b, time.perf_counter() - t)
    return b * 1e3, out
def one(): return c.decode(np.ascontiguousarray(codes[None]))[0, 0]
class Parts(list):  # (the chunks are compared piece by piece: concatenating 33 MB is not vocoder time)
    pass
def stateful():
    c.stream_begin(); p = Parts(c.stream_decode(np.ascontiguousarray(codes[:, a:a + CH])) for a in range(0, T, CH)); c.stream_end(); return p
def halo(): return Parts(fishrt.decode_chunk(c, codes, a, a + CH) for a in range(0, T, CH))
one()
t1, p1 = best(one); t2, p2 = best(stateful); t3, p3 = best(halo)
p2, p3 = np.concatenate(p2), np.concatenate(p3)
print(f"[{prec}] {T} frames: one shot {t1:.1f} ms | 16 stateful chunks {t2:.1f} ms ({t2 / t1:.2f} x, identical {np.array_equal(p1, p2)}) | "
      f"16 halo chunks {t3:.1f} ms ({t3 / t1:.2f} x, identical {np.array_equal(p1, p3)})")
