"""Vocoder A/B: decode time of the default mode and its PCM distance from the exact-f32 mode (which is 4e-8 from the oracle)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = sys.argv[2:] or ["bf16x3"]
G = os.path.join(ROOT, "tests", "golden")
voice = np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32)
codes = np.ascontiguousarray(np.tile(voice, (1, (T + voice.shape[1] - 1) // voice.shape[1]))[:, :T])[None]
c = fishrt.FireflyCodec(0, precision="f32").load_synthetic(0xC0DEC)
ref = c.decode(codes)[0, 0].astype(np.float64); c.close()
for m in modes:
    c = fishrt.FireflyCodec(0, precision=m).load_synthetic(0xC0DEC)
    for _ in range(3): pcm = c.decode(codes)
    best = 1e9
    for _ in range(7):
        t = time.perf_counter(); pcm = c.decode(codes); best = min(best, time.perf_counter() - t)
    st = c.last_stats() if hasattr(c, "last_stats") else None
    c.close()
    r = float(np.sqrt(np.mean((pcm[0, 0].astype(np.float64) - ref) ** 2)))
    print(f"[{m}] T={T}: {best*1e3:.2f} ms/decode wall (incl. D2H)  stats {st}  rms vs f32 mode {r:.2e}  signal rms {float(np.sqrt(np.mean(ref**2))):.4f}  max|d| {float(np.abs(pcm[0,0]-ref).max()):.2e}")
