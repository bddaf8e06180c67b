"""Soak of the persistent decode kernels: R greedy requests of F frames each on one handle (Fish-1.5 shapes, bf16); every request must
produce the first one's tokens (a missed / torn edge granule or a timed-out wait would change them or raise)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
kw = dict(temp=0.7, top_p=0.8, top_k=256) if len(sys.argv) > 4 and sys.argv[4] == "sampled" else dict(temp=0.0, top_p=1.0, top_k=0)
lm = fishrt.DualARTransformer(dict(fcfg.FISH_1_5, max_seq_len=8192), fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(9)
p = np.zeros((9, 64), np.uint32); p[0] = rng.randint(0, 100000, 64)
ref, t0 = None, time.time()
for r in range(R):
    lm.clear_slow_layer_caches()
    out = lm.generate_blocking(p, F + 62, repetition_penalty=1.2, seed=7, ignore_eos=True, **kw)
    st = lm.last_stats()
    assert out.shape == (8, F) and st["kernels_per_frame"] == 2, (out.shape, st)
    ref = out if ref is None else ref
    assert np.array_equal(out, ref), f"request {r} differs from request 0 at frame {int(np.argmax((out != ref).any(0)))}"
print(f"soak ok [{dtype}, {kw}]: {R} x {F} frames = {R * F} persistent frames, identical tokens, {time.time() - t0:.1f} s, last decode {st['decode_ms'] / (F - 1) * 1e3:.1f} us/frame")
