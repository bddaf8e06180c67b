"""Quick A/B of the persistent decode path at configs[1] shapes: decode us/frame (+ per-stage profile with FISHRT_PERSIST_PROF=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
M = 256 + p.shape[1] - 2
outs = []
for i in range(3):
    lm.clear_slow_layer_caches()
    out = lm.generate_blocking(p, M, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, seed=1, ignore_eos=True)
    outs.append(out)
st = lm.last_stats()
import zlib
print(f"[{dtype}] greedy {st['decode_ms']*1e3/255:.1f} us/frame  kernels/frame {st.get('kernels_per_frame')}  crc {zlib.crc32(outs[-1].tobytes()):08x} same={all(np.array_equal(outs[0], o) for o in outs)}")
