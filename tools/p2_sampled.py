"""Like p2_quick.py for a SAMPLED request (temp 0.7 / top-p 0.8 / top-k 256 by default; argv: dtype temp top_p top_k): decode us/frame
(+ the per-stage profile of k_fast_persist<true> with FISHRT_PERSIST_PROF=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
temp, top_p, top_k = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7, float(sys.argv[3]) if len(sys.argv) > 3 else 0.8, int(sys.argv[4]) if len(sys.argv) > 4 else 256
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
M = 256 + p.shape[1] - 2
for i in range(3):
    lm.clear_slow_layer_caches()
    out = lm.generate_blocking(p, M, temp=temp, top_p=top_p, top_k=top_k, repetition_penalty=1.2, seed=1, ignore_eos=True)
st = lm.last_stats()
import zlib
print(f"[{dtype}] temp {temp} top_p {top_p} top_k {top_k}: {st['decode_ms']*1e3/255:.1f} us/frame  kernels/frame {st.get('kernels_per_frame')}  crc {zlib.crc32(out.tobytes()):08x}")
