"""Decode us per R-row frame of the request-row kernels (fs_lm_generate_multi, configs[1] prompt x R, 256 frames, greedy), best of 3.
usage: rows_quick.py R [dtype]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import numpy as np, bench, fishrt
from fishrt import config as fcfg
R = int(sys.argv[1]); dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype, max_batch=8).load_synthetic(bench.SEED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
best, crc = 1e9, None
for _ in range(3):
    out = lm.generate_multi([p] * R, [256 + p.shape[1] - 2] * R, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    st = lm.last_stats()
    best = min(best, st["decode_ms"] * 1e3 / 255)
    crc = zlib.crc32(b"".join(o.tobytes() for o in out))
print(f"[{dtype}] R={R}: {best:.1f} us per {R}-row frame  launches/frame {st.get('kernels_per_frame')}  crc {crc:08x}")
