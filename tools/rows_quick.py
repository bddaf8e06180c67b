"""us per R-row decode frame of fs_lm_generate_multi at configs[1] shapes (R x the default-voice request, 128 frames): rows_quick.py [dtype] [R ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt, bench
from fishrt import config as fcfg
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
Rs = [int(v) for v in sys.argv[2:]] or [2, 4, 8]
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype, max_batch=8).load_synthetic(0xF15E5EED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
F = 128
for R in Rs:
    for kw, name in ((dict(temp=0.0, top_p=1.0, top_k=0), "greedy"), (dict(temp=0.7, top_p=0.8, top_k=256), "sampled")):
        best = 1e9
        for _ in range(2):
            outs = lm.generate_multi([p] * R, F + p.shape[1] - 2, repetition_penalty=1.2, seeds=list(range(1, R + 1)), ignore_eos=True, **kw)
            st = lm.last_stats()
            best = min(best, st["decode_ms"] * 1e3 / (F - 1))
        import zlib
        print(f"[{dtype}] R={R} {name}: {best:.1f} us per {R}-row frame = {R * 1e6 / best:.0f} frames/s  launches/frame {st['kernels_per_frame']}  crc {zlib.crc32(np.concatenate(outs).tobytes()):08x}")
