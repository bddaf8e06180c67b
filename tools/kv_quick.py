"""Decode us/frame of the batch-1 persistent path behind a random text prompt of KV tokens (long-form regime of BASELINE configs[4]);
with FISHRT_PERSIST_PROF=1 the per-stage profile of the last call.  usage: kv_quick.py KV [dtype] [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import fishrt
from fishrt import config as fcfg
KV = int(sys.argv[1]); dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"; F = int(sys.argv[3]) if len(sys.argv) > 3 else 96
lm = fishrt.DualARTransformer(dict(fcfg.FISH_1_5, max_seq_len=8192), fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
p = np.zeros((9, KV), np.uint32); p[0] = np.random.RandomState(1).randint(0, 100000, KV)
best = 1e9
for _ in range(2):
    lm.clear_slow_layer_caches()
    lm.generate_blocking(p, KV + F - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, seed=1, ignore_eos=True)
    st = lm.last_stats()
    best = min(best, st["decode_ms"] * 1e3 / (F - 1))
print(f"[{dtype}] KV {KV}..{KV + F}: {best:.1f} us/frame  kernels/frame {st.get('kernels_per_frame')} prefill {st['prefill_ms']:.2f} ms")
