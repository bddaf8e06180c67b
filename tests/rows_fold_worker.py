"""Worker of tests/test_rows_fold_gpu.py: one seeded generate_static_batch with decision capture; codes + capture to an .npz.
usage: rows_fold_worker.py {fish15|tiny|mid} {bf16|fp8} B FRAMES OUT.npz   (the FISHRT_ROWS_* knobs come from the environment)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import numpy as np

import fishrt
from fishrt import config as fcfg

cfg_name, dtype, B, F, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if cfg_name == "fish15":
    cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
elif cfg_name == "mid":
    cfg, tok = dict(fcfg.TINY, dim=256, n_head=4, n_local_heads=2, head_dim=64, intermediate_size=1024), fcfg.TINY_TOKENS
else:
    cfg, tok = fcfg.TINY, fcfg.TINY_TOKENS
n_slow = cfg["vocab_size"] - tok["im_end_id"]
greedy = cfg_name != "fish15"
lm = fishrt.DualARTransformer(cfg, tok, 0, dtype, max_batch=B).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(5)
prompts = []
for L in rng.randint(6, 40, B):
    p = np.zeros((9, int(L)), np.uint32)
    p[0] = rng.randint(0, tok["im_end_id"] - 1, int(L))
    prompts.append(p)
Lmax = max(p.shape[1] for p in prompts)
lm.debug_capture(F)
kw = dict(temp=0.0, top_p=1.0, top_k=0) if greedy else dict(temp=0.7, top_p=0.8, top_k=256)
outs = lm.generate_static_batch(prompts, Lmax + F - 1, seed=11, ignore_eos=True, **kw)
cap = np.stack([lm.debug_read_row(b, F) for b in range(B)])  # (B, F, 9, 2048)
codes = np.stack([o[:, :F] for o in outs])
np.savez(out, codes=codes, cap=cap, n_slow=n_slow, greedy=greedy)
mode = "nofold" if os.environ.get("FISHRT_ROWS_NO_FOLD") == "1" else "fold"
print(json.dumps({"graph_nodes_hint": mode, "frames": int(codes.shape[2])}))
