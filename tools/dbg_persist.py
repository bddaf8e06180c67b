import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc
import test_persist_gpu as T
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(T.SEED)
o = orc.OracleLM(orc.FISH15).load_synthetic(T.SEED, bf16=True); o.set_kv_round_bf16(True)
for p in (T._text_prompt(16, 1234), T._vq_prompt(96, 7), T._text_prompt(200, 99)):
    L = p.shape[1]
    lm.clear_slow_layer_caches(); a = lm.generate_blocking(p, L + 62, repetition_penalty=1.2, persistent=False, **T.GREEDY)
    lm.clear_slow_layer_caches(); b = lm.generate_blocking(p, L + 62, repetition_penalty=1.2, persistent=True, **T.GREEDY)
    o.clear_slow(); e = o.generate(p, L + 62, temp=0.0, repetition_penalty=1.2, ignore_eos=True)
    da = np.nonzero((a != e).any(0))[0]; db = np.nonzero((b != e).any(0))[0]; dab = np.nonzero((a != b).any(0))[0]
    fa = int(da[0]) if da.size else -1; fb = int(db[0]) if db.size else -1; fab = int(dab[0]) if dab.size else -1
    print("L", L, "per-node vs oracle first diff", fa, "persist vs oracle", fb, "per-node vs persist", fab)
    for f in sorted(set([x for x in (fa, fb, fab) if x >= 0])):
        print("  frame", f, "oracle margin", o.last_margins[f], "oracle", e[:, f].tolist(), "per-node", a[:, f].tolist(), "persist", b[:, f].tolist())
