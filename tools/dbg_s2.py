import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, fishrt
from fishrt import config as fcfg
import test_persist_gpu as tp
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(tp.SEED)
p = tp._text_prompt(12, 1010)
lm.debug_capture(80)
lm.clear_slow_layer_caches()
b = lm.generate_blocking(p, 12 + 70, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, persistent=True)
cap = lm.debug_read(80)
for f in range(48, 56):
    s = cap[f, 0, :2037]; s = np.where(np.isfinite(s), s, -1e30)
    o = np.argsort(-s)[:2]
    print("frame", f, "slow pick", int(cap[f, 0, 2047]), "top2", o, "margin", float(s[o[0]] - s[o[1]]))
