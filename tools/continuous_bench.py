import sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, torch
import bench, fishrt
from fishrt import config as fcfg
cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
prompts = bench.config2_prompts(tok, 64)
lmb = fishrt.DualARTransformer(cfg, tok, 0, "bf16", max_batch=32).load_synthetic(bench.SEED)
for _ in range(2):
    print(json.dumps(bench.continuous_vs_lockstep(lmb, prompts)))
