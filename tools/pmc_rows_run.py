"""R concurrent greedy requests on the request-row persistent kernels (fs_lm_generate_multi), every row with the SAME frame budget, for the
rocprofv3 --pmc passes of tools/pmc_batch.sh.  usage: pmc_rows_run.py R frames.  torch is imported first (see tools/pmc_run.py)."""
import os, sys
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import numpy as np, bench, fishrt
from fishrt import config as fcfg
R, frames = int(sys.argv[1]), int(sys.argv[2])
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(bench.SEED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
out = lm.generate_multi([p] * R, [frames + p.shape[1] - 2] * R, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
print("frames", [o.shape[1] for o in out], lm.last_stats())
