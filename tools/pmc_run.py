"""One configs[1]-shaped request (default-voice prompt + N frames, greedy) for the rocprofv3 --pmc passes of tools/pmc_traffic.sh.
torch is imported first: under rocprofv3 the process must load torch's HIP runtime before libfishrt (as bench.py does)."""
import os, sys
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import bench, fishrt
from fishrt import config as fcfg
frames = int(sys.argv[1]); persistent = sys.argv[2] == "1"
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(bench.SEED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
out = lm.generate_blocking(p, frames + p.shape[1] - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True, persistent=persistent)
print("frames", out.shape[1], lm.last_stats())
