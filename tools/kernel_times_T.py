"""attention / wo node times vs KV length (fs_lm_bench_kernel, bf16)"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import fishrt
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
for T in (120, 250, 495, 1000, 2000, 4000, 8000):
    print(T, "attention %.2f us, wo %.2f us" % (lm.bench_kernel(1, T, 30), lm.bench_kernel(2, T, 30)))
