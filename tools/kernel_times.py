"""us per graph node of the five batch-1 slow-layer kernels (fs_lm_bench_kernel) for a bf16 and an fp8 handle."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import fishrt
from fishrt import config as fcfg
names = ["qkv", "attention", "wo", "ffn_up", "ffn_down", "fast-layer node (avg of 4)", "fast head", "slow head"]
for dt in ("bf16", "fp8"):
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dt).load_synthetic(0xF15E5EED)
    t = [lm.bench_kernel(k, 495, 50) for k in range(8)]
    print(dt, " ".join(f"{n} {v:.2f}" for n, v in zip(names, t)), f"| slow layer {sum(t[:5]):.2f} us, fast layer {4 * t[5]:.2f} us")
    lm.close()
