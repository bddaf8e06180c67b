"""BASELINE.json configs[2]: Fish-1.5 bf16, B concurrent requests, top-p 0.8 / temp 0.7 / top-k 256, 256 frames, one MI355X."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, torch, fishrt  # torch first: under rocprofv3 its HIP runtime must load before libfishrt
from fishrt import config as fcfg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=B).load_synthetic(0xF15E5EED)
rng = np.random.RandomState(77)
prompts = []
for L in rng.randint(64, 385, B):   # prompt lengths U{64..384} seed 77 (SURVEY.md §8d configs[2])
    p = np.zeros((9, int(L)), np.uint32); p[0] = rng.randint(0, 100000, int(L)); prompts.append(p)
Lmax = max(p.shape[1] for p in prompts)
M = frames + Lmax - 2
for rep in range(reps):
    t = time.perf_counter()
    outs = lm.generate_static_batch(prompts, M, temp=0.7, top_p=0.8, top_k=256, seed=42, ignore_eos=True)
    dt = time.perf_counter() - t
    st = lm.last_stats()
    tot = sum(o.shape[1] for o in outs)
    step_us = st["decode_ms"] * 1e3 / (frames - 1)
    print(f"B={B} Lmax={Lmax}: {tot} frames in {dt*1e3:.1f} ms wall (prefill {st['prefill_ms']:.1f} ms, decode {st['decode_ms']:.1f} ms) -> "
          f"{tot/dt:.0f} frames/s wall, decode {B*(frames-1)/(st['decode_ms']*1e-3):.0f} frames/s, {step_us:.0f} us/step, "
          f"roofline frac {(1.695e9 + 12288*B*(Lmax+frames/2))/(step_us*1e-6)/8e12:.3f}")
