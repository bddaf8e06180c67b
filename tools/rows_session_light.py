"""An 8-slot request-row session under LIGHT load: k concurrent sampled requests (k = 1, 2, 3, 5) on a handle with 8 slots; step time per frame.
usage: rows_session_light.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
import numpy as np, bench, fishrt
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(bench.SEED)
p = bench.default_voice_prompt(fcfg.FISH_1_5_TOKENS)
F = 128
for k in (1, 2, 3, 4, 5, 8):
    with lm.session(seed=3, ignore_eos=True, rows=True, repetition_penalty=1.2, temp=0.7, top_p=0.8, top_k=256) as s:
        slots = [s.add(p, F + p.shape[1] - 2) for _ in range(k)]
        s.step(1)                      # admission (prefill) + first frame
        t0 = time.perf_counter()
        n = 0
        while s.step(16):
            n += 16
        dt = time.perf_counter() - t0
        got = [s.poll(sl, codes=False) for sl in slots]
        assert all(g[0] == F and g[1] for g in got), got
        for sl in slots: s.release(sl)
    print(f"{k} live requests on 8 slots: {dt / (F - 1) * 1e6:.0f} us per frame -> {k * (F - 1) / dt:.0f} frames/s")
