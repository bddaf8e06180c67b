"""Is a static batch of 32 faster as N concurrent static batches of 32/N (own handle / stream / host thread each)?  The B = 32 step is a chain of
363 latency-bound nodes whose duration does not depend on the row count, so independent chains should overlap like concurrent batch-1 streams do."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fish-speech.rs_amd"))
import torch  # noqa: F401
import fishrt
from fishrt import config as fcfg

B, frames = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.RandomState(77)
prompts = []
for L in rng.randint(64, 385, B):
    p = np.zeros((9, int(L)), np.uint32); p[0] = rng.randint(0, 100000, int(L)); prompts.append(p)
Lmax = max(p.shape[1] for p in prompts)
kw = dict(temp=0.7, top_p=0.8, top_k=256, seed=42, ignore_eos=True)
for n in (1, 2, 4):
    per = B // n
    lms = [fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=per).load_synthetic(0xF15E5EED) for _ in range(n)]
    res = [None] * n

    def work(i):
        ps = prompts[i * per:(i + 1) * per]
        res[i] = (lms[i].generate_static_batch(ps, frames + Lmax - 2, **kw), lms[i].last_stats())
    for rep in range(2):
        ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
    tot = sum(o.shape[1] for r in res for o in r[0])
    dec = max(r[1]["decode_ms"] for r in res)
    print(f"{n} x B={per}: {tot} frames in {dt*1e3:.1f} ms wall; slowest chain's decode {dec:.1f} ms -> {B*(frames-1)/(dec*1e-3):.0f} frames/s decode, "
          f"{dec*1e3/(frames-1):.0f} us per step of all 32 rows")
    for lm in lms: lm.close()
