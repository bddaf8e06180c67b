"""GPU: replica start-up (fs_lm_weights_arena / fs_lm_weights_adopt).  One process plays both sides: the
arena of a loaded handle is copied into an unloaded handle's arena through the same torch view fanout.broadcast_weights hands to
RCCL (CUDA array interface over the raw device pointer), then adopted; the receiver must generate the sender's tokens, persistent
kernels included (their weight images are derived at adopt time).  Runs in a subprocess: torch's HIP runtime has to be loaded first."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, "{root}/fish-speech.rs_amd")
import fishrt
from fishrt import config as fcfg, fanout
for dtype in ("bf16", "fp8"):
    a = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
    b = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype)
    p = np.zeros((9, 24), np.uint32); p[0] = np.random.RandomState(1).randint(0, 4000, 24)
    try:
        b.generate_blocking(p, 30)
        raise SystemExit("an unloaded handle generated")
    except RuntimeError as e:
        assert "not loaded" in str(e)
    (pa, na), (pb, nb) = a.weights_arena(), b.weights_arena()
    assert na == nb and na > (600 << 20 if dtype == "fp8" else 1200 << 20), (na, nb)
    ta = torch.as_tensor(fanout._DeviceBytes(pa, na), device="cuda")
    tb = torch.as_tensor(fanout._DeviceBytes(pb, nb), device="cuda")
    assert ta.data_ptr() == pa and tb.data_ptr() == pb      # views, not copies
    tb.copy_(ta); torch.cuda.synchronize()
    b.adopt_weights()
    kw = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    ca, cb = a.generate_blocking(p, 60, **kw), b.generate_blocking(p, 60, **kw)
    assert ca.shape == (8, 38) and np.array_equal(ca, cb), dtype
    assert a.last_stats()["kernels_per_frame"] == b.last_stats()["kernels_per_frame"]
    kw.update(temp=0.7, top_p=0.8, top_k=256)
    assert np.array_equal(a.generate_blocking(p, 40, seed=5, **kw), b.generate_blocking(p, 40, seed=5, **kw))
    assert fanout.broadcast_weights(None, b) == 0            # world 1: nothing to do
    a.close(); b.close()
print("ADOPT_OK")
'''


def test_adopted_arena_generates_the_senders_tokens():
    p = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ADOPT_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
