"""GPU: the folded static-batch decode step (round 5: layer-closing down projection with in-launch split-K sums, RMS in the Wqkv / head
epilogues, the first fast layer's norm in the sampler, no attention node on the first codebook pass) against the step it replaces.
The knob is read once per process, so both variants are subprocesses of tests/rows_fold_worker.py on the same seeded workload:
 * two runs of the folded step give bit-identical codes and captured logits (the in-launch split-K sums add in block order: deterministic);
 * FISHRT_ROWS_NO_FOLD=1 (split-K slabs + k_prep nodes: the round-4 step, still what > 32 rows run) sums the down projection in another order:
   the first frame's slow logits agree to 2e-3 absolute (f32-grade activations end to end) and the first frame's codes are identical wherever
   the captured top-2 margin exceeds that.
The oracle parity of the folded step itself is tests/test_batch_capture_gpu.py + tests/test_lm_gpu.py (they run it by default)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, cfg, dtype, B, frames, env):
    out = tmp_path / f"{name}.npz"
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "rows_fold_worker.py"), cfg, dtype, str(B), str(frames), str(out)],
                       capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("cfg,dtype,B", [("fish15", "bf16", 32), ("fish15", "fp8", 12), ("tiny", "bf16", 21), ("mid", "bf16", 32)])
def test_folded_step_is_deterministic_and_agrees_with_the_slab_path(tmp_path, cfg, dtype, B):
    F = 6
    a, sa = _run(tmp_path, "fold", cfg, dtype, B, F, {})
    b, sb = _run(tmp_path, "fold2", cfg, dtype, B, F, {})
    c, sc = _run(tmp_path, "nofold", cfg, dtype, B, F, {"FISHRT_ROWS_NO_FOLD": "1"})
    assert sa["graph_nodes_hint"] == "fold" and sc["graph_nodes_hint"] == "nofold"
    # deterministic: the exchange adds the K partials in block order whatever order they arrive in
    assert np.array_equal(a["codes"], b["codes"])
    assert np.array_equal(a["cap"], b["cap"])
    # another summation order of the down projection: first-frame slow logits (no sampled token upstream of them)
    n = int(a["n_slow"])
    la, lc = a["cap"][:, 0, 0, :n], c["cap"][:, 0, 0, :n]
    fin = np.isfinite(la) & np.isfinite(lc)
    worst = float(np.abs(la[fin] - lc[fin]).max())
    print(f"{cfg} {dtype} B={B}: fold vs slab path, frame-0 slow logits max |d| = {worst:.2e}")
    assert worst < 2e-3
    srt = np.sort(np.where(fin, la, -np.inf), axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 2 * 2e-3
    pa, pc = a["cap"][:, 0, 0, 2047], c["cap"][:, 0, 0, 2047]
    if a["greedy"]:
        assert np.array_equal(pa[clear], pc[clear])


def test_fold_is_gated_on_co_residency(tmp_path):
    """k_gemm_down's owner waves wait for units from the other blocks of the SAME launch, so the folded step is only taken where
    the occupancy query says the whole grid is resident.  FISHRT_DOWN_FAKE_CAPACITY pretends a device that holds 100 blocks (the Fish-1.5
    grid has 256): the step must fall back to the slab path -- bit for bit what FISHRT_ROWS_NO_FOLD=1 computes -- instead of spinning."""
    a, _ = _run(tmp_path, "small_dev", "fish15", "bf16", 8, 3, {"FISHRT_DOWN_FAKE_CAPACITY": "100"})
    b, _ = _run(tmp_path, "nofold", "fish15", "bf16", 8, 3, {"FISHRT_ROWS_NO_FOLD": "1"})
    c, _ = _run(tmp_path, "big_dev", "fish15", "bf16", 8, 3, {"FISHRT_DOWN_FAKE_CAPACITY": "256"})
    d, _ = _run(tmp_path, "fold", "fish15", "bf16", 8, 3, {})
    assert np.array_equal(a["codes"], b["codes"]) and np.array_equal(a["cap"], b["cap"])
    assert np.array_equal(c["codes"], d["codes"]) and np.array_equal(c["cap"], d["cap"])
    assert not np.array_equal(a["cap"], d["cap"])  # (the two steps sum the down projection in different orders)
