"""GPU: the on-device samplers at PRODUCTION size (BASELINE.json configs[2]: n = 2037 slow candidates / 1024 codebook entries, top-k 256,
top-p 0.8, temp 0.7) against the CPU oracle, whose RNG chain is pinned to public vectors (tests/test_oracle_known_answers.py).
 (1) the static-batch sampler in isolation on identical logits (fs_selftest_sample_rows vs oracle batched_sample): token-exact;
 (2) the batch-1 sampler end to end on an f32 Fish-1.5 handle (f32 logits agree with the oracle to ~1e-6, so the sampled streams agree
     except where a uniform draw lands within that distance of a CDF boundary)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import _ffi, config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED


def _gpu_rows(logits, temp, top_p, top_k, seed, call):
    B, n = logits.shape
    out = np.zeros(B, np.uint32)
    s = _ffi.Sampling(float(temp), float(top_p), int(top_k), 1.0)
    _ffi.check(_ffi.lib().fs_selftest_sample_rows(0, logits.ctypes.data_as(C.POINTER(C.c_float)), B, n, C.byref(s), C.c_uint64(seed), call,
                                                  out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def _orc_rows(logits, temp, top_p, top_k, seed, call):
    B, n = logits.shape
    out = np.zeros(B, np.uint32)
    orc.lib().orc_batched_sample(C.c_uint64(seed), C.c_double(temp), C.c_double(top_p), C.c_uint64(top_k),
                                 logits.ctypes.data_as(C.POINTER(C.c_float)), B, n, call, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


@pytest.fixture(params=["block1024", "par512"])
def impl(request):
    """block1024: the static-batch sampler kernel (one-wave top-k path inside a 1024-thread block); par512: the block-parallel sampler the
    persistent fast decoder runs in-launch (csrc/lm_bsample_dev.h), 512 threads per row, same RNG derivation"""
    old = os.environ.get("FISHRT_SAMPLER_IMPL")
    if request.param == "par512":
        os.environ["FISHRT_SAMPLER_IMPL"] = "par512"
    else:
        os.environ.pop("FISHRT_SAMPLER_IMPL", None)
    yield request.param
    if old is None:
        os.environ.pop("FISHRT_SAMPLER_IMPL", None)
    else:
        os.environ["FISHRT_SAMPLER_IMPL"] = old


@pytest.mark.parametrize("n", [2037, 1024])
@pytest.mark.parametrize("temp,top_p,top_k", [(0.7, 0.8, 256), (0.7, 0.9, 50), (1.0, 1.0, 256), (0.7, 0.3, 256), (1e-8, 0.8, 256), (0.7, 0.8, 1), (2.0, 0.99, 255)])
def test_static_batch_sampler_rows_token_exact(n, temp, top_p, top_k, impl):
    rs = np.random.RandomState(n + top_k)
    total = agree = 0
    for call in (0, 1, 9, 300):
        for scale in (1.0, 4.0):  # flat (synthetic-weight-like) and peaked (trained-model-like) logit rows
            logits = np.ascontiguousarray((rs.randn(32, n) * scale).astype(np.float32))
            logits[3, 5] = logits[3, 900] = logits[3].max() + 1.0  # an exact tie at the top (first-max rule when temp <= 1e-7)
            logits[4, 100:400] = logits[4, 100]                    # 300 equal candidates straddling the top-k boundary (ties: lower index first)
            logits[5, 7:] = -80.0                                  # fewer non-zero probabilities than top_k (the rest underflow to 0)
            g, o = _gpu_rows(logits, temp, top_p, top_k, 42, call), _orc_rows(logits, temp, top_p, top_k, 42, call)
            assert np.array_equal(g, o), (call, scale, np.nonzero(g != o)[0], g[g != o], o[g != o])
            total += 32
            agree += int((g == o).sum())
    print(f"n={n} temp={temp} top_p={top_p} top_k={top_k}: {agree}/{total} rows identical")


def test_fast_tail_guards_token_exact_in_volume():
    """Round 6: the block-parallel sampler takes its three cumulative sums as PARALLEL prefix scans and accepts a threshold comparison only
    where the scan stays a guard band (6.2e-5 / 1.3e-4 x total: twice the worst-case distance between two summation orders of 256 terms) away
    from the threshold; inside the band the call falls through to the reference's sequential chains (lm_bsample_dev.h, "fast tail").  On flat
    rows a few per cent of the draws land in the band, so BOTH tails -- and the boundary between them -- are exercised here in volume:
    12 800 rows per setting, every pick identical to the oracle's sequential sampler."""
    old = os.environ.get("FISHRT_SAMPLER_IMPL")
    os.environ["FISHRT_SAMPLER_IMPL"] = "par512"
    try:
        rs = np.random.RandomState(99)
        for temp, top_p, top_k, n in ((0.7, 0.8, 256, 1024), (0.7, 0.5, 256, 2037), (1.0, 0.9, 128, 1024), (0.9, 0.97, 256, 1024), (0.7, 0.999, 256, 1024)):
            bad = total = 0
            for call in range(25):
                for scale in (1.0, 2.5):
                    logits = np.ascontiguousarray((rs.randn(256, n) * scale).astype(np.float32))
                    g, o = _gpu_rows(logits, temp, top_p, top_k, 7, call), _orc_rows(logits, temp, top_p, top_k, 7, call)
                    bad += int((g != o).sum())
                    total += 256
            print(f"temp={temp} top_p={top_p} top_k={top_k} n={n}: {total - bad}/{total} rows identical")
            assert bad == 0
    finally:
        if old is None:
            os.environ.pop("FISHRT_SAMPLER_IMPL", None)
        else:
            os.environ["FISHRT_SAMPLER_IMPL"] = old


def test_batch_path_compares_top_p_in_f64(impl):
    """sampling/mod.rs:68: `top_p >= sum_p as f64` with top_p kept as f64 -- for a top_p a hair below the f32 sum of the kept probabilities
    the f64 rule takes the top-p branch where an f32 comparison (top_p rounds up to the sum) would draw from all k.  A sweep of top_p values
    1e-9 apart across that sum: device and oracle must make the same choice at every one of them."""
    rs = np.random.RandomState(5)
    logits = np.ascontiguousarray((rs.randn(4, 1024) * 2.0).astype(np.float32))
    p = np.exp((logits[0] / 0.7).astype(np.float64)); p /= p.sum()
    s256 = float(np.sort(p)[-256:].sum())  # ~ the f32 sum the sampler compares with
    for k in range(-40, 41):
        top_p = s256 + k * 2.5e-9
        g, o = _gpu_rows(logits, 0.7, top_p, 256, 42, 3), _orc_rows(logits, 0.7, top_p, 256, 42, 3)
        assert np.array_equal(g, o), (k, top_p, g, o)


def test_fish15_f32_sampled_stream_vs_oracle():
    """BASELINE configs[2] sampling at batch 1 on the f32 handle: 20 frames x 9 draws (slow n = 2037, fast n = 1024, top-k 256)"""
    cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
    lm = fishrt.DualARTransformer(cfg, tok, 0, "f32").load_synthetic(SEED)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=False)
    rs = np.random.RandomState(4)
    p = np.zeros((9, 12), np.uint32)
    p[0] = rs.randint(0, tok["im_end_id"], 12)
    for seed, kw in ((42, dict(temp=0.7, top_p=0.8, top_k=256)), (7, dict(temp=1.0, top_p=0.95, top_k=64))):
        lm.clear_slow_layer_caches(); o.clear_slow()
        got = lm.generate_blocking(p, 12 + 18, repetition_penalty=1.2, seed=seed, ignore_eos=True, **kw)
        exp = o.generate(p, 12 + 18, repetition_penalty=1.2, seed=seed, ignore_eos=True, **kw)
        assert got.shape == exp.shape == (8, 20)
        bad = np.nonzero((got != exp).any(0))[0]
        first = int(bad[0]) if bad.size else 20
        print(f"seed {seed} {kw}: {first}/20 frames identical")
        assert first >= 16, (seed, first)
    lm.close()
