"""CPU: the C++ oracle restatement vs the committed second-implementation goldens (tests/golden/make_golden.py,
PyTorch f32).  Tolerances: f32 accumulation-order noise only."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
LMG = np.load(os.path.join(G, "lm_tiny.npz"))
CG = np.load(os.path.join(G, "codec_tiny.npz"))
TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module", params=["f32w", "bf16w"])
def lm_tag(request):
    tag = request.param
    lm = orc.OracleLM(orc.TINY).load_synthetic(int(LMG["seed"]), bf16=(tag == "bf16w"))
    return lm, tag


def test_prefill_and_decode(lm_tag):
    lm, tag = lm_tag
    lm.clear_slow()
    logits, hidden = lm.forward_generate(LMG["prompt"], 0)
    np.testing.assert_allclose(logits, LMG[f"{tag}_prefill_logits"], **TOL)
    np.testing.assert_allclose(hidden, LMG[f"{tag}_prefill_hidden"], **TOL)
    assert lm.kv_len() == LMG["prompt"].shape[1]
    l2, h2 = lm.forward_generate(LMG["decode_step_tokens"], lm.kv_len())
    np.testing.assert_allclose(l2, LMG[f"{tag}_decode_logits"], **TOL)
    np.testing.assert_allclose(h2, LMG[f"{tag}_decode_hidden"], **TOL)
    lm.clear_fast()
    fe = lm.fast_embeddings()
    f0 = lm.forward_generate_fast(h2, 0)
    f1 = lm.forward_generate_fast(fe[11], 1)
    f2 = lm.forward_generate_fast(fe[50], 2)
    np.testing.assert_allclose(np.concatenate([f0, f1, f2]), LMG[f"{tag}_fast_logits"], **TOL)


def test_chunked_prefill_with_cached_prefix(lm_tag):
    lm, tag = lm_tag
    lm.clear_slow()
    p = LMG["prompt"]
    lm.forward_generate(p[:, :5], 0)
    l3, _ = lm.forward_generate(p[:, 5:], 5)
    np.testing.assert_allclose(l3, LMG[f"{tag}_chunked_logits"], **TOL)
    # truncate-to-prefix (dual_ar.rs:392-404) then re-run the suffix: same result
    lm.clear_slow_until(5)
    assert lm.kv_len() == 5
    l4, _ = lm.forward_generate(p[:, 5:], 5)
    np.testing.assert_array_equal(l3, l4)


def test_batched_prefill_ignores_pad_mask(lm_tag):
    lm, tag = lm_tag
    lm.clear_slow()
    lb, hb = lm.forward_generate(LMG["batch2_prompt"], 0)
    np.testing.assert_allclose(lb, LMG[f"{tag}_batch2_logits"], **TOL)
    np.testing.assert_allclose(hb, LMG[f"{tag}_batch2_hidden"], **TOL)


@pytest.mark.parametrize("rp", [1.0, 1.2])
def test_greedy_rollout_token_exact(lm_tag, rp):
    lm, tag = lm_tag
    lm.clear_slow()
    p = LMG["prompt"]
    out = lm.generate(p, 24 + p.shape[1] - 2, temp=0.0, repetition_penalty=rp, ignore_eos=True)
    assert np.array_equal(out, LMG[f"{tag}_rollout_rp{int(rp * 10)}"])


@pytest.mark.parametrize("name", ["s1", "s2", "s3"])
def test_sampled_rollout_token_exact(name):
    """temp / top-k / top-p / WeightedIndex / StdRng: the oracle's sampled token stream against the fixture produced by the independent
    PyTorch model + pure-Python sampler and RNG (make_golden.py PySampler, rng_ref.py; the RNG itself is pinned to public vectors in
    test_oracle_known_answers.py).  f32 weights; 24 frames x 9 draws."""
    seed, temp, top_p, top_k = LMG[f"sampled_{name}_cfg"]
    lm = orc.OracleLM(orc.TINY).load_synthetic(int(LMG["seed"]), bf16=False)
    p = LMG["prompt"]
    out = lm.generate(p, 24 + p.shape[1] - 2, temp=float(temp), top_p=float(top_p), top_k=int(top_k), repetition_penalty=1.2,
                      seed=int(seed), ignore_eos=True)
    assert np.array_equal(out, LMG[f"sampled_{name}_rollout"])


def test_partial_head_equals_full_head(lm_tag):
    lm, _ = lm_tag
    lm.clear_slow()
    a, _ = lm.forward_generate(LMG["prompt"], 0, full_head=True)
    lm.clear_slow()
    b, _ = lm.forward_generate(LMG["prompt"], 0, full_head=False)
    lo = orc.TINY["im_end_id"]
    np.testing.assert_array_equal(a[:, lo:], b[:, lo:])


def test_codec_tiny_stages_and_pcm():
    c = orc.OracleCodec(tiny=True).load_synthetic(int(CG["seed"]))
    codes = CG["codes"]
    assert c.hop == 2048
    pcm = c.decode(codes)
    assert pcm.shape == (2048 * codes.shape[1],)
    rms = float(np.sqrt(np.mean((pcm - CG["pcm"]) ** 2)))
    assert rms < 1e-6, rms
    for i in range(9):
        exp = CG[f"stage{i}"]
        _, st = c.decode(codes, stage=i, stage_size=exp.size)
        np.testing.assert_allclose(st.reshape(exp.shape), exp, rtol=1e-4, atol=2e-5)


def test_static_batch_rows_equal_padded_single_under_greedy(lm_tag):
    """static_batch.rs: rows are independent sequences; the left padding is NOT masked (dual_ar.rs:589-615) and batch
    rep-pen is a no-op (static_batch.rs:204-206), so under greedy decoding row i == generate_blocking(padded prompt i)."""
    lm, _ = lm_tag
    rng = np.random.RandomState(21)
    prompts = []
    for L in (4, 9, 6):
        p = np.zeros((9, L), np.uint32)
        p[0] = rng.randint(0, 400, L)
        prompts.append(p)
    outs = lm.generate_batch(prompts, 20, temp=0.0, ignore_eos=True)
    assert [o.shape for o in outs] == [(8, 13)] * 3  # frames = M - Lmax + 2
    for p, got in zip(prompts, outs):
        pad = 9 - p.shape[1]
        pp = np.concatenate([np.zeros((9, pad), np.uint32), p], 1)
        pp[0, :pad] = 400
        lm.clear_slow()
        exp = lm.generate(pp, 20, temp=0.0, repetition_penalty=1.0, ignore_eos=True)
        assert np.array_equal(got, exp)
    # sampled: reproducible per seed, different across seeds
    a = lm.generate_batch(prompts, 20, temp=0.8, top_p=0.9, top_k=32, seed=42, ignore_eos=True)
    b = lm.generate_batch(prompts, 20, temp=0.8, top_p=0.9, top_k=32, seed=42, ignore_eos=True)
    c = lm.generate_batch(prompts, 20, temp=0.8, top_p=0.9, top_k=32, seed=43, ignore_eos=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not all(np.array_equal(x, y) for x, y in zip(a, c))
