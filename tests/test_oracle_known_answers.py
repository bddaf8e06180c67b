"""CPU: the oracle restatement vs the weight-free known answers derivable from the reference sources
(SURVEY.md §8c): FSQ implicit codebook, causal mask, repetition-penalty quirk trace, RoPE table, frame budget,
host-ArgMax tie rule, synthetic-weight generator spec."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

KA = np.load(os.path.join(os.path.dirname(__file__), "golden", "known_answers.npz"))


def test_fsq_codebook():  # fsq.rs:53-58,119-144
    c = orc.OracleCodec(tiny=True)
    cb = np.stack([c.fsq_code(i) for i in range(1000)])
    assert np.array_equal(cb, KA["fsq_codebook"])
    assert tuple(cb[0]) == (-1, -1, -1, -1) and tuple(cb[999]) == (0.75, 1, 1, 1)


def test_mask_abs():  # dual_ar.rs:702-712
    import ctypes as C
    m = np.zeros((3, 5), np.uint8)
    orc.lib().orc_get_mask_abs(3, 5, 8192, m.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert np.array_equal(m, KA["mask_3_5"])
    # sliding-window term: size1 + j + context < size2 + i
    m = np.zeros((1, 6), np.uint8)
    orc.lib().orc_get_mask_abs(1, 6, 2, m.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert m.tolist() == [[1, 1, 1, 0, 0, 0]]


def test_reppen_quirk_trace():  # rep_pen.rs:37-65
    import ctypes as C
    L = orc.lib()
    r = C.c_void_p(L.orc_reppen_create(16, 3, C.c_float(2.0)))
    for tok, expect in zip(KA["reppen_tokens"], KA["reppen_penalised"]):
        logits = np.full(16, 4.0, np.float32)
        logits[3] = -4.0  # division ignores the logit sign (rep_pen.rs:64)
        assert L.orc_reppen_apply(r, logits.ctypes.data_as(C.POINTER(C.c_float)), 16, int(tok)) == 0
        pen = sorted(np.nonzero(logits == 2.0)[0].tolist())
        assert pen == sorted(int(e) for e in expect if e >= 0), (tok, pen)
        assert logits[3] == -4.0
    # out-of-vocab token errors (rep_pen.rs:38-40)
    assert L.orc_reppen_apply(r, logits.ctypes.data_as(C.POINTER(C.c_float)), 16, 16) != 0
    L.orc_reppen_destroy(r)


def test_rope_table():  # dual_ar.rs:174-185
    lm = orc.OracleLM(orc.FISH15 | dict(n_layer=0, n_fast_layer=0, vocab_size=8, max_seq_len=64))
    cos, sin = lm.freqs()
    theta = KA["rope_theta_fish15"]
    pos = np.arange(64)[:, None]
    np.testing.assert_allclose(cos, np.cos(pos * theta), atol=1e-5)
    np.testing.assert_allclose(sin, np.sin(pos * theta), atol=1e-5)
    assert cos.shape == (64, 32) and cos[0].tolist() == [1.0] * 32


def test_argmax_tie_rule():  # host ArgMax: max_by(total_cmp) -> last maximal index
    import ctypes as C
    L = orc.lib()
    s = C.c_void_p(L.orc_sampler_create(C.c_uint64(1), C.c_double(0.0), C.c_double(1.0), C.c_uint64(0)))
    v = np.array([0.5, 2.0, -1.0, 2.0, 1.0], np.float32)
    assert L.orc_sampler_sample(s, v.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(5)) == 3
    L.orc_sampler_destroy(s)


@pytest.mark.parametrize("L,M,p0,frames", [tuple(r) for r in KA["budget_cases"].tolist()])
def test_frame_budget(L, M, p0, frames):  # single_batch.rs:61,77,193-197
    cfg = orc.TINY | dict(max_seq_len=512)
    lm = orc.OracleLM(cfg).load_synthetic(1)
    prompt = np.zeros((9, L), np.uint32)
    prompt[0] = np.arange(L) % cfg["im_end_id"]
    out = lm.generate(prompt, M, ignore_eos=True)
    assert out.shape == (8, frames)
    assert lm.kv_len() == L + frames - 1


def test_synth_generator_spec():
    """oracle/fsgen.h vs the numpy statement of the same spec (tests/golden/make_golden.py::synth)."""
    M64 = (1 << 64) - 1

    def fnv(name):
        h = 0xCBF29CE484222325
        for b in name.encode():
            h = ((h ^ b) * 0x100000001B3) & M64
        return h

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    for name, seed, mean, std, bf16 in [("layers.0.attention.wqkv.weight", 0xF15E5EED, 0.0, 0.02, False),
                                        ("norm.weight", 7, 1.0, 0.1, True)]:
        got = orc.synth(name, 257, seed, mean, std, bf16)
        key = fnv(name) ^ seed
        exp = np.empty(257, np.float32)
        for i in range(257):
            h = mix((key + (i + 1) * 0x9E3779B97F4A7C15) & M64)
            s = (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48) - 131070
            v = np.float32(mean) + np.float32(s) * np.float32(std / 37837.2272)
            if bf16:
                u = int(np.float32(v).view(np.uint32))
                u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
                v = np.uint32(u).view(np.float32)
            exp[i] = v
        assert np.array_equal(got, exp)
    big = orc.synth("x", 200000, 3, 0.0, 1.0)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01


@pytest.mark.parametrize("shape", [(1, 2, 1, 64), (1, 2, 166, 64)])
def test_repeat_kv_is_head_index_division(shape):  # lm/ops/repeat_kv.rs:126-162 (the reference's only kernel-level test)
    """RepeatKV(n_rep = 8) == cat-reshape == "query head h reads kv head h // n_rep" (how every fishrt attention kernel and the
    oracle index the cache instead of materialising the repeat), at the two shapes the reference tests."""
    rng = np.random.RandomState(0)
    x = rng.standard_normal(shape).astype(np.float32)
    b, hk, t, d = shape
    n_rep = 8
    cat_reshape = np.concatenate([x[:, :, None]] * n_rep, axis=2).reshape(b, hk * n_rep, t, d)  # dual_ar.rs:336-357 fallback
    by_index = x[:, np.arange(hk * n_rep) // n_rep]
    assert np.array_equal(cat_reshape, by_index)
    assert np.array_equal(cat_reshape, np.repeat(x, n_rep, axis=1))


def test_rescale_semantic_tokens_arithmetic():  # generate/utils.rs:36-56 (contiguous Fish-1.5 case :45-46) + :13-16
    """Sampling happens over logits[im_end_id:], so index 0 is <|im_end|> and index k is <|semantic:k-1|> = semantic_start + k - 1;
    with im_end_id == semantic_start_id - 1 both branches of the reference collapse to idx + im_end_id."""
    im_end, sem0 = 100011, 100012
    idx = np.array([0, 1, 2, 1024], np.int64)
    generic = np.where(idx == 0, im_end, idx - 1 + sem0)  # the reference's non-contiguous branch
    assert np.array_equal(idx + im_end, generic)


def test_default_voice_fixture_properties():  # voices-template/default.npy
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "default_voice_codes.npy"))
    assert v.shape == (8, 274) and v.dtype == np.int64
    assert v.min() >= 3 and v.max() <= 999  # FSQ indices of the (8, 5, 5, 5) levels: < 1000


# ---- the sampler's RNG chain (rand 0.8.5 StdRng = ChaCha12 behind BlockRng; rand_core seed_from_u64; WeightedIndex<f32>):
# public vectors -> independent pure-Python restatement (tests/golden/rng_ref.py) -> oracle/'s C++ (which the HIP sampler is tested against)
import ctypes as _C
import sys as _sys

_sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import rng_ref  # noqa: E402

TC1 = {  # "Test Vectors for the Stream Cipher ChaCha" (Strombergson draft), TC1: all-zero 256-bit key and IV, keystream block 0
    8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42",
    12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be",
    20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586",
}


def test_python_rng_reference_matches_public_vectors():
    for rounds, hexs in TC1.items():
        assert rng_ref.keystream_hex([0] * 8, rounds) == hexs
    # pcg32 demo (pcg-random.org, pcg32_srandom_r(42, 54)): pins the LCG multiplier and the XSH-RR output permutation
    assert rng_ref.pcg32_demo(42, 54, 6) == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]


def _orc_block(key, counter):
    out = (_C.c_uint32 * 16)()
    orc.lib().orc_rng_chacha12_block((_C.c_uint32 * 8)(*key), _C.c_uint64(counter), out)
    return list(out)


def test_oracle_chacha12_block_known_answer():
    import struct
    assert struct.pack("<16I", *_orc_block([0] * 8, 0)).hex() == TC1[12]
    # 64-bit block counter in words 12-13 (rand_chacha), arbitrary keys: against the Python reference
    rs = np.random.RandomState(5)
    for ctr in (1, 2, 0xFFFFFFFF, 0x100000000, 0x123456789ABC):
        key = [int(v) for v in rs.randint(0, 2 ** 32, 8, dtype=np.uint64)]
        assert _orc_block(key, ctr) == rng_ref.chacha_block(key, ctr, 12)


@pytest.mark.parametrize("seed", [0, 1, 42, 0xF15E5EED, 2 ** 63 + 12345, 2 ** 64 - 1])
def test_oracle_seed_from_u64_and_stream(seed):
    key = (_C.c_uint32 * 8)()
    orc.lib().orc_rng_seed_key(_C.c_uint64(seed), key)
    assert list(key) == rng_ref.seed_from_u64(seed)
    # next_u32 x 61, then next_u64 across the 64-word buffer boundary (index 61, 63 -> straddles two buffers), then more
    n32, n64 = 61, 70
    o32, o64 = (_C.c_uint32 * n32)(), (_C.c_uint64 * n64)()
    orc.lib().orc_rng_stream(_C.c_uint64(seed), n32, o32, n64, o64)
    r = rng_ref.StdRng(seed)
    assert list(o32) == [r.next_u32() for _ in range(n32)]
    assert list(o64) == [r.next_u64() for _ in range(n64)]


def test_oracle_weighted_index_matches_python_reference():
    rs = np.random.RandomState(11)
    cases = [rs.rand(7).astype(np.float32), rs.rand(256).astype(np.float32) ** 8, np.array([0, 0, 1, 0, 2, 0], np.float32),
             np.array([1e-30, 1.0, 1e-30], np.float32), (rs.rand(1024) * (rs.rand(1024) > 0.7)).astype(np.float32) + np.float32(1e-12)]
    for i, w in enumerate(cases):
        draws = 300
        out = (_C.c_uint32 * draws)()
        orc.lib().orc_rng_weighted_index(_C.c_uint64(1000 + i), w.ctypes.data_as(_C.POINTER(_C.c_float)), len(w), draws, out)
        r = rng_ref.StdRng(1000 + i)
        assert list(out) == [rng_ref.weighted_index_sample(r, [float(v) for v in w]) for _ in range(draws)]
        assert all(w[j] > 0 for j in out)
