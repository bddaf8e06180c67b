"""GPU parity tests of the Firefly-GAN-VQ vocoder (fs_codec_decode) vs the CPU oracle and the committed goldens.
Tolerance: PCM within 1e-4 RMS (BASELINE.json north_star).  Three precision modes (fishrt.h: fs_codec_set_precision):
"f32" -- exact f32 products (f32 FMA / f32 MFMA chains), measured ~1e-6 of the oracle, asserted at 1e-6;
"bf16x3" -- split-bf16 matrix products, measured 3e-7 (full size), asserted at 2.5e-5;
"f16" (the default) -- single f16 operands, f32 accumulation: measured 1.6e-5 at signal rms 0.031 (full size), asserted at 4e-5."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
CG = np.load(os.path.join(G, "codec_tiny.npz"))


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.fixture(scope="module")
def tiny():
    return fishrt.FireflyCodec(0, channel_div=8, precision="f32").load_synthetic(int(CG["seed"]))


MATRIX_MODES = [("bf16x3", 2.5e-5), ("f16", 4e-5)]  # (mode, asserted PCM rms vs the f32 oracle / golden)


@pytest.fixture(scope="module", params=MATRIX_MODES, ids=[m for m, _ in MATRIX_MODES])
def tiny_bf3(request):
    c = fishrt.FireflyCodec(0, channel_div=8, precision=request.param[0]).load_synthetic(int(CG["seed"]))
    c.rms_tol = request.param[1]
    return c


def test_default_precision_is_f16():
    c = fishrt.FireflyCodec(0, channel_div=8)
    assert c.precision == "f16"
    c.close()


def test_tiny_matrix_modes_vs_golden_and_f32_mode(tiny, tiny_bf3):
    """the matrix-core precision modes on the tiny topology (its 64..16-channel convs run the split-bf16 / f16 kernel): within the
    mode's bound of the golden PCM, and a code prefix still decodes to the bit-identical PCM prefix (tile-shape independent
    summation order)"""
    assert tiny.precision == "f32"
    codes = CG["codes"]
    pcm = tiny_bf3.decode(np.ascontiguousarray(codes[None]))[0, 0]
    r = rms(pcm, CG["pcm"])
    print(f"tiny vocoder {tiny_bf3.precision}: PCM rms diff {r:.2e}")
    assert 0 < r < tiny_bf3.rms_tol and np.abs(pcm).max() <= 1.0
    rng = np.random.RandomState(5)
    long = rng.randint(0, 1000, (8, 70)).astype(np.uint32)
    full = tiny_bf3.decode(np.ascontiguousarray(long[None]))[0, 0]
    assert rms(full, tiny.decode(np.ascontiguousarray(long[None]))[0, 0]) < tiny_bf3.rms_tol
    for T1 in (1, 2, 33):
        part = tiny_bf3.decode(np.ascontiguousarray(long[None, :, :T1]))[0, 0]
        assert np.array_equal(part, full[: 2048 * T1]), T1


def test_tiny_vs_golden_and_oracle(tiny):
    codes = CG["codes"]
    pcm = tiny.decode(np.ascontiguousarray(codes[None]))
    assert pcm.shape == (1, 1, 2048 * codes.shape[1]) and tiny.sample_rate == 44100
    assert rms(pcm[0, 0], CG["pcm"]) < 1e-6
    o = orc.OracleCodec(tiny=True).load_synthetic(int(CG["seed"]))
    assert rms(pcm[0, 0], o.decode(codes)) < 1e-6
    assert np.abs(pcm).max() <= 1.0


def test_tiny_edge_cases(tiny):
    o = orc.OracleCodec(tiny=True).load_synthetic(int(CG["seed"]))
    rng = np.random.RandomState(3)
    for T in (1, 2, 33, 70):  # single frame, tile boundaries of the 128/256-sample conv tiles
        codes = rng.randint(0, 1000, (8, T)).astype(np.uint32)
        codes[:, 0] = [0, 999, 7, 8, 39, 40, 199, 200]  # FSQ level boundaries (fsq.rs:137-144)
        pcm = tiny.decode(np.ascontiguousarray(codes[None]))[0, 0]
        assert rms(pcm, o.decode(codes)) < 1e-6, T
    with pytest.raises(RuntimeError):
        tiny.decode(np.full((1, 8, 4), 1000, np.uint32))  # gather out of range
    with pytest.raises(ValueError):
        tiny.decode(np.zeros((1, 8, 8), np.uint32)[:, :, ::2])  # non-contiguous (codec.rs:97-101)


def test_batch_follows_reference_raw_reshape(tiny):
    """quantizer.rs:138-143: (b, g, t) is RESHAPED (not permuted) to (g, b, t, 1): group slot g' of batch item b' reads
    row g'*B + b' of the flattened (B*8, T) index matrix.  Identity for B == 1."""
    o = orc.OracleCodec(tiny=True).load_synthetic(int(CG["seed"]))
    rng = np.random.RandomState(11)
    B, T = 2, 5
    codes = rng.randint(0, 1000, (B, 8, T)).astype(np.uint32)
    pcm = tiny.decode(codes)
    flat = codes.reshape(B * 8, T)
    for b in range(B):
        eff = np.stack([flat[g * B + b] for g in range(8)])
        assert rms(pcm[b, 0], o.decode(eff)) < 1e-6


@pytest.fixture(scope="module")
def full():
    return fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)  # default precision: f16


@pytest.fixture(scope="module")
def full_bf3():
    return fishrt.FireflyCodec(0, precision="bf16x3").load_synthetic(0xC0DEC)


@pytest.fixture(scope="module")
def full_f32():
    return fishrt.FireflyCodec(0, precision="f32").load_synthetic(0xC0DEC)


def test_fullsize_vs_oracle(full, full_bf3, full_f32):
    o = orc.OracleCodec(tiny=False).load_synthetic(0xC0DEC)
    voice = np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32)  # (8, 274) in [3, 999]
    codes = np.ascontiguousarray(voice[:, :12])
    ref = o.decode(codes)
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    for name, h, tol in (("f16", full, 4e-5), ("bf16x3", full_bf3, 2.5e-5), ("f32", full_f32, 1e-5)):
        pcm = h.decode(codes[None])[0, 0]
        assert pcm.shape == ref.shape == (2048 * 12,)
        r = rms(pcm, ref)
        print(f"full-size vocoder [{name}]: PCM rms diff {r:.2e} at signal rms {sig:.3f}")
        assert r < tol and sig > 1e-3, name


def test_fullsize_matrix_modes_vs_f32_mode_long(full, full_bf3, full_f32):
    """the whole default voice (274 frames): the matrix-core modes stay within their bound of the exact-f32 mode (itself ~1e-6 of the
    oracle)"""
    voice = np.ascontiguousarray(np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32))
    b = full_f32.decode(voice[None])[0, 0]
    for h, tol in ((full, 4e-5), (full_bf3, 2.5e-5)):
        a = h.decode(voice[None])[0, 0]
        r = rms(a, b)
        print(f"full-size vocoder, 274 frames: {h.precision} vs f32 mode rms {r:.2e}, max {np.abs(a - b).max():.2e}")
        assert r < tol, h.precision


def test_fullsize_causality_property(full):
    """Every conv of the 1.4+/1.5 codec is causal (utils/mod.rs:53-62,110-122): the PCM of a code prefix is the prefix
    of the PCM, bit for bit (size-independent property at the full default-voice length, 274 frames = 12.7 s)."""
    voice = np.ascontiguousarray(np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32))
    pcm = full.decode(voice[None])[0, 0]
    assert pcm.shape == (2048 * 274,) and np.isfinite(pcm).all() and np.abs(pcm).max() <= 1.0
    for T1 in (1, 37, 200):
        part = full.decode(np.ascontiguousarray(voice[None, :, :T1]))[0, 0]
        assert np.array_equal(part, pcm[: 2048 * T1]), T1


def test_chunked_streaming_decode_is_bit_identical(full):
    """Streaming vocoder (BASELINE configs[4]): chunks of 64 frames with a 24-frame left halo reproduce the one-shot PCM
    bit for bit (causal convs, receptive field ~14.5 frames: fishrt/stream.py)."""
    from fishrt import decode_chunk
    voice = np.ascontiguousarray(np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32))
    ref = full.decode(voice[None])[0, 0]
    parts = [decode_chunk(full, voice, a, min(a + 64, 274)) for a in range(0, 274, 64)]
    assert np.array_equal(np.concatenate(parts), ref)
    # a halo shorter than the receptive field is NOT exact (the bound is real)
    short = decode_chunk(full, voice, 128, 192, halo=4)
    assert not np.array_equal(short, ref[2048 * 128: 2048 * 192])


def test_overlapped_lm_vocoder_pipeline(tiny):
    """LM generation with the vocoder consuming 16-frame chunks in a worker thread == generate then decode."""
    from fishrt import StreamingSynth, config as fcfg
    lm = fishrt.DualARTransformer(fcfg.TINY, fcfg.TINY_TOKENS, 0, "f32").load_synthetic(7)
    p = np.zeros((9, 6), np.uint32)
    p[0] = [1, 2, 3, 4, 5, 6]
    synth = StreamingSynth(lm, tiny, chunk=16)
    codes, pcm = synth(p, 80, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    codes = np.minimum(codes, 999)  # tiny LM codebook (64) is already < 1000
    lm.clear_slow_layer_caches()
    ref_codes = lm.generate_blocking(p, 80, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    assert np.array_equal(codes, ref_codes) and codes.shape == (8, 76)
    ref_pcm = tiny.decode(np.ascontiguousarray(ref_codes[None]))[0, 0]
    assert pcm.shape == ref_pcm.shape and np.array_equal(pcm, ref_pcm)
    assert synth.stats["frames"] == 76


def test_f16_range_guard_counts_and_falls_back(tmp_path):
    """fs_codec_set_range_check: the f16 mode saturates beyond +-65504, loses operands below 2^-24 and has a relative
    error against an absolute bound.  With the check on, decode() counts out-of-range operands in the f32 -> f16 conversions (range-counting
    kernel twins), decodes in bf16x3 mode as well, and answers from bf16x3 when an operand saturated or the two PCMs are more than 5e-5 RMS
    apart.  Four checkpoints from the synthetic generator (tiny topology): as is; conv_pre and ups.0 x 1e3 (activations saturate); conv_post x 40 (loud
    output: the relative error crosses the absolute bound); one ResBlock conv x 1e-10 (weights flush to zero: reported, harmless)."""
    import test_safetensors_gpu as tsf
    codes = np.random.RandomState(3).randint(0, 1000, (1, 8, 12)).astype(np.uint32)

    def codec_from(t, precision="f16"):
        path = str(tmp_path / "c.safetensors")
        tsf._save(t, path, False)
        return fishrt.FireflyCodec(0, channel_div=8, precision=precision).load_safetensors(path)

    def variant(**scale):
        t = {k: v.copy() for k, v in base.items()}
        for k, f in scale.items():
            t[k.replace("__", ".")] *= np.float32(f)
        return t

    base = tsf._codec_tensors(64, 1234)
    # (a) in range: nothing saturates, no fall-back, and the checked kernels produce the unchecked kernels' PCM bit for bit
    c = codec_from(base)
    ref = c.decode(codes)
    c.set_range_check(True)
    got = c.decode(codes)
    st = c.range_stats()
    assert st["act_saturated"] == 0 and st["weights_saturated"] == 0 and st["weights_flushed"] == 0 and st["fallbacks"] == 0, st
    assert st["act_flushed"] < 100 and 0 < st["last_pcm_rms_diff"] < 4e-5, st  # (SiLU tails; the f16-vs-bf16x3 distance of an ordinary signal)
    assert np.array_equal(got, ref)
    c.set_range_check(False)
    assert np.array_equal(c.decode(codes), ref)
    c.close()
    # (b) activations beyond 65504 (two convs x 1000 each; the weights themselves stay far inside the range): counted, answered from bf16x3
    # (== a bf16x3 handle's PCM); the unguarded f16 PCM is something else
    t = variant(head__conv_pre__conv__weight=1e3, head__ups__0__conv__weight=1e3)
    c = codec_from(t)
    unguarded = c.decode(codes)
    c.set_range_check(True)
    got = c.decode(codes)
    st = c.range_stats()
    assert st["act_saturated"] > 0 and st["fallbacks"] == 1 and st["weights_saturated"] == 0, st
    c3 = codec_from(t, "bf16x3")
    exp = c3.decode(codes)
    assert np.isfinite(got).all() and np.array_equal(got, exp) and not np.array_equal(unguarded, exp)
    c.decode(codes)
    assert c.range_stats()["fallbacks"] == 2
    c.close(); c3.close()
    # (c) loud output (pre-tanh x 40): nothing saturates, but f16's relative error is now > 5e-5 absolute -> bf16x3 answers
    t = variant(head__conv_post__conv__weight=40.0)
    c = codec_from(t).set_range_check(True)
    got = c.decode(codes)
    st = c.range_stats()
    assert st["act_saturated"] == 0 and st["last_pcm_rms_diff"] > 5e-5 and st["fallbacks"] == 1, st
    c3 = codec_from(t, "bf16x3")
    assert np.array_equal(got, c3.decode(codes)) and float(np.sqrt(np.mean(got.astype(np.float64) ** 2))) > 0.2
    c.close(); c3.close()
    # (d) weights below 2^-24: found when the check is switched on, reported, and (absolute error < 6e-8 each) not a reason to fall back
    c = codec_from(variant(head__resblocks__1__blocks__0__convs1__0__conv__weight=1e-10)).set_range_check(True)
    st = c.range_stats()
    assert st["weights_flushed"] > 0 and st["weights_saturated"] == 0, st
    c.decode(codes)
    st = c.range_stats()
    assert st["fallbacks"] == 0 and st["last_pcm_rms_diff"] < 4e-5, st
    c.close()


def test_hard_checkpoint_weight_norm_scales_vs_f32_oracle(tmp_path):
    """the vocoder's f16 operand mode has only seen N(0, 1 / fan_in) weights.  A real Firefly checkpoint is weight-normed and
    "must be pre-merged" (codec/utils/mod.rs:28-40): after merging g / ||v|| every output channel carries its own gain.  This builds that kind
    of checkpoint: per-output-channel gains log-uniform in [0.05, 20] on the first conv of every ResBlock pair and on conv_pre (weights AND
    biases), the inverse gain on the matching input channel of the conv that consumes it (so the network's function stays in range while its
    intermediate planes span 400 x in scale, channel by channel), biases of size 0.2, and code indices at the extremes 0 / 999 among random
    ones.  Contract: with the range guard on, decode() answers within 1e-4 RMS of the f32 oracle -- from the f16 mode, or from the bf16x3 mode
    it falls back to (server/lib/utils/load.rs:161-164 runs the codec in f32; the bound is the north star's)."""
    import test_safetensors_gpu as tsf
    rng = np.random.RandomState(11)
    t = {k: v.copy() for k, v in tsf._codec_tensors(64, 1234).items()}
    gains = lambda n: np.exp(rng.uniform(np.log(0.05), np.log(20.0), n)).astype(np.float32)
    for k in list(t):
        if k.endswith(".bias") and k.startswith("head."):
            t[k] = (t[k] * 10.0).astype(np.float32)  # std 0.02 -> 0.2
    n_pairs = 0
    for s_ in range(5):
        for j in range(3):
            for m in range(3):
                q = f"head.resblocks.{s_}.blocks.{j}"
                w1, b1, w2 = q + f".convs1.{m}.conv.weight", q + f".convs1.{m}.conv.bias", q + f".convs2.{m}.conv.weight"
                g = gains(t[w1].shape[0])
                t[w1] = t[w1] * g[:, None, None]            # Conv1d [out, in, k]: output channel gains
                t[b1] = t[b1] * g
                t[w2] = t[w2] / g[None, :, None]            # the consumer's input channels
                n_pairs += 1
    g = gains(t["head.conv_pre.conv.weight"].shape[0])
    t["head.conv_pre.conv.weight"] *= g[:, None, None]
    t["head.conv_pre.conv.bias"] *= g
    t["head.ups.0.conv.weight"] /= g[:, None, None]          # ConvTranspose1d [in, out, k]
    assert n_pairs == 45
    codes = rng.randint(0, 1000, (1, 8, 16)).astype(np.uint32)
    codes[0, :, 0] = 0; codes[0, :, 5] = 999; codes[0, ::2, 9] = 0; codes[0, 1::2, 9] = 999
    ref = orc.OracleCodec(tiny=True).load_synthetic(1234).set_tensors({k: v for k, v in t.items() if k.startswith(("head.", "quantizer.upsample", "quantizer.residual_fsq")) and "project_in" not in k}).decode(codes[0])
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    path = str(tmp_path / "hard.safetensors")
    tsf._save(t, path, False)
    rms = lambda a: float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2)))
    c32 = fishrt.FireflyCodec(0, channel_div=8, precision="f32").load_safetensors(path)
    e32 = rms(c32.decode(codes)[0, 0])
    c32.close()
    c = fishrt.FireflyCodec(0, channel_div=8, precision="f16").load_safetensors(path)
    e16_unguarded = rms(c.decode(codes)[0, 0])
    c.set_range_check(True)
    got = c.decode(codes)[0, 0]
    st = c.range_stats()
    c.close()
    c3 = fishrt.FireflyCodec(0, channel_div=8, precision="bf16x3").load_safetensors(path)
    e3 = rms(c3.decode(codes)[0, 0])
    c3.close()
    print(f"hard checkpoint: signal rms {sig:.3f}; rms error vs the f32 oracle: f32 mode {e32:.2e}, bf16x3 {e3:.2e}, f16 unguarded {e16_unguarded:.2e}, "
          f"guarded answer {rms(got):.2e} (fallbacks {st['fallbacks']}, act_saturated {st['act_saturated']}, act_flushed {st['act_flushed']}, "
          f"weights_saturated {st['weights_saturated']}, weights_flushed {st['weights_flushed']}, f16-vs-bf16x3 {st['last_pcm_rms_diff']:.2e})")
    assert np.isfinite(got).all() and e32 < 2e-6 and e3 < 2.5e-5
    assert rms(got) < 1e-4, "the guarded decode left the 1e-4 RMS bound on a weight-norm-scaled checkpoint"
    # which mode answered is data; that the guard's rule was followed is the check: a fall-back answer IS the bf16x3 PCM
    if st["fallbacks"]:
        assert abs(rms(got) - e3) < 1e-9
    else:
        assert st["act_saturated"] == 0 and st["last_pcm_rms_diff"] <= 5e-5


def test_range_check_switched_on_before_load_still_scans_the_weights(tmp_path):
    """fs_codec_set_range_check(1) BEFORE the weights exist used to leave the weight counters at zero for good (the scan ran only inside
    set_range_check).  The pack path now scans when the check is already on."""
    import test_safetensors_gpu as tsf
    t = {k: v.copy() for k, v in tsf._codec_tensors(64, 1234).items()}
    t["head.resblocks.1.blocks.0.convs1.0.conv.weight"] *= np.float32(1e-10)
    path = str(tmp_path / "c.safetensors")
    tsf._save(t, path, False)
    c = fishrt.FireflyCodec(0, channel_div=8, precision="f16")
    c.set_range_check(True)
    c.load_safetensors(path)
    st = c.range_stats()
    assert st["weights_flushed"] > 0 and st["weights_saturated"] == 0, st
    c.close()
