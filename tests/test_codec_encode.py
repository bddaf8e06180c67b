"""FireflyCodec::encode: the CPU oracle against the independent numpy/PyTorch restatement committed as
tests/golden/codec_enc_tiny.npz (make_golden.py; its mel front-end uses numpy's f64 FFT and the REFERENCE's own embedded mel
table), and the GPU path against the oracle.  Tolerances: log-mel 5e-5 abs (f32 sum order + the regenerated filterbank,
max |table diff| 1.8e-7); activations 2e-5; FSQ indices identical except where the pre-round value sits within 1e-3 of a
rounding boundary (the oracle reports that margin)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "codec_enc_tiny.npz"))
SEED = int(G["seed"])


@pytest.fixture(scope="module")
def otiny():
    c = orc.OracleCodec(tiny=True)
    c.load_synthetic(SEED)
    return c


def test_mel_table_matches_reference_sample():
    fb = orc.mel_filterbank().reshape(-1)
    assert np.abs(fb[G["mel_table_idx"]] - G["mel_table_val"]).max() < 2.5e-7
    assert fb.shape == (1025 * 160,)


def test_oracle_log_mel_and_frame_count(otiny):
    mel = otiny.log_mel(G["pcm"])
    assert mel.shape == G["mel"].shape
    np.testing.assert_allclose(mel, G["mel"], atol=5e-5, rtol=0)
    # frames the streaming STFT emits (stft.rs:52-90): padded length n + 1536, one frame per hop once 2048 samples are in
    for n, exp in ((768, 2), (2048, 4), (512 * 10, 10), (512 * 10 + 1, 11), (44100, 87)):
        x = np.zeros(n, np.float32); x[::7] = 0.1
        assert otiny.log_mel(x).shape == (160, exp), n
    with pytest.raises(RuntimeError):
        otiny.log_mel(np.zeros(100, np.float32))  # shorter than the reflect padding: the reference slices out of range


def test_oracle_encoder_vs_independent_restatement(otiny):
    codes = otiny.encode_mel(G["mel"])
    assert codes.dtype == np.uint32 and np.array_equal(codes, G["codes"])
    for i in range(7):
        _, st = otiny.encode_mel(G["mel"], stage=i, stage_size=G[f"stage{i}"].size)
        np.testing.assert_allclose(st.reshape(G[f"stage{i}"].shape), G[f"stage{i}"], atol=2e-5, rtol=0)
    assert codes.shape == (8, G["mel"].shape[1] // 4) and codes.max() < 1000
    # end to end from PCM (own FFT + regenerated table): same indices on this fixture
    assert np.array_equal(otiny.encode(G["pcm"]), G["codes"])


def _clip(n, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 44100.0
    return (0.25 * np.sin(2 * np.pi * 330.0 * t) + 0.1 * np.sin(2 * np.pi * 2500.0 * t + 0.5) + 0.05 * rng.randn(n)).astype(np.float32)


def _check_codes(got, o, pcm):
    exp = o.encode(pcm)
    mel = o.log_mel(pcm)
    _, margin = o.encode_mel(mel, stage=7, stage_size=exp.size)
    margin = margin.reshape(exp.shape)
    assert got.shape == (1,) + exp.shape
    bad = got[0] != exp
    assert not (bad & (margin > 1e-3)).any(), f"{int(bad.sum())} indices differ, worst margin {float(margin[bad].max()):.2e}"
    return int(bad.sum()), exp.size


@pytest.mark.gpu
def test_gpu_encode_tiny_vs_oracle(otiny):
    import fishrt
    c = fishrt.FireflyCodec(0, channel_div=8).load_synthetic(SEED)
    got = c.encode(G["pcm"][None, None])
    assert got.dtype == np.uint32 and np.array_equal(got[0], G["codes"])
    for n, seed in ((768 * 3, 1), (512 * 37 + 11, 2), (44100, 3)):
        nbad, tot = _check_codes(c.encode(_clip(n, seed)[None, None]), otiny, _clip(n, seed))
        assert nbad <= max(1, tot // 200)
    assert np.array_equal(c.encode(G["pcm"][None, None]), got)  # deterministic
    with pytest.raises(RuntimeError):
        c.encode(np.zeros((1, 1, 100), np.float32))
    with pytest.raises(ValueError):
        c.encode(np.zeros((1, 1, 5000), np.float32)[:, :, ::2])  # not contiguous
    # (b, 1, n) -> (b, 8, L) (firefly.rs:36-39, codec.rs:73-92): every clip on its own -- the reference's front-end would glue the clips
    # together (spectrogram.rs:33); batch rows equal the one-clip calls, ragged clips via `lengths`
    clips = [_clip(512 * 37 + 11, 20 + i) for i in range(3)]
    batch = np.stack(clips)[:, None, :]
    gb = c.encode(np.ascontiguousarray(batch))
    assert gb.shape[0] == 3 and gb.shape[1] == 8
    for i, cl in enumerate(clips):
        one = c.encode(cl[None, None])
        assert np.array_equal(gb[i], one[0]), i
        nbad, tot = _check_codes(gb[i:i + 1], otiny, cl)
        assert nbad <= max(1, tot // 200)
    lens = [512 * 37 + 11, 768 * 3, 9000]
    rag = c.encode(np.ascontiguousarray(batch), lengths=lens)
    for i, n in enumerate(lens):
        assert np.array_equal(rag[i], c.encode(np.ascontiguousarray(clips[i][None, None, :n]))[0]), i
    with pytest.raises(ValueError, match="lengths"):
        c.encode(np.ascontiguousarray(batch), lengths=[1, 2])
    # decode accepts what encode produces (same handle)
    pcm = c.decode(np.ascontiguousarray(got))
    assert pcm.shape == (1, 1, 2048 * got.shape[2]) and np.isfinite(pcm).all()


@pytest.mark.gpu
def test_gpu_encode_fullsize_vs_oracle():
    import fishrt
    o = orc.OracleCodec(tiny=False)
    o.load_synthetic(SEED)
    pcm = _clip(44100 * 2, 5)
    c = fishrt.FireflyCodec(0).load_synthetic(SEED)
    nbad, tot = _check_codes(c.encode(pcm[None, None]), o, pcm)
    print(f"full-size encode: {tot - nbad}/{tot} indices identical to the oracle ({nbad} at rounding boundaries)")
    assert nbad <= max(1, tot // 100)
