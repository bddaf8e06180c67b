"""Stateful streaming decode of the vocoder (fs_codec_stream_*): the chunks of one code sequence, decoded one after the other with the
convolutions' left context carried on the device, give the SAME PCM as decoding the whole sequence at once -- bit for bit, in both
matrix-core precision modes, for ragged chunk sizes -- and no frame is decoded twice (no halo)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("precision", ["f16", "bf16x3"])
def test_streamed_chunks_equal_one_shot_bit_for_bit(precision):
    voice = np.ascontiguousarray(np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32))  # (8, 274)
    c = fishrt.FireflyCodec(0, precision=precision).load_synthetic(0xC0DEC)
    ref = c.decode(voice[None])[0, 0]
    for sizes in ([16] * 17 + [2], [64, 17, 100, 93], [274], [33, 16, 225]):
        assert sum(sizes) == 274
        c.stream_begin()
        parts, a = [], 0
        for n in sizes:
            if n >= c.STREAM_MIN_FRAMES:
                parts.append(c.stream_decode(voice[:, a:a + n]))
            else:  # a stream's short tail: stateless decode with a halo (fishrt.stream.decode_chunk), the state is not needed afterwards
                parts.append(fishrt.decode_chunk(c, voice, a, a + n))
            a += n
        c.stream_end()
        pcm = np.concatenate(parts)
        assert pcm.shape == ref.shape and np.array_equal(pcm, ref), (precision, sizes, float(np.abs(pcm - ref).max()))
    # two streams one after the other on the same handle start from a clean state
    c.stream_begin()
    again = np.concatenate([c.stream_decode(voice[:, :137]), c.stream_decode(voice[:, 137:])])
    c.stream_end()
    assert np.array_equal(again, ref)
    c.close()


def test_stream_errors():
    c = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
    codes = np.zeros((8, 32), np.uint32)
    with pytest.raises(RuntimeError, match="stream_begin"):
        c.stream_decode(codes)
    c.stream_begin()
    with pytest.raises(RuntimeError, match="already open"):  # a second begin would zero the running stream's left context
        c.stream_begin()
    with pytest.raises(RuntimeError, match="16 frames"):
        c.stream_decode(codes[:, :8])
    c.stream_decode(codes)
    c.stream_end()
    c.close()
    c32 = fishrt.FireflyCodec(0, precision="f32").load_synthetic(0xC0DEC)
    with pytest.raises(RuntimeError, match="plane data flow"):
        c32.stream_begin()
    c32.close()
    tiny = fishrt.FireflyCodec(0, channel_div=8).load_synthetic(1)
    with pytest.raises(RuntimeError, match="plane data flow"):
        tiny.stream_begin()
    tiny.close()


def test_fused_vocoder_kernels_are_bit_identical_to_the_separate_ones(monkeypatch):
    """f16 mode: the thin stages' ResBlock pairs run as ONE kernel (intermediate in LDS, halo recomputed) and the ParallelBlock mean sits in the
    last residual conv's epilogue.  Both are the same f32 / f16 operations in the same order as the separate kernels, so the PCM must be
    identical bit for bit -- one-shot at lengths that are not multiples of the 128-sample tile, and streamed in uneven chunks (the
    intermediate's streaming context is carried although the intermediate itself is never written)."""
    codes = np.random.RandomState(3).randint(0, 1000, (1, 8, 83)).astype(np.uint32)
    pcm = {}
    for name, env in (("fused", {}), ("no_pair", {"FISHRT_VOC_NO_PAIR_FUSION": "1"}), ("no_mean", {"FISHRT_VOC_NO_FOLD_MEAN": "1"}),
                      ("neither", {"FISHRT_VOC_NO_PAIR_FUSION": "1", "FISHRT_VOC_NO_FOLD_MEAN": "1"})):
        for k in ("FISHRT_VOC_NO_PAIR_FUSION", "FISHRT_VOC_NO_FOLD_MEAN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = fishrt.FireflyCodec(0, precision="f16").load_synthetic(0xC0DEC)  # (the mean switch is read when the handle is created)
        one = c.decode(codes)
        parts = []
        c.stream_begin()
        for a, b in ((0, 17), (17, 40), (40, 83)):
            parts.append(c.stream_decode(np.ascontiguousarray(codes[0, :, a:b])))
        c.stream_end()
        assert np.array_equal(np.concatenate(parts), one[0, 0]), name
        for T in (16, 31):
            pcm[(name, T)] = c.decode(np.ascontiguousarray(codes[:, :, :T]))
        pcm[(name, 83)] = one
        c.close()
    for T in (16, 31, 83):
        for name in ("no_pair", "no_mean", "neither"):
            assert np.array_equal(pcm[("fused", T)], pcm[(name, T)]), (name, T)
