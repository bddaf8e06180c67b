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
