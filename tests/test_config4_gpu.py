"""GPU: BASELINE.json configs[4] -- Fish-1.4 shapes (V = 32000, single <|semantic|> id, legacy 2-way slow sampler), fp8-e4m3 weights,
one long stream with the Firefly vocoder overlapped on a second HIP stream.
 (1) teacher-forced steps at KV > 4096 against the fp8-mode oracle (same e4m3 bytes + row scales, K/V rounded to bf16);
 (2) a 4096-frame sampled run: StreamingSynth's PCM (chunks vocoded while the LM keeps generating) is bit-identical to one-shot
     decoding of the same codes, the codes equal those of plain generate_blocking, and the first audio arrives early."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED
FISH14_ORC = dict(orc.FISH15, vocab_size=32000, im_end_id=4, pad_id=5, semantic_start_id=5, semantic_end_id=0, has_semantic_end=0)
FP8_TOL = 2e-2  # tests/test_fp8_gpu.py: fp8 oracle vs device at Fish-1.5 shapes, measured < 1e-2


def _prompt14(L, seed):
    rng = np.random.RandomState(seed)
    p = np.zeros((9, L), np.uint32)
    p[0] = rng.randint(6, 32000, L)
    return p


@pytest.fixture(scope="module")
def lm14():
    lm = fishrt.DualARTransformer(fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS, 0, "fp8").load_synthetic(SEED)
    yield lm
    lm.close()


def test_fish14_fp8_teacher_forced_beyond_4096_tokens(lm14):
    """4100 prompt tokens (three MFMA prefill passes of <= 2048 rows on the device), then three single-token steps at KV 4100..4102
    (attention graph bucket of 64 chunks): logits and hidden state of every call against the fp8-mode oracle"""
    o = orc.OracleLM(FISH14_ORC).load_synthetic(SEED, fp8=True)
    o.set_kv_round_bf16(True)
    L0 = 4100
    p = _prompt14(L0 + 3, 21)
    lm14.clear_slow_layer_caches()
    worst = 0.0
    for lo_, hi_ in ((0, L0), (L0, L0 + 1), (L0 + 1, L0 + 2), (L0 + 2, L0 + 3)):
        chunk = np.ascontiguousarray(p[:, lo_:hi_])
        lg, hg = lm14.forward_generate(chunk, lo_)
        lo, ho = o.forward_generate(chunk, lo_)
        dl = float(np.abs(lg[0] - lo[0]).max())
        dh = float(np.abs(hg - ho).max() / np.sqrt(np.mean(ho ** 2)))
        worst = max(worst, dl, dh)
        assert dl < FP8_TOL and dh < FP8_TOL, (lo_, hi_, dl, dh)
    assert lm14.curr_kv_size() == L0 + 3
    print(f"Fish-1.4 fp8, KV {L0}..{L0 + 3}: worst max|dlogit| / |dh|/rms vs the fp8 oracle {worst:.2e}")


class _Clamp:  # random synthetic weights emit codebook entries up to 1023; the FSQ codebook has 1000
    def __init__(self, c):
        self.c = c

    def decode(self, codes):
        return self.c.decode(np.minimum(codes, 999))

    STREAM_MIN_FRAMES = 16

    def stream_begin(self):
        self.c.stream_begin()

    def stream_decode(self, codes):
        return self.c.stream_decode(np.minimum(codes, 999))

    def stream_end(self):
        self.c.stream_end()


def test_fish14_fp8_4096_frames_streamed_pcm_equals_one_shot(lm14):
    codec = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
    p = _prompt14(64, 4)
    frames = 4096
    M = frames + 64 - 2
    kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=1, ignore_eos=True)
    lm14.clear_slow_layer_caches()
    codes = lm14.generate_blocking(p, M, **kw)
    assert codes.shape == (8, frames)
    pcm = _Clamp(codec).decode(np.ascontiguousarray(codes[None]))[0, 0]
    lm14.clear_slow_layer_caches()
    synth = fishrt.StreamingSynth(lm14, _Clamp(codec), chunk=256, first_chunk=32)
    c2, pcm2 = synth(p, M, **kw)
    st = synth.stats
    assert np.array_equal(c2, codes), "the frame callback changed the token stream"
    assert pcm2.shape == pcm.shape == (2048 * frames,) and np.array_equal(pcm2, pcm), "streamed PCM differs from one-shot decoding"
    assert np.isfinite(pcm).all() and float(np.abs(pcm).max()) <= 1.0
    print(f"4096 frames: LM {st['lm_s']:.3f}s, vocoder busy {st['vocoder_busy_s']:.3f}s, total {st['total_s']:.3f}s, first audio after "
          f"{st['first_audio_s'] * 1e3:.0f} ms, overlap efficiency {st['overlap_efficiency']:.2f}")
    # (a latency property, not a parity one: normally 40-50 ms of 2.9 s; the bound is loose so that a momentarily busy box cannot fail it)
    assert st["first_audio_s"] < 0.5 * st["total_s"], st
    codec.close()


@pytest.mark.parametrize("inline", [False, True])
def test_stateful_stream_with_chunks_below_the_codec_minimum_stays_bit_identical(lm14, inline):
    """first_chunk = 8 / chunk = 12 on a stateful stream used to send non-final chunks through the stateless halo path, after
    which the next stateful chunk started from a stale left context.  The sizes are raised to the codec's minimum now; PCM == one-shot."""
    p = _prompt14(24, 21)
    frames = 77
    M = frames + p.shape[1] - 2
    kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=9, ignore_eos=True)
    codec = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
    lm14.clear_slow_layer_caches()
    codes = lm14.generate_blocking(p, M, **kw)
    pcm = _Clamp(codec).decode(np.ascontiguousarray(codes[None]))[0, 0]
    lm14.clear_slow_layer_caches()
    synth = fishrt.StreamingSynth(lm14, _Clamp(codec), chunk=12, first_chunk=8, inline=inline)
    c2, pcm2 = synth(p, M, **kw)
    assert synth.stats["stateful"] and np.array_equal(c2, codes)
    assert pcm2.shape == pcm.shape and np.array_equal(pcm2, pcm), float(np.abs(pcm2 - pcm).max())
    codec.close()


def test_streaming_synth_inline_mode_equals_threaded(lm14):
    """inline=True vocodes each chunk inside the frame callback (the LM pauses) instead of in the worker thread: same codes, same PCM,
    ragged tail included (frames not a multiple of the chunk), errors from the vocoder still reach the caller"""
    p = _prompt14(24, 9)
    frames = 300
    M = frames + p.shape[1] - 2
    kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=5, ignore_eos=True)
    codec = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
    outs = []
    for inline in (False, True):
        lm14.clear_slow_layer_caches()
        synth = fishrt.StreamingSynth(lm14, _Clamp(codec), chunk=64, first_chunk=16, inline=inline)
        outs.append(synth(p, M, **kw))
        assert synth.stats["frames"] == frames and synth.stats["first_audio_s"] is not None
    (c0, p0), (c1, p1) = outs
    assert c0.shape == (8, frames) and np.array_equal(c0, c1)
    assert p0.shape == (2048 * frames,) and np.array_equal(p0, p1)
    one_shot = _Clamp(codec).decode(np.ascontiguousarray(c0[None]))[0, 0]
    assert np.array_equal(p1, one_shot)

    class Boom:
        def decode(self, codes):
            raise ValueError("vocoder exploded")
    lm14.clear_slow_layer_caches()
    with pytest.raises(ValueError, match="vocoder exploded"):
        fishrt.StreamingSynth(lm14, Boom(), chunk=16, first_chunk=8, inline=True)(p, 24 + 40, temp=0.0, top_p=1.0, top_k=0, ignore_eos=True)
    codec.close()


def test_streaming_synth_surfaces_errors_and_never_hangs(lm14):
    """the worker always gets its sentinel; an exception in the vocoder thread or in generate_blocking reaches the caller"""
    class Boom:
        def decode(self, codes):
            raise ValueError("vocoder exploded")
    p = _prompt14(8, 5)
    lm14.clear_slow_layer_caches()
    with pytest.raises(ValueError, match="vocoder exploded"):
        fishrt.StreamingSynth(lm14, Boom(), chunk=16, first_chunk=8)(p, 8 + 40, temp=0.0, top_p=1.0, top_k=0, ignore_eos=True)
    codec = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
    with pytest.raises(RuntimeError):  # prompt longer than max_seq_len: generate_blocking raises, the worker thread is joined
        fishrt.StreamingSynth(lm14, _Clamp(codec), chunk=16)(_prompt14(9000, 6), 9100, temp=0.0, top_p=1.0, top_k=0)
    codec.close()
