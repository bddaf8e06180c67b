"""Stress of tests/test_config4_gpu.py::test_fish14_fp8_4096_frames_streamed_pcm_equals_one_shot: N repetitions in one process, reporting WHICH
comparison fails (token stream under the concurrent vocoder / streamed PCM vs one-shot PCM).  usage: stress_config4.py [N] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, fishrt
import test_config4_gpu as t4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lm14 = fishrt.DualARTransformer(t4.fcfg.FISH_1_4, t4.fcfg.FISH_1_4_TOKENS, 0, "fp8").load_synthetic(0xF15E5EED)
codec = fishrt.FireflyCodec(0).load_synthetic(0xC0DEC)
p = t4._prompt14(64, 4)
M = frames + 64 - 2
kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=1, ignore_eos=True)
lm14.clear_slow_layer_caches()
codes = lm14.generate_blocking(p, M, **kw)
pcm = t4._Clamp(codec).decode(np.ascontiguousarray(codes[None]))[0, 0]
bad_tok = bad_pcm = bad_one = bad_lm = 0
for it in range(N):
    lm14.clear_slow_layer_caches()
    c1 = lm14.generate_blocking(p, M, **kw)
    if not np.array_equal(c1, codes):
        bad_lm += 1; print(it, "LM alone differs at frame", int(np.argmax((c1 != codes).any(0))), flush=True)
    p1 = t4._Clamp(codec).decode(np.ascontiguousarray(codes[None]))[0, 0]
    if not np.array_equal(p1, pcm):
        bad_one += 1; print(it, "one-shot PCM differs: first sample", int(np.argmax(p1 != pcm)), "count", int((p1 != pcm).sum()), flush=True)
    lm14.clear_slow_layer_caches()
    synth = fishrt.StreamingSynth(lm14, t4._Clamp(codec), chunk=256, first_chunk=32)
    c2, pcm2 = synth(p, M, **kw)
    if not np.array_equal(c2, codes):
        bad_tok += 1; print(it, "streamed LM tokens differ at frame", int(np.argmax((c2 != codes).any(0))), flush=True)
    elif not np.array_equal(pcm2, pcm):
        d = np.nonzero(pcm2 != pcm)[0]
        bad_pcm += 1; print(it, "streamed PCM differs:", len(d), "samples, first", int(d[0]), "= frame", int(d[0]) // 2048, "last", int(d[-1]), "max", float(np.abs(pcm2 - pcm).max()), flush=True)
print(f"{N} repetitions: LM alone differs {bad_lm}, one-shot PCM differs {bad_one}, streamed tokens differ {bad_tok}, streamed PCM differs {bad_pcm}")
