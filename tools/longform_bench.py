"""BASELINE.json configs[4]: Fish-1.4 shapes (V = 32000, single <|semantic|> id, legacy 2-way slow sampler),
one stream of 4096 frames (KV grows to 4096 + L; RoPE table 8192 -- the reference would fail past its max_seq_len 4096,
dual_ar.rs:179,623), Firefly vocoder on a second stream consuming a 32-frame first chunk, then 256-frame chunks (+24-frame halo) while the
LM continues.
Reports LM-only RTF, end-to-end RTF and overlap efficiency.  usage: longform_bench.py [frames] [fp8|bf16] (configs[4] names
fp8-e4m3 weights; bf16 is the comparison run)."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fish-speech.rs_amd")
import numpy as np, fishrt
from fishrt import config as fcfg
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
vprec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
inline = len(sys.argv) > 4 and sys.argv[4] == "inline"
lm = fishrt.DualARTransformer(fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
codec = fishrt.FireflyCodec(0, precision=vprec).load_synthetic(0xC0DEC)
rng = np.random.RandomState(4)
L = 64
p = np.zeros((9, L), np.uint32); p[0] = rng.randint(6, 32000, L)
kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=1, ignore_eos=True)
M = frames + L - 2
# fast codes may exceed the FSQ range (codebook 1024 vs 1000 FSQ entries) with random weights: clamp for the vocoder
class Clamp:
    def __init__(s, c): s.c = c
    def decode(s, codes): return s.c.decode(np.minimum(codes, 999))
    STREAM_MIN_FRAMES = 16
    def stream_begin(s): s.c.stream_begin()
    def stream_decode(s, codes): return s.c.stream_decode(np.minimum(codes, 999))
    def stream_end(s): s.c.stream_end()
for rep in range(2):
    lm.clear_slow_layer_caches()
    t = time.perf_counter(); codes = lm.generate_blocking(p, M, **kw); t_lm = time.perf_counter() - t
    t = time.perf_counter(); pcm = Clamp(codec).decode(np.ascontiguousarray(codes[None])); t_voc = time.perf_counter() - t
    lm.clear_slow_layer_caches()
    synth = fishrt.StreamingSynth(lm, Clamp(codec), chunk=256, first_chunk=32, inline=inline)
    c2, pcm2 = synth(p, M, **kw)
    st = synth.stats
    audio_s = frames / 21.535
    print(f"[{dtype}, vocoder {vprec}{', inline' if inline else ''}] frames={codes.shape[1]}: LM alone {t_lm:.3f}s (RTF {audio_s/t_lm:.1f}), vocoder alone (one shot) {t_voc:.3f}s, sequential {t_lm+t_voc:.3f}s | "
          f"overlapped total {st['total_s']:.3f}s (RTF {audio_s/st['total_s']:.1f}), vocoder busy {st['vocoder_busy_s']:.3f}s, "
          f"first audio after {st['first_audio_s'] * 1e3:.0f} ms, overlap efficiency {st['overlap_efficiency']:.2f}, same codes {np.array_equal(codes, c2)}, pcm identical {np.array_equal(pcm[0,0], pcm2)}")

# round 5: N concurrent configs[4]-style requests on the request-row kernels (fs_lm_generate_multi on the fp8 Fish-1.4 handle) against the same
# requests one after the other (what the fallback did before the fp8 / legacy row path existed, and what the reference's mutex does)
if len(sys.argv) > 5:
    n = int(sys.argv[5])
    F2 = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    lm.close()
    lmr = fishrt.DualARTransformer(fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS, 0, dtype, max_batch=n).load_synthetic(0xF15E5EED)
    ps = []
    for i in range(n):
        q = np.zeros((9, L), np.uint32); q[0] = np.random.RandomState(10 + i).randint(6, 32000, L); ps.append(q)
    kw2 = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, ignore_eos=True)
    M2 = F2 + L - 2
    t_seq = 0.0
    for rep in range(2):
        t = time.perf_counter()
        for i in range(n):
            lmr.clear_slow_layer_caches()
            lmr.generate_blocking(ps[i], M2, seed=10 + i, **kw2)
        t_seq = time.perf_counter() - t
        t = time.perf_counter(); outs = lmr.generate_multi(ps, M2, seeds=[10 + i for i in range(n)], **kw2); t_rows = time.perf_counter() - t
        st = lmr.last_stats()
    assert all(o.shape == (8, F2) for o in outs) and st["kernels_per_frame"] == 1 + (n + 3) // 4, st
    print(f"[{dtype}] {n} concurrent Fish-1.4 requests x {F2} frames (sampled top-k 256 / top-p 0.8, KV to {L + F2}): request rows {t_rows:.3f}s = {n * F2 / t_rows:.0f} frames/s "
          f"({st['decode_ms'] * 1e3 / (F2 - 1):.0f} us per {n}-row frame, {st['kernels_per_frame']} launches per frame) vs one at a time {t_seq:.3f}s = {n * F2 / t_seq:.0f} frames/s "
          f"-> {t_seq / t_rows:.2f} x; RTF of the group {n * F2 / 21.535 / t_rows:.1f}")
