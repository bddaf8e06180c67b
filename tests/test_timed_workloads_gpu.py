"""GPU: the workloads bench.py times, checked for CORRECTNESS at full size (bench.py itself only checks determinism):
 (a) BASELINE.json configs[1] in full -- the 367-position default-voice prompt, 256 frames, the two persistent launches per frame: the
     logits of every one of the 256 x 9 decisions (fs_lm_debug_capture) against the CPU oracle teacher-forced on the GPU's own tokens,
     bf16 protocol (bf16-rounded weights, bf16 K/V), and every recorded pick == the rule applied to the recorded logits;
 (b) configs[2] shapes: Fish-1.5 generate_static_batch with 32 rows, greedy, against the oracle's static-batch restatement -- a row may
     leave the oracle's stream only on a near-tie the oracle reports;
 (c) the full-size vocoder on 64 frames against the oracle, both precision modes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bench
import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc
from test_lm_gpu import _rows_leave_oracle_only_at_near_ties
from test_persist_gpu import _RepPen

SEED = 0xF15E5EED
TOK = fcfg.FISH_1_5_TOKENS
IM_END = TOK["im_end_id"]
N_AUDIO = fcfg.FISH_1_5["vocab_size"] - IM_END
BF16_TOL = 1e-2  # DESIGN.md parity protocol: bf16 K/V rounding-boundary flips feed back through 24 layers (measured ~7e-3 at logit scale 3)


def _argmax_last(v):
    return int(np.nonzero(v == v.max())[0][-1])


def test_config1_full_workload_every_decision_vs_teacher_forced_oracle():
    F, rp = 256, 1.2
    p = bench.default_voice_prompt(TOK)
    L = p.shape[1]
    assert L == 367
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16").load_synthetic(SEED)
    lm.debug_capture(F)
    codes = lm.generate_blocking(p, F + L - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=rp, ignore_eos=True)
    assert codes.shape == (8, F) and lm.last_stats()["kernels_per_frame"] == 2
    cap = lm.debug_read(F)
    lm.close()
    slow_tok = cap[:, 0, 2047].astype(np.int64) + IM_END
    assert np.array_equal(cap[:, 1:, 1024].astype(np.int64).T, codes.astype(np.int64))
    # the recorded picks are the greedy rule (LAST maximal index) on the recorded logits
    for f in range(F):
        assert _argmax_last(cap[f, 0, :N_AUDIO]) + IM_END == slow_tok[f]
        for c in range(8):
            assert _argmax_last(cap[f, 1 + c, :1024]) == codes[c, f], (f, c)
    # teacher-forced oracle on the GPU's tokens (bf16 protocol)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    rps = [_RepPen(1024, rp) for _ in range(8)]
    femb = o.fast_embeddings()
    cur, pos, prev = p, 0, None
    worst_slow = worst_fast = 0.0
    for f in range(F):
        lg, hd = o.forward_generate(cur, pos, full_head=False)
        s = lg[0, IM_END:].copy()
        s[0] = -np.inf  # ignore_eos
        d = np.abs(s[1:] - cap[f, 0, 1:N_AUDIO]).max()
        worst_slow = max(worst_slow, float(d))
        assert d < BF16_TOL, ("slow logits", f, d)
        o.clear_fast()
        x = hd[0]
        for c in range(8):
            fg = o.forward_generate_fast(x, c)[0]
            if prev is not None:
                fg = rps[c].apply(fg, int(prev[c + 1]))
            d = np.abs(fg - cap[f, 1 + c, :1024]).max()
            worst_fast = max(worst_fast, float(d))
            assert d < BF16_TOL, ("fast logits", f, c, d)
            x = femb[int(codes[c, f])]
        frame = np.array([slow_tok[f]] + [int(v) for v in codes[:, f]], np.uint32)
        pos += cur.shape[1]
        prev, cur = frame, frame.reshape(9, 1)
    print(f"configs[1] (367-position prompt, {F} frames, KV to {pos}): max |dlogit| vs the teacher-forced oracle: slow {worst_slow:.2e}, fast {worst_fast:.2e} "
          f"over {F * 9} decisions (tolerance {BF16_TOL:.0e})")


def test_fish15_static_batch_32_rows_vs_oracle():
    B, frames = 32, 8
    rng = np.random.RandomState(77)
    lens = [int(v) for v in rng.randint(8, 25, B)]
    prompts = []
    for i, Ln in enumerate(lens):
        q = np.zeros((9, Ln), np.uint32)
        q[0] = np.random.RandomState(500 + i).randint(0, IM_END, Ln)
        prompts.append(q)
    M = max(lens) + frames - 2
    kw = dict(seed=42, temp=0.0, top_p=1.0, top_k=0, ignore_eos=True)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16", max_batch=B).load_synthetic(SEED)
    got = lm.generate_static_batch(prompts, M, **kw)
    lm.close()
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    exp = o.generate_batch(prompts, M, **kw)
    assert [g.shape for g in got] == [e.shape for e in exp] == [(8, frames)] * B
    flips = _rows_leave_oracle_only_at_near_ties(got, exp, o, "Fish-1.5 B=32")
    print(f"Fish-1.5 static batch, {B} rows x {frames} frames: {B - flips} rows identical to the oracle, {flips} left it on a near-tie the oracle reports")


def test_full_size_vocoder_64_frames_vs_oracle():
    G = os.path.join(os.path.dirname(__file__), "golden")
    voice = np.ascontiguousarray(np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32)[:, 100:164])  # 64 frames
    ref = orc.OracleCodec(tiny=False).load_synthetic(0xC0DEC).decode(voice)
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    for precision, tol in (("f16", 4e-5), ("bf16x3", 2.5e-5), ("f32", 1e-5)):
        c = fishrt.FireflyCodec(0, precision=precision).load_synthetic(0xC0DEC)
        pcm = c.decode(voice[None])[0, 0]
        c.close()
        r = float(np.sqrt(np.mean((pcm.astype(np.float64) - ref) ** 2)))
        print(f"full-size vocoder, 64 frames [{precision}]: PCM rms diff {r:.2e} at signal rms {sig:.3f}")
        assert pcm.shape == ref.shape == (2048 * 64,) and r < tol and sig > 1e-3


def _fp8_replay(o, cfg_name, cfg, tok, p, cap, codes, F, rp, greedy=True):
    """the fp8-mode oracle teacher-forced on one request's tokens: max |dlogit| of the slow / fast decisions against the capture record"""
    o.clear_slow()
    rps = [_RepPen(1024, rp) for _ in range(8)]
    femb = o.fast_embeddings()
    im_end, n_audio = tok["im_end_id"], cfg["vocab_size"] - tok["im_end_id"]
    cur, pos, prev, worst = p, 0, None, [0.0, 0.0]
    for f in range(F):
        lg, hd = o.forward_generate(cur, pos, full_head=(cfg_name == "fish14"))
        if cfg_name == "fish15":
            s = lg[0, im_end:].copy()
            d = float(np.abs(s[1:] - cap[f, 0, 1:n_audio]).max())
            worst[0] = max(worst[0], d)
            assert d < BF16_TOL, ("slow logits", f, d)
            slow = int(cap[f, 0, 2047]) + im_end
            assert not greedy or slow == _argmax_last(np.concatenate([[-np.inf], cap[f, 0, 1:n_audio]])) + im_end
        else:
            slow = tok["pad_id"]  # ignore_eos: the legacy 2-way draw always yields the <|semantic|> / pad token (single_batch.rs:104-124)
            two = np.array([lg[0, tok["pad_id"]], lg[0, im_end]])  # the in-launch decision saw {pad, im_end} logits, drew u, picked pad
            d = float(np.abs(two - cap[f, 0, :2]).max())
            worst[0] = max(worst[0], d)
            assert d < BF16_TOL and 0.0 <= cap[f, 0, 2] < 1.0 and cap[f, 0, 2047] == 0.0, ("legacy slow decision", f, d)
        o.clear_fast()
        x = hd[0]
        for c in range(8):
            fg = o.forward_generate_fast(x, c)[0]
            if prev is not None:
                fg = rps[c].apply(fg, int(prev[c + 1]))
            d = float(np.abs(fg - cap[f, 1 + c, :1024]).max())  # (both token layouts: the capture hook lives in the folded prologue path)
            worst[1] = max(worst[1], d)
            assert d < BF16_TOL, ("fast logits", f, c, d)
            assert int(cap[f, 1 + c, 1024]) == codes[c, f] and (not greedy or _argmax_last(cap[f, 1 + c, :1024]) == codes[c, f]), (f, c)
            x = femb[int(codes[c, f])]
        frame = np.array([slow] + [int(v) for v in codes[:, f]], np.uint32)
        pos += cur.shape[1]
        prev, cur = frame, frame.reshape(9, 1)
    return worst


@pytest.mark.parametrize("cfg_name", ["fish15", "fish14"])
def test_fp8_persistent_path_every_decision_vs_fp8_oracle(cfg_name):
    """FS_FP8 handles take the persistent kernels too (e4m3 weight images for the slow kernel, bf16-widened e4m3 + row scales for the
    resident fast decoder): 48 frames on a 200-position prompt, the logits of all 48 x 9 decisions against the fp8-mode oracle (same
    per-row e4m3 quantiser, computes on the dequantised values) teacher-forced on the GPU's tokens.  fish14 = BASELINE configs[4] shapes
    (32k vocabulary, legacy 2-way slow token: its decision stays in k_sample_slow, the codebook decisions are in-launch)."""
    F, rp = 48, 1.2
    cfg, tok = (fcfg.FISH_1_5, TOK) if cfg_name == "fish15" else (fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS)
    ocfg = dict(cfg, **tok)
    rs = np.random.RandomState(3)
    p = np.zeros((9, 200), np.uint32)
    p[0] = rs.randint(6, min(tok["im_end_id"] if cfg_name == "fish15" else cfg["vocab_size"], cfg["vocab_size"]), 200)
    lm = fishrt.DualARTransformer(cfg, tok, 0, "fp8").load_synthetic(SEED)
    lm.debug_capture(F)
    codes = lm.generate_blocking(p, F + 200 - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=rp, ignore_eos=True)
    kpf = lm.last_stats()["kernels_per_frame"]
    assert codes.shape == (8, F) and kpf == 2, kpf  # (round 4: the Fish <= 1.4 2-way slow draw is taken inside k_fast_persist too)
    cap = lm.debug_read(F)
    lm.close()
    o = orc.OracleLM(ocfg).load_synthetic(SEED, fp8=True)
    o.set_kv_round_bf16(True)
    worst = _fp8_replay(o, cfg_name, cfg, tok, p, cap, codes, F, rp)
    print(f"fp8 persistent path [{cfg_name}], {F} frames: max |dlogit| vs the fp8-mode oracle: slow {worst[0]:.2e}, fast {worst[1]:.2e}")


@pytest.mark.parametrize("cfg_name,sampled", [("fish14", False), ("fish15", False), ("fish14", True)], ids=["fish14-greedy", "fish15-greedy", "fish14-sampled"])
def test_fp8_request_rows_every_decision_vs_fp8_oracle(cfg_name, sampled):
    """round 5: FS_FP8 and Fish <= 1.4 handles take the request-row kernels too (fs_lm_generate_multi: the e4m3 weights widened to bf16 in the
    MFMA images, the quantiser's row scales in the publishing lanes; the legacy 2-way slow draw per row inside k_fast_rows).  4 concurrent
    BASELINE configs[4]-style requests: one slow + one fast launch per frame, every decision of every row against the fp8-mode oracle
    teacher-forced on the row's own tokens (sampled rows: the logits every draw saw; the draws themselves are replayed through the oracle
    sampler in tests/test_rows_gpu.py on the bf16 handle -- the sampler code is the same)."""
    F, rp, n = 32, 1.2, 4
    cfg, tok = (fcfg.FISH_1_5, TOK) if cfg_name == "fish15" else (fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS)
    lens = [60, 141, 33, 97]
    prompts = []
    for i, L in enumerate(lens):
        q = np.zeros((9, L), np.uint32)
        q[0] = np.random.RandomState(40 + i).randint(6, min(tok["im_end_id"] if cfg_name == "fish15" else cfg["vocab_size"], cfg["vocab_size"]), L)
        prompts.append(q)
    kw = dict(temp=0.7, top_p=0.8, top_k=256) if sampled else dict(temp=0.0, top_p=1.0, top_k=0)
    lm = fishrt.DualARTransformer(cfg, tok, 0, "fp8", max_batch=4).load_synthetic(SEED)
    assert lm.rows_supported(n, **kw)
    lm.debug_capture(F)
    got = lm.generate_multi(prompts, [L + F - 2 for L in lens], repetition_penalty=rp, seeds=[70 + i for i in range(n)], ignore_eos=True, **kw)
    assert lm.last_stats()["kernels_per_frame"] == 2, "the request-row kernels were not taken"
    caps = [lm.debug_read_row(i, F) for i in range(n)]
    lm.close()
    o = orc.OracleLM(dict(cfg, **tok)).load_synthetic(SEED, fp8=True)
    o.set_kv_round_bf16(True)
    for i in range(n):
        assert got[i].shape == (8, F)
        worst = _fp8_replay(o, cfg_name, cfg, tok, prompts[i], caps[i], got[i], F, rp, greedy=not sampled)
        print(f"fp8 request rows [{cfg_name}{', sampled' if sampled else ''}] row {i} (L {lens[i]}): max |dlogit| vs the fp8-mode oracle: slow {worst[0]:.2e}, fast {worst[1]:.2e}")
