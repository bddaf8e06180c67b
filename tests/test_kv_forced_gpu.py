"""GPU: the persistent decode kernels the bench times (k_slow_persist / k_fast_persist, and the request-row kernels) against the CPU oracle
at a tolerance that separates "right" from "nearly right" (What's weak 1).

The bf16 protocol of the other tests (BF16_TOL = 1e-2) is as loose as it is because bf16 K/V entries sit on rounding boundaries: a 1e-6
difference in the f32 value (summation order) flips an entry by one bf16 ulp (0.4 %), the flipped entry feeds every later step, and 24
layers compound it to ~7e-3 on logits of scale 3.  Here the oracle is teacher-forced on the GPU's tokens AND on the GPU's own cached K/V
(fs_lm_debug_read_kv -> OracleLM.set_kv): every cached row the oracle attends over is bit-for-bit the row the kernel attended over, so
what separates the two logit vectors is the summation order of the CURRENT step only (plus the current token's own K/V entries, 1 row of
T).  A summation-order bug of 1e-3 in any stage of the persistent slow kernel shows up as a failure of this test; it would pass the
1e-2 ones."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bench
import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc
from test_persist_gpu import _RepPen

SEED = 0xF15E5EED
TOK = fcfg.FISH_1_5_TOKENS
IM_END = TOK["im_end_id"]
N_AUDIO = fcfg.FISH_1_5["vocab_size"] - IM_END
SLOW_TOL = 2e-4   # slow logits, KV forced: only the current step's summation order differs (measured 4e-5 .. 6e-5; logit scale ~3)
GREEDY = dict(temp=0.0, top_p=1.0, top_k=0)
SAMPLED = dict(temp=0.7, top_p=0.8, top_k=256)
FAST_TOL = 2e-4   # fast logits: the fast decoder's per-frame K/V rows ride in the capture record (raw bf16 pairs) and are forced the same way


def _prompt(L, seed):
    q = np.zeros((9, L), np.uint32)
    q[0] = np.random.RandomState(seed).randint(0, IM_END, L)
    return q


_kv_units = []  # worst fast-decoder K/V distance of every replay (reported next to the logit differences)


def _pairs(u):
    """64 raw bf16 pairs (low half = even element) -> f32 (1, 2 kv heads, 64 dims)"""
    out = np.empty((64, 2), np.float32)
    out[:, 0] = (u << np.uint32(16)).view(np.float32)
    out[:, 1] = (u & np.uint32(0xFFFF0000)).view(np.float32)
    return out.reshape(1, 2, 64)


def _kv_forced_replay(o, lm, slot, p, cap, codes, rp, n_layer):
    """returns (max |dlogit| slow over frames >= 1, the same for frame 0, max |dlogit| fast)"""
    F, L = codes.shape[1], p.shape[1]
    T = L + F - 1
    gk = [lm.debug_read_kv(l, 0, T, slot=slot) for l in range(n_layer)]
    slow_tok = cap[:F, 0, 2047].astype(np.int64) + IM_END
    o.clear_slow()
    rps = [_RepPen(1024, rp) for _ in range(8)]
    femb = o.fast_embeddings()
    cur, pos, prev = p, 0, None
    w0 = ws = wf = 0.0
    kv_ulp = [0.0]
    slow_units = 0.0
    for f in range(F):
        if f > 0:
            # single-token steps: the step's OWN K/V row is the kernel's too (one-shot force; what the oracle computed for it is checked to one
            # rounding step below).  Without this the newest row -- 1 / T of the attention weight, i.e. most for the short prompts -- could sit
            # on the other side of a bf16 rounding boundary in some layer: measured up to 3e-4 on the logits of 40 .. 150-token prompts
            # (sampled runs), against <= 6e-5 with it
            for l in range(n_layer):
                o.force_kv(l, gk[l][0][pos], gk[l][1][pos])
        lg, hd = o.forward_generate(cur, pos, full_head=False)
        n = cur.shape[1]
        if f > 0:
            slow_units = max([slow_units] + [o.force_kv_diff(l) for l in range(n_layer)])
        for l in range(n_layer):  # from now on the oracle attends over the rows the kernel cached
            o.set_kv(l, pos, gk[l][0][pos:pos + n], gk[l][1][pos:pos + n])
        s = lg[0, IM_END:].copy()
        d = float(np.abs(s[1:] - cap[f, 0, 1:N_AUDIO]).max())
        if f == 0:
            w0 = d  # (the prefill pass ran on the oracle's own K/V rows: bf16 protocol)
        else:
            ws = max(ws, d)
        o.clear_fast()
        x = hd[0]
        for c in range(8):
            # the fast decoder attends over <= 8 rows, so the CURRENT pass's own row carries real weight: the oracle takes the kernel's row for
            # it too (one-shot force) -- after checking that its own row is the same to one bf16 ulp (a rounding-boundary flip at most)
            raw = cap[f, 1 + c, 1025:1025 + 512].view(np.uint32).reshape(4, 2, 64)  # [layer][K | V][64 bf16 pairs = (kv head, dim pair)]
            for l in range(4):
                o.force_kv(1000 + l, _pairs(raw[l, 0]), _pairs(raw[l, 1]))
            fg = o.forward_generate_fast(x, c)[0]
            if f > 0:  # (frame 0's hidden state comes out of the oracle's own-K/V prefill: bf16 protocol, not comparable at one ulp)
                kv_ulp[0] = max([kv_ulp[0]] + [o.force_kv_diff(1000 + l) for l in range(4)])
            if prev is not None:
                fg = rps[c].apply(fg, int(prev[c + 1]))
            d = float(np.abs(fg - cap[f, 1 + c, :1024]).max())
            if f == 0:
                w0 = max(w0, d)
            else:
                wf = max(wf, d)
            x = femb[int(codes[c, f])]
        frame = np.array([slow_tok[f]] + [int(v) for v in codes[:, f]], np.uint32)
        pos += n
        prev, cur = frame, frame.reshape(9, 1)
    # one bf16 ulp (a rounding-boundary flip), or -- for entries below ~1e-3, where the unit is floored at 2^-17 = 7.6e-6 -- what the hidden
    # state's own summation-order difference (up to 6e-5, the SLOW_TOL quantity) leaves on a projection: measured 1 .. 9 units over the 30
    # instantiation runs of this file (profiles/r05_rows_parity.txt); 16 units = 1.2e-4 is the same bound as SLOW_TOL / FAST_TOL
    assert kv_ulp[0] <= 16.0, f"a fast-decoder K/V entry of the kernel is {kv_ulp[0]:.2f} units (bf16 ulp, floored at 2^-17) from the oracle's"
    assert slow_units <= 16.0, f"a slow-layer K/V entry the kernel cached is {slow_units:.2f} units (bf16 ulp, floored at 2^-17) from the oracle's"
    _kv_units.append(max(kv_ulp[0], slow_units))
    return ws, w0, wf


def _report(line):
    """every measured max |dlogit| goes to stdout (pytest -s) and, when FISHRT_PARITY_LOG names a file, into it (profiles/r05_rows_parity.txt)"""
    print(line)
    path = os.environ.get("FISHRT_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def _b1_case(dtype, F, p, rp, sampling, kernels):
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, dtype).load_synthetic(SEED)
    lm.debug_capture(F)
    codes = lm.generate_blocking(p, F + p.shape[1] - 2, repetition_penalty=rp, ignore_eos=True, seed=4321, **sampling)
    assert codes.shape == (8, F) and lm.last_stats()["kernels_per_frame"] == 2
    cap = lm.debug_read(F)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=(dtype == "bf16"), fp8=(dtype == "fp8"))
    o.set_kv_round_bf16(True)
    ws, w0, wf = _kv_forced_replay(o, lm, 0, p, cap, codes, rp, fcfg.FISH_1_5["n_layer"])
    lm.close()
    _report(f"{kernels}: L {p.shape[1]}, {F} frames, {sampling}: max |dlogit| slow {ws:.2e} (frame 0, own K/V: {w0:.2e}), fast {wf:.2e}  "
            f"[tolerances {SLOW_TOL:.0e} / {FAST_TOL:.0e}; the unforced protocol allows 1e-2]; own K/V rows within {_kv_units[-1]:.1f} units")
    return ws, w0, wf


def test_persistent_kernels_kv_forced_oracle_configs1():
    """configs[1] prompt (367 positions), 128 frames on the two persistent launches per frame"""
    ws, w0, wf = _b1_case("bf16", 128, bench.default_voice_prompt(TOK), 1.2, GREEDY, "k_slow_persist<bf16> / k_fast_persist<greedy>")
    assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2


def test_sampled_fast_kernel_kv_forced_oracle():
    """the in-launch-sampler instantiation k_fast_persist<SAMPLED> (the server default: temp 0.7 / top-p 0.8 / top-k 256).  The replay is
    teacher-forced on the GPU's SAMPLED tokens; what is compared are the logits every draw saw (the draws themselves are replayed through the
    oracle sampler in test_persist_sampled_gpu.py)."""
    ws, w0, wf = _b1_case("bf16", 64, _prompt(150, 77), 1.2, SAMPLED, "k_slow_persist<bf16> / k_fast_persist<sampled>")
    assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2


@pytest.mark.parametrize("sampling", [GREEDY, SAMPLED], ids=["greedy", "sampled"])
def test_fp8_persistent_kernels_kv_forced_fp8_oracle(sampling):
    """the e4m3 images of both persistent kernels against the fp8-mode oracle (same integer quantiser, computes on the dequantised values):
    the kernels widen e4m3 -> bf16 exactly and apply the row scale to the K-summed accumulator, the oracle multiplies scale x code first --
    rounding-level difference, so the bf16 bound holds (measured 2e-5 .. 6e-5)"""
    ws, w0, wf = _b1_case("fp8", 64, _prompt(150, 78), 1.2, sampling, "k_slow_persist<fp8> / k_fast_persist<fp8>")
    assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2


# n requests -> k_slow_rows<R> + the fast launch groups: 2 -> <2> + <2>; 4 -> <4> + <4>; 5 -> <8> (two column tiles) + <4>, <1>;
# 6 -> <8> + <4>, <2>; 8 -> <8> + <4>, <4>
ROW_LENS = [40, 130, 77, 250, 61, 199, 33, 160]


@pytest.mark.parametrize("n,sampling,dtype", [(2, GREEDY, "bf16"), (4, GREEDY, "bf16"), (5, GREEDY, "bf16"), (6, GREEDY, "bf16"), (8, GREEDY, "bf16"),
                                              (4, SAMPLED, "bf16"), (5, SAMPLED, "bf16"), (4, GREEDY, "fp8"), (5, SAMPLED, "fp8")],
                         ids=["R2", "R4", "R5", "R6", "R8", "R4-sampled", "R5-sampled", "R4-fp8", "R5-fp8-sampled"])
def test_row_kernels_kv_forced_oracle(n, sampling, dtype):
    """every instantiation of the request-row kernels (k_slow_rows<2|4|8>, k_fast_rows<1|2|4, greedy|sampled>) against the ORACLE, each row
    teacher-forced on its own tokens and its own cached K/V"""
    F, rp = 40 if n <= 5 else 32, 1.2
    lens = ROW_LENS[:n]
    prompts = [_prompt(L, 900 + i) for i, L in enumerate(lens)]
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, dtype, max_batch=8 if n > 4 else n).load_synthetic(SEED)
    lm.debug_capture(F)
    got = lm.generate_multi(prompts, [L + F - 2 for L in lens], repetition_penalty=rp, ignore_eos=True, seeds=[100 + i for i in range(n)], **sampling)
    assert lm.last_stats()["kernels_per_frame"] == 1 + (n + 3) // 4
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=(dtype == "bf16"), fp8=(dtype == "fp8"))
    o.set_kv_round_bf16(True)
    R, worst = (2 if n <= 2 else 4 if n <= 4 else 8), [0.0, 0.0]
    for i in range(n):
        assert got[i].shape == (8, F)
        cap = lm.debug_read_row(i, F)
        ws, w0, wf = _kv_forced_replay(o, lm, i, prompts[i], cap, got[i], rp, fcfg.FISH_1_5["n_layer"])
        left = min(4, n - 4 * (i // 4))
        _report(f"k_slow_rows<{R}> / k_fast_rows<{4 if left >= 3 else left}, {'sampled' if sampling is SAMPLED else 'greedy'}>{' [fp8 handle]' if dtype == 'fp8' else ''}: n {n} row {i} (L {lens[i]}), "
                f"{F} frames: max |dlogit| slow {ws:.2e} (frame 0: {w0:.2e}), fast {wf:.2e}; own K/V rows within {_kv_units[-1]:.1f} units")
        assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2, (i, ws, w0, wf)
        worst = [max(worst[0], ws), max(worst[1], wf)]
    lm.close()


def _kv_forced_batch_row(o, lm, row, padded, cap, codes, n_layer):
    """static-batch row (left-padded prompt, no repetition penalty): max |dlogit| of the SLOW logits over frames >= 1 with the oracle attending
    over the rows the GPU cached for this row (the row path's captures carry no fast-decoder K/V, so the fast logits stay with the bf16
    protocol of tests/test_batch_capture_gpu.py), and for frame 0 (oracle's own prefill rows)"""
    F, L = codes.shape[1], padded.shape[1]
    gk = [lm.debug_read_kv(l, 0, L + F - 1, slot=row) for l in range(n_layer)]
    slow_tok = cap[:F, 0, 2047].astype(np.int64) + IM_END
    o.clear_slow()
    cur, pos, ws, w0, units = padded, 0, 0.0, 0.0, 0.0
    for f in range(F):
        if f > 0:
            for l in range(n_layer):
                o.force_kv(l, gk[l][0][pos], gk[l][1][pos])
        lg, _ = o.forward_generate(cur, pos, full_head=False)
        n = cur.shape[1]
        if f > 0:
            units = max([units] + [o.force_kv_diff(l) for l in range(n_layer)])
        for l in range(n_layer):
            o.set_kv(l, pos, gk[l][0][pos:pos + n], gk[l][1][pos:pos + n])
        d = float(np.abs(lg[0, IM_END:][1:] - cap[f, 0, 1:N_AUDIO]).max())
        if f == 0:
            w0 = d
        else:
            ws = max(ws, d)
        pos += n
        cur = np.array([slow_tok[f]] + [int(v) for v in codes[:, f]], np.uint32).reshape(9, 1)
    assert units <= 16.0, f"a slow-layer K/V entry the row path cached is {units:.2f} units (bf16 ulp, floored at 2^-17) from the oracle's"
    return ws, w0, units


@pytest.mark.parametrize("B,dtype", [(32, "bf16"), (12, "fp8")], ids=["B32", "B12-fp8"])
def test_static_batch_step_kv_forced_oracle(B, dtype):
    """the folded static-batch decode step (round 5: Wqkv with the rms epilogue, k_attn_rows, Wo, W13, k_gemm_down with the in-launch split-K
    sums, head with the rms epilogue; B <= 16 takes the half-panel GEMMs) against the ORACLE at the tolerance of the persistent kernels:
    rows of a BASELINE configs[2]-shaped batch, each teacher-forced on its own sampled tokens and its own cached K/V rows"""
    F = 10
    kw = dict(temp=0.7, top_p=0.8, top_k=256)
    prompts = bench.config2_prompts(TOK, B)
    lens = [p.shape[1] for p in prompts]
    Lmax = max(lens)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, dtype, max_batch=B).load_synthetic(SEED)
    lm.debug_capture(F)
    outs = lm.generate_static_batch(prompts, F + Lmax - 2, seed=9, ignore_eos=True, **kw)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=(dtype == "bf16"), fp8=(dtype == "fp8"))
    o.set_kv_round_bf16(True)
    for b in (0, B // 2 - 1, B - 1):
        padded = np.zeros((9, Lmax), np.uint32)
        padded[0, : Lmax - lens[b]] = IM_END
        padded[:, Lmax - lens[b]:] = prompts[b]
        ws, w0, units = _kv_forced_batch_row(o, lm, b, padded, lm.debug_read_row(b, F), outs[b], fcfg.FISH_1_5["n_layer"])
        _report(f"static-batch step (folded, B {B}, {dtype}): row {b} (L {lens[b]} left-padded to {Lmax}), {F} frames: max |dlogit| slow {ws:.2e} "
                f"(frame 0, own K/V: {w0:.2e}); own K/V rows within {units:.1f} units")
        assert ws < SLOW_TOL and w0 < 1e-2, (b, ws, w0)
    lm.debug_capture(0)
    lm.close()
