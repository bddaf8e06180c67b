"""GPU: the persistent decode kernels the bench times (k_slow_persist / k_fast_persist, and the request-row kernels) against the CPU oracle
at a tolerance that separates "right" from "nearly right" (What's weak 1).

The bf16 protocol of the other tests (BF16_TOL = 1e-2) is as loose as it is because bf16 K/V entries sit on rounding boundaries: a 1e-6
difference in the f32 value (summation order) flips an entry by one bf16 ulp (0.4 %), the flipped entry feeds every later step, and 24
layers compound it to ~7e-3 on logits of scale 3.  Here the oracle is teacher-forced on the GPU's tokens AND on the GPU's own cached K/V
(fs_lm_debug_read_kv -> OracleLM.set_kv): every cached row the oracle attends over is bit-for-bit the row the kernel attended over, so
what separates the two logit vectors is the summation order of the CURRENT step only (plus the current token's own K/V entries, 1 row of
T).  A summation-order bug of 1e-3 in any stage of the persistent slow kernel shows up as a failure of this test; it would pass the
1e-2 ones."""
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
FAST_TOL = 2e-4   # fast logits: the fast decoder's per-frame K/V rows ride in the capture record (raw bf16 pairs) and are forced the same way


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
    for f in range(F):
        lg, hd = o.forward_generate(cur, pos, full_head=False)
        n = cur.shape[1]
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
    # one bf16 ulp (a rounding-boundary flip), or -- for entries below ~1e-3, where the unit is floored at 2^-17 -- the 3e-5 that the
    # hidden state's own 5e-5 leaves on a projection
    assert kv_ulp[0] <= 4.0, f"a fast-decoder K/V entry of the kernel is {kv_ulp[0]:.2f} units (bf16 ulp, floored at 2^-17) from the oracle's"
    return ws, w0, wf


def test_persistent_kernels_kv_forced_oracle_configs1():
    """configs[1] prompt (367 positions), 128 frames on the two persistent launches per frame"""
    F, rp = 128, 1.2
    p = bench.default_voice_prompt(TOK)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16").load_synthetic(SEED)
    lm.debug_capture(F)
    codes = lm.generate_blocking(p, F + p.shape[1] - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=rp, ignore_eos=True)
    assert codes.shape == (8, F) and lm.last_stats()["kernels_per_frame"] == 2
    cap = lm.debug_read(F)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    ws, w0, wf = _kv_forced_replay(o, lm, 0, p, cap, codes, rp, fcfg.FISH_1_5["n_layer"])
    lm.close()
    print(f"k_slow_persist / k_fast_persist, {F} frames, oracle forced on the GPU's tokens and cached K/V: max |dlogit| slow {ws:.2e} (frame 0, "
          f"own K/V: {w0:.2e}), fast {wf:.2e}  [tolerances {SLOW_TOL:.0e} / {FAST_TOL:.0e}; the unforced protocol allows 1e-2]")
    assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2


def test_row_kernels_kv_forced_oracle():
    """the request-row kernels (k_slow_rows / k_fast_rows), 4 rows x 48 frames"""
    F, rp = 48, 1.2
    lens = [40, 130, 77, 250]
    prompts = []
    for i, L in enumerate(lens):
        q = np.zeros((9, L), np.uint32)
        q[0] = np.random.RandomState(900 + i).randint(0, IM_END, L)
        prompts.append(q)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16", max_batch=4).load_synthetic(SEED)
    lm.debug_capture(F)
    got = lm.generate_multi(prompts, [L + F - 2 for L in lens], temp=0.0, top_p=1.0, top_k=0, repetition_penalty=rp, ignore_eos=True)
    assert lm.last_stats()["kernels_per_frame"] == 2
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    for i in range(4):
        cap = lm.debug_read_row(i, F)
        ws, w0, wf = _kv_forced_replay(o, lm, i, prompts[i], cap, got[i], rp, fcfg.FISH_1_5["n_layer"])
        print(f"row {i} (L {lens[i]}): max |dlogit| slow {ws:.2e} (frame 0: {w0:.2e}), fast {wf:.2e}")
        assert ws < SLOW_TOL and wf < FAST_TOL and w0 < 1e-2, (i, ws, w0, wf)
    lm.close()
