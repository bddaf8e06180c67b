"""GPU: decision capture on the ROW path (generate_static_batch / sessions: fs_lm_debug_capture -> fs_lm_debug_read_row) and BASELINE.json
configs[2] as written -- B = 32, prompt lengths U{64..384} (seed 77), temp 0.7 / top-p 0.8 / top-k 256 -- checked decision by decision:
 * every captured logit vector of every row through the oracle's BatchedLogitsProcessor restatement (sampling/mod.rs:77-109: per-call child
   StdRng of every row seeded from the master's next u64): same logits + same stream => the picks must be identical, all 32 x F x 9;
 * the logits of a sample of rows against the CPU oracle teacher-forced on the GPU's tokens (left-padded prompt, static_batch.rs:68-111)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bench
import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED
TOK = fcfg.FISH_1_5_TOKENS
IM_END = TOK["im_end_id"]
N_AUDIO = fcfg.FISH_1_5["vocab_size"] - IM_END
BF16_TOL = 1e-2


def _orc_rows(logits, temp, top_p, top_k, seed, call):
    B, n = logits.shape
    out = np.zeros(B, np.uint32)
    logits = np.ascontiguousarray(logits, np.float32)
    orc.lib().orc_batched_sample(C.c_uint64(seed), C.c_double(temp), C.c_double(top_p), C.c_uint64(top_k),
                                 logits.ctypes.data_as(C.POINTER(C.c_float)), B, n, call, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out.astype(np.int64)


def replay_batch_decisions(lm, B, F, kw, seed, outs, n_audio=N_AUDIO, cb_size=1024):
    """for other test modules: read the row-path capture of the last generate_static_batch call (debug_capture(F) armed before it) and
    check every decision of every row against the oracle batch sampler on the captured logits"""
    return _replay_sampler([lm.debug_read_row(b, F) for b in range(B)], outs, F, kw, seed, n_audio, cb_size)


def _replay_sampler(caps, outs, F, kw, seed, n_audio=N_AUDIO, cb_size=1024):
    """caps: [B][F][9][2048]; every decision of every row == the oracle batch sampler on the captured logits"""
    B = len(caps)
    cap = np.stack(caps)  # (B, F, 9, 2048)
    bad = 0
    for f in range(F):
        for d in range(9):
            n = n_audio if d == 0 else cb_size
            exp = _orc_rows(cap[:, f, d, :n], kw["temp"], kw["top_p"], kw["top_k"], seed, f * 9 + d)
            got = cap[:, f, d, 2047 if d == 0 else 1024].astype(np.int64)
            bad += int((exp != got).sum())
            assert np.array_equal(exp, got), (f, d, np.nonzero(exp != got)[0][:4], exp[:4], got[:4])
    for b in range(B):
        assert np.array_equal(cap[b, :F, 1:, 1024].astype(np.int64).T, outs[b][:, :F].astype(np.int64)), b
    return bad


def _teacher_forced_row(o, padded, cap, codes, F):
    """one row of the static batch on the oracle (its left-padded prompt; no repetition penalty on the batch path)"""
    o.clear_slow()
    femb = o.fast_embeddings()
    slow_tok = cap[:F, 0, 2047].astype(np.int64) + IM_END
    cur, pos, ws, wf = padded, 0, 0.0, 0.0
    for f in range(F):
        lg, hd = o.forward_generate(cur, pos, full_head=False)
        s = lg[0, IM_END:].copy()
        ws = max(ws, float(np.abs(s[1:] - cap[f, 0, 1:N_AUDIO]).max()))
        o.clear_fast()
        x = hd[0]
        for c in range(8):
            fg = o.forward_generate_fast(x, c)[0]
            wf = max(wf, float(np.abs(fg - cap[f, 1 + c, :1024]).max()))
            x = femb[int(codes[c, f])]
        frame = np.array([slow_tok[f]] + [int(v) for v in codes[:, f]], np.uint32)
        pos += cur.shape[1]
        cur = frame.reshape(9, 1)
    return ws, wf


def test_configs2_as_written_every_decision():
    B, F, seed = 32, 32, 42
    kw = dict(temp=0.7, top_p=0.8, top_k=256)
    prompts = bench.config2_prompts(TOK, B)
    lens = [p.shape[1] for p in prompts]
    assert min(lens) >= 64 and max(lens) <= 384
    Lmax = max(lens)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16", max_batch=B).load_synthetic(SEED)
    lm.debug_capture(F)
    outs = lm.generate_static_batch(prompts, F + Lmax - 2, seed=seed, ignore_eos=True, **kw)
    assert all(o.shape == (8, F) for o in outs)
    caps = [lm.debug_read_row(b, F) for b in range(B)]
    lm.debug_capture(0)
    lm.close()
    _replay_sampler(caps, outs, F, kw, seed)
    print(f"configs[2] (B = {B}, prompts {min(lens)}..{Lmax}, {kw}): {B * F * 9} decisions identical to the oracle batch sampler on the captured logits")
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    for b in (0, 13, 31):
        L = lens[b]
        padded = np.zeros((9, Lmax), np.uint32)
        padded[0, : Lmax - L] = IM_END
        padded[:, Lmax - L:] = prompts[b]
        ws, wf = _teacher_forced_row(o, padded, caps[b], outs[b], 12)
        print(f"row {b} (L {L}, left-padded to {Lmax}): max |dlogit| vs the teacher-forced oracle over 12 frames: slow {ws:.2e} fast {wf:.2e}")
        assert ws < BF16_TOL and wf < BF16_TOL, (b, ws, wf)


def test_session_slots_every_decision():
    """continuous batching: slots join and leave; a slot is row 0 of a one-prompt static batch (fishrt.h), but its sampler calls are indexed by
    the SESSION's rows -- replay with B = max_batch on the captured logits of all slots per step"""
    Bs, F, seed = 4, 20, 7
    kw = dict(temp=0.7, top_p=0.8, top_k=256)
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16", max_batch=Bs).load_synthetic(SEED)
    lm.debug_capture(F)
    rs = np.random.RandomState(3)
    prompts = []
    for i in range(Bs):
        q = np.zeros((9, 20 + 9 * i), np.uint32)
        q[0] = rs.randint(0, IM_END, q.shape[1])
        prompts.append(q)
    with lm.session(seed=seed, ignore_eos=True, **kw) as s:
        slots = [s.add(p, p.shape[1] + F - 2) for p in prompts]
        assert sorted(slots) == list(range(Bs))
        while s.step(8):
            pass
        outs = {sl: s.poll(sl)[0] for sl in slots}
    caps = [lm.debug_read_row(b, F) for b in range(Bs)]
    lm.debug_capture(0)
    lm.close()
    got_outs = [outs[b] for b in range(Bs)]
    assert all(o.shape == (8, F) for o in got_outs)
    _replay_sampler(caps, got_outs, F, kw, seed)
    print(f"session, {Bs} slots x {F} frames: every decision identical to the oracle batch sampler on the captured logits")
