"""GPU: the persistent decode kernels (csrc/lm_persist_slow.hip: the 24 slow blocks + head of a step as one launch, any sampler;
csrc/lm_persist.hip: the 8 codebook passes of the fast decoder as one launch, greedy decoding; both default on bf16 Fish-geometry handles) against
(1) the per-node graph path of the same handle (FS_GEN_NO_PERSIST), token for token, and (2) the CPU oracle under the bf16 protocol
of DESIGN.md (divergence only at a near-tie the oracle reports).  Plus its building blocks and its exclusivity rule."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import _ffi, config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED
BF16_TOL = 1e-2
GREEDY = dict(temp=0.0, top_p=1.0, top_k=0, ignore_eos=True)


@pytest.fixture(scope="module")
def lm15():
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(SEED)
    yield lm
    lm.close()


def _text_prompt(L, seed):
    rng = np.random.RandomState(seed)
    p = np.zeros((9, L), np.uint32)
    p[0] = rng.randint(0, fcfg.FISH_1_5_TOKENS["im_end_id"], L)
    return p


def _vq_prompt(L, seed):
    """text, then a span of semantic tokens with their 8 codebook rows (the embed() mask is live), then text"""
    tok = fcfg.FISH_1_5_TOKENS
    rng = np.random.RandomState(seed)
    p = _text_prompt(L, seed + 1)
    a, b = L // 4, 3 * L // 4
    codes = rng.randint(0, 1024, (8, b - a))
    p[0, a:b] = tok["semantic_start_id"] + codes[0]
    p[1:, a:b] = codes
    return p


class _RepPen:  # python twin of rep_pen.rs:37-65 for the teacher-forced replay
    def __init__(self, n, amt):
        self.mask, self.ctx, self.seen, self.amt = np.ones(n, np.float32), [], set(), np.float32(amt)

    def apply(self, logits, last):
        self.seen.add(last); self.mask[last] = self.amt
        self.ctx.insert(0, last)
        if len(self.ctx) > 16:
            d = self.ctx.pop()
            if d in self.seen:
                self.seen.discard(d); self.mask[d] = 1.0
        return logits / self.mask


def _argmax_last(v):
    return int(np.nonzero(v == v.max())[0][-1])


def _replay_gap(lm, p, a, b, rep_pen):
    """Referee for a divergence between the two paths: replay path A's stream through the handle's teacher-forced per-node API
    (forward_generate / forward_generate_fast: the same kernels the per-node graph runs) up to the first differing decision and
    return the penalised-logit gap between the two paths' choices there.  The replay must reproduce A's own choices on the way."""
    im_end = fcfg.FISH_1_5_TOKENS["im_end_id"]
    f = int(np.argmax((a != b).any(0)))
    cbd = int(np.argmax(a[:, f] != b[:, f]))
    lm.clear_slow_layer_caches()
    cur, pos, prev = p, 0, None
    rps = [_RepPen(1024, rep_pen) for _ in range(8)]
    slow_margin = []
    for it in range(f + 1):
        lg, hg = lm.forward_generate(cur, pos)
        s = lg[0, im_end:].copy()
        s[0] = -np.inf  # ignore_eos
        srt = np.sort(s[np.isfinite(s)])
        slow_margin.append(float(srt[-1] - srt[-2]))
        frame = [_argmax_last(s) + im_end]
        lm.clear_fast_layer_caches()
        x = hg
        for ci in range(8):
            fg = lm.forward_generate_fast(x, ci)[0]
            if prev is not None:
                fg = rps[ci].apply(fg, prev[ci + 1])
            if it == f and ci == cbd:
                gap = float(abs(fg[a[ci, f]] - fg[b[ci, f]]))
                # the codes do not carry the slow token: the paths may have parted at the SLOW decision of frame f or f - 1 (whose eight codes
                # can still coincide): a near-tie of the slow logits there is that parting
                gap = min([gap] + slow_margin[-2:])
                return f, cbd, gap, float(np.sort(fg)[-1] - np.sort(fg)[-2])
            assert _argmax_last(fg) == a[ci, it], f"replay lost path A at frame {it} codebook {ci}"
            frame.append(int(a[ci, it]))
            x = lm.fast_embeddings([int(a[ci, it])])
        pos += cur.shape[1]
        prev = frame
        cur = np.array(frame, np.uint32).reshape(9, 1)
    raise AssertionError("no differing decision found")


def test_reduction_trees_selftest():
    _ffi.check(_ffi.lib().fs_selftest(0, b"pf_reduce"))
    assert _ffi.lib().fs_selftest(0, b"no-such-test") != 0


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_handle_selftest_persist(dtype):
    """fs_lm_selftest("persist"): the persistent kernels of this binary against the per-node kernels on the handle's own weights --
    what a deployment runs after a toolchain change (unseen loads / pinned weight registers depend on hipcc's register allocation)."""
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, dtype).load_synthetic(0xF15E5EED)
    lm.selftest("persist")
    assert lm.curr_kv_size() == 0  # the handle is handed back with empty caches
    assert _ffi.lib().fs_lm_selftest(lm._h, b"no-such-test") != 0
    # the handle still generates, on the persistent path, after the self-test
    p = _text_prompt(16, 1234)
    out = lm.generate_blocking(p, 16 + 6, repetition_penalty=1.2, persistent=True, **GREEDY)
    assert out.shape == (8, 8) and lm.last_stats()["kernels_per_frame"] == 2


def test_handle_selftest_persist_legacy_fp8():
    """the same on a Fish-1.4-shaped fp8 handle (configs[4]: 2-way {pad, im_end} slow decision inside k_fast_persist)"""
    lm = fishrt.DualARTransformer(fcfg.FISH_1_4, fcfg.FISH_1_4_TOKENS, 0, "fp8").load_synthetic(0xF15E5EED)
    lm.selftest("persist")


@pytest.mark.parametrize("rep_pen", [1.0, 1.2])
def test_persistent_equals_per_node_path(lm15, rep_pen):
    # KV lengths 16..4300: attention stage with 1, 2, 4, 16 token slices per head, one to three 128-token tiles per workgroup
    for p in (_text_prompt(16, 1234), _vq_prompt(96, 7), _text_prompt(200, 99), _text_prompt(1100, 5), _vq_prompt(4200, 6)):
        L = p.shape[1]
        lm15.clear_slow_layer_caches()
        a = lm15.generate_blocking(p, L + 62, repetition_penalty=rep_pen, persistent=False, **GREEDY)
        assert lm15.last_stats()["kernels_per_frame"] == 266
        lm15.clear_slow_layer_caches()
        b = lm15.generate_blocking(p, L + 62, repetition_penalty=rep_pen, persistent=True, **GREEDY)
        assert lm15.last_stats()["kernels_per_frame"] == 2, "the persistent launches were not taken"
        assert a.shape == b.shape == (8, 64)
        if not np.array_equal(a, b):
            # the two paths sum in different orders; an f32 rounding difference in a new K / V element can flip its bf16 rounding in the
            # cache (2^-9 relative) and feed back through up to 24 blocks, so the logits of the two paths differ by up to ~1e-3 (measured
            # 1.3e-3 at logit scale 3 with both kernels persistent, 1.3e-4 with the fast decoder alone; each path is within BF16_TOL = 1e-2
            # of the oracle): they may part ways only where their two choices are that close
            f, cbd, gap, top2 = _replay_gap(lm15, p, a, b, rep_pen)
            print(f"L={L} rep_pen={rep_pen}: paths part at frame {f} codebook {cbd}: gap between the two choices {gap:.2e} (top-2 margin {top2:.2e})")
            assert gap < 5e-3, (f, cbd, gap)  # (where they part is data; THAT they part only on a near-tie is the check)
        else:
            print(f"L={L} rep_pen={rep_pen}: 64/64 frames identical")


@pytest.mark.parametrize("persistent", [True, False])
def test_free_running_greedy_vs_oracle(lm15, persistent):
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    p = _text_prompt(16, 1234)
    M = 16 + 46
    lm15.clear_slow_layer_caches()
    got = lm15.generate_blocking(p, M, repetition_penalty=1.2, persistent=persistent, **GREEDY)
    exp = o.generate(p, M, temp=0.0, repetition_penalty=1.2, ignore_eos=True)
    assert got.shape == exp.shape == (8, 48)
    bad = np.nonzero((got != exp).any(0))[0]
    if bad.size:
        f = int(bad[0])
        assert o.last_margins[f] < BF16_TOL, f"diverged at frame {f} on a margin of {o.last_margins[f]:.2e}"
        print(f"persistent={persistent}: identical for {f} frames, then a near-tie (margin {o.last_margins[f]:.2e})")
        assert f >= 24
    else:
        print(f"persistent={persistent}: all 48 frames identical to the oracle")


def test_slow_kernel_hidden_states_vs_per_node_path_and_oracle(lm15):
    """the persistent slow kernel's output itself: the hidden state of every frame (generate_blocking_with_hidden) against the per-node
    path (same handle) and against the bf16-mode oracle, at KV lengths that exercise 1, 4 and 16 token slices per head"""
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    for L, seed in ((16, 1234), (300, 8), (1300, 9)):
        p = _text_prompt(L, seed)
        kw = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
        lm15.clear_slow_layer_caches()
        ca, ha = lm15.generate_blocking_with_hidden(p, L + 10, True, persistent=False, **kw)
        lm15.clear_slow_layer_caches()
        cb, hb = lm15.generate_blocking_with_hidden(p, L + 10, True, persistent=True, **kw)
        o.clear_slow()
        ce, he = o.generate(p, L + 10, temp=0.0, repetition_penalty=1.2, ignore_eos=True, collect_hidden=True)
        n = 12
        same_ab = n if np.array_equal(ca, cb) else int(np.argmax((ca != cb).any(0)))
        same_be = n if np.array_equal(cb, ce) else int(np.argmax((cb != ce).any(0)))
        rms = float(np.sqrt(np.mean(he ** 2)))
        d_ab = max(float(np.abs(ha[f, 0] - hb[f, 0]).max()) / rms for f in range(same_ab + 1 if same_ab < n else n))
        d_be = max(float(np.abs(hb[f, 0] - he[f]).max()) / rms for f in range(same_be + 1 if same_be < n else n))
        print(f"L={L}: hidden |d|/rms persistent vs per-node {d_ab:.2e} over {same_ab} identical frames; persistent vs oracle {d_be:.2e} over {same_be}")
        assert d_ab < 5e-3 and d_be < BF16_TOL, (L, d_ab, d_be)
        assert same_ab >= 1 and same_be >= 1


def test_eos_and_budget_semantics_match_per_node_path(lm15):
    """no ignore_eos: runs that sample <|im_end|> stop at the same frame with the same codes on both paths (first frame recorded
    unconditionally, zeros pushed for the terminating frame: single_batch.rs:153-156,250,264-266); runs that do not, fill the
    budget.  Where the two paths part (their logits differ by ~1e-3: summation order, bf16 K/V rounding flips), the decision at which they
    part is refereed on the logits the persistent path recorded for it (fs_lm_debug_capture): the two choices must be < 5e-3 apart."""
    same_eos = same_full = parted = 0
    lm15.debug_capture(80)
    try:
        for seed in range(40):
            p = _text_prompt(12, 1000 + seed)
            outs, kv = [], []
            for persistent in (False, True):
                lm15.clear_slow_layer_caches()
                outs.append(lm15.generate_blocking(p, 12 + 70, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, persistent=persistent))
                kv.append(lm15.curr_kv_size())
            a, b = outs
            if a.shape == b.shape and np.array_equal(a, b):
                assert kv[0] == kv[1], "KV length after the call differs between the paths"
                same_eos += a.shape[1] < 72
                same_full += a.shape[1] == 72
                continue
            parted += 1
            cap = lm15.debug_read(80)  # what the persistent path's decisions saw
            n = min(a.shape[1], b.shape[1])
            if np.array_equal(a[:, :n], b[:, :n]):  # one path sampled <|im_end|> at iteration n where the other went on: the slow decision
                s = cap[n, 0, :2037]
                gap, what = float(abs(s[0] - s[1:].max())), f"<|im_end|> decision of iteration {n}"
            else:
                f = int(np.argmax((a[:, :n] != b[:, :n]).any(0)))
                c = int(np.argmax(a[:, f] != b[:, f]))
                lg = cap[f, 1 + c, :1024]
                gap, what = float(abs(lg[a[c, f]] - lg[b[c, f]])), f"frame {f} codebook {c}"
                if gap >= 5e-3:
                    # the codes do not carry the slow token: the paths may have parted at the SLOW decision of frame f or f - 1 (whose eight
                    # codes can still coincide) -- a near-tie of the slow logits the persistent path recorded there is that parting
                    for g in (f, f - 1):
                        if g >= 0:
                            sl = np.sort(cap[g, 0, :2037][np.isfinite(cap[g, 0, :2037])])
                            if float(sl[-1] - sl[-2]) < gap:
                                gap, what = float(sl[-1] - sl[-2]), f"slow decision of frame {g} (codes first differ at frame {f})"
            assert gap < 5e-3, (seed, what, gap)
    finally:
        lm15.debug_capture(0)
    print(f"identical runs: {same_eos} ended on <|im_end|> before the budget, {same_full} filled it; {parted} parted, every one at a refereed near-tie (< 5e-3)")
    # tripwire next to the referee: 12-token prompts on flat synthetic logits part in roughly 25-32 of 40 runs over 70 frames
    # (measured r2..r4); all 40 parting would point at a systematic bias below the 5e-3 referee
    assert parted <= 38, parted
    assert same_eos >= 1, "no run sampled <|im_end|>: the EOS branch of the persistent kernel went unexercised"


def test_only_one_handle_per_gpu_takes_the_persistent_launch(lm15):
    """two handles generating at the same time: the persistent launch needs every CU, so the second call falls back to the per-node
    graph -- both finish (no grid-wide wait on CUs the other launch holds) with the tokens of a solo run"""
    lm2 = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(SEED)
    p = _text_prompt(32, 3)
    lm15.clear_slow_layer_caches()
    ref = lm15.generate_blocking(p, 32 + 126, repetition_penalty=1.2, **GREEDY)
    res, kpf = {}, {}

    def work(name, lm):
        for _ in range(3):
            lm.clear_slow_layer_caches()
            res[name] = lm.generate_blocking(p, 32 + 126, repetition_penalty=1.2, **GREEDY)
            kpf.setdefault(name, set()).add(lm.last_stats()["kernels_per_frame"])

    ths = [threading.Thread(target=work, args=("a", lm15)), threading.Thread(target=work, args=("b", lm2))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a generate call hung"
    assert np.array_equal(res["a"], ref) and np.array_equal(res["b"], ref)
    print("kernels per frame seen:", kpf)
    assert kpf["a"] | kpf["b"] <= {2, 266} and 2 in (kpf["a"] | kpf["b"])
    lm2.close()


def test_sampled_calls_take_the_persistent_launches_too(lm15):
    """temp > 0 with top_k <= 256 (the server default): both persistent kernels, the decisions by the in-launch block-parallel sampler
    (refereed decision by decision in tests/test_persist_sampled_gpu.py); other sampler settings keep the per-node fast decoder behind
    the persistent slow kernel.  The sampled stream equals the all-per-node one except where a draw lands on a CDF boundary (the two
    paths' logits differ by summation order, ~1e-3)."""
    p = _text_prompt(16, 11)
    kw = dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.2, seed=3, ignore_eos=True)
    lm15.clear_slow_layer_caches()
    a = lm15.generate_blocking(p, 16 + 30, **kw)
    assert lm15.last_stats()["kernels_per_frame"] == 2
    lm15.clear_slow_layer_caches()
    a2 = lm15.generate_blocking(p, 16 + 30, **kw)
    assert np.array_equal(a, a2), "sampled decoding with a fixed seed must be deterministic"
    lm15.clear_slow_layer_caches()
    b = lm15.generate_blocking(p, 16 + 30, persistent=False, **kw)
    assert lm15.last_stats()["kernels_per_frame"] == 266
    same = int(np.argmax((a != b).any(0))) if (a != b).any() else a.shape[1]
    print(f"sampled: {same}/{a.shape[1]} frames identical between the paths")
    assert same >= 1
    lm15.clear_slow_layer_caches()
    c = lm15.generate_blocking(p, 16 + 30, temp=0.7, top_p=0.8, top_k=0, repetition_penalty=1.2, seed=3, ignore_eos=True)  # no top-k: general sampler
    assert lm15.last_stats()["kernels_per_frame"] == 2 + 8 * 18 and c.shape == a.shape
