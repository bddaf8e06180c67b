"""GPU: the in-launch decisions of the persistent batch-1 decode path (lm_persist.hip: slow-token decision in the prologue + 8 codebook
decisions, greedy or the block-parallel top-k / top-p sampler of lm_bsample_dev.h) refereed by the CPU oracle's sampler.

fs_lm_debug_capture records the logits every decision saw (audio-range slow logits after the <|im_end|> mask, codebook logits after the
repetition penalty) and the index it picked.  The oracle's LogitsProcessor, seeded like the request, is fed exactly those rows in the
order the generator samples them (slow, c0..c7 per frame: single_batch.rs:102-183): same logits + same StdRng stream => the picks
must be identical token for token -- no near-tie excuse, the comparison is on the GPU's own logits."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED
TOK = fcfg.FISH_1_5_TOKENS
N_AUDIO = fcfg.FISH_1_5["vocab_size"] - TOK["im_end_id"]


@pytest.fixture(scope="module")
def lm15():
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, TOK, 0, "bf16").load_synthetic(SEED)
    yield lm
    lm.close()


def _prompt(L, seed):
    rs = np.random.RandomState(seed)
    p = np.zeros((9, L), np.uint32)
    p[0] = rs.randint(0, TOK["im_end_id"], L)
    return p


def _oracle_picks(cap, seed, temp, top_p, top_k):
    L = orc.lib()
    s = L.orc_sampler_create(C.c_uint64(seed), C.c_double(temp), C.c_double(top_p), C.c_uint64(top_k))
    picks = np.zeros((cap.shape[0], 9), np.int64)
    try:
        for f in range(cap.shape[0]):
            for r in range(9):
                n = N_AUDIO if r == 0 else 1024
                row = np.ascontiguousarray(cap[f, r, :n])
                picks[f, r] = L.orc_sampler_sample(C.c_void_p(s), row.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(n))
    finally:
        L.orc_sampler_destroy(C.c_void_p(s))
    return picks


@pytest.mark.parametrize("kw", [dict(temp=0.7, top_p=0.8, top_k=256), dict(temp=0.7, top_p=0.9, top_k=50), dict(temp=1.0, top_p=0.3, top_k=256),
                                dict(temp=1.3, top_p=1.0, top_k=200), dict(temp=0.0, top_p=1.0, top_k=0),
                                # PEAKED rows: the synthetic weights give flat logits; a trained head does not.  The sampler sees
                                # logits / temp, so temp = 0.7 / 8 IS "output heads scaled x 8 at the server's temp 0.7" (a power-of-two scale is exact),
                                # and temp 0.02 is the regime where the largest probability alone exceeds top_p (one-weight shortcut) on most rows
                                dict(temp=0.0875, top_p=0.8, top_k=256), dict(temp=0.02, top_p=0.8, top_k=256), dict(temp=0.05, top_p=0.95, top_k=40)])
@pytest.mark.parametrize("rep_pen", [1.0, 1.2])
def test_in_launch_decisions_equal_the_oracle_sampler_on_the_same_logits(lm15, kw, rep_pen):
    F, seed = 40, 1234
    p = _prompt(24, 5)
    lm15.debug_capture(F)
    try:
        lm15.clear_slow_layer_caches()
        codes = lm15.generate_blocking(p, 24 + F - 2, repetition_penalty=rep_pen, seed=seed, ignore_eos=True, **kw)
        assert lm15.last_stats()["kernels_per_frame"] == 2, "the persistent launches were not taken"
        assert codes.shape == (8, F)
        cap = lm15.debug_read(F)
    finally:
        lm15.debug_capture(0)
    got = np.concatenate([cap[:, :1, 2047], cap[:, 1:, 1024]], axis=1).astype(np.int64)  # [F][9] picks recorded by the kernel
    assert np.array_equal(got[:, 1:].T, codes.astype(np.int64)), "captured picks are not the generated codes"
    assert np.isneginf(cap[:, 0, 0]).all(), "ignore_eos must mask the <|im_end|> logit"
    exp = _oracle_picks(cap, seed, kw["temp"], kw["top_p"], kw["top_k"])
    bad = np.argwhere(got != exp)
    assert bad.size == 0, f"{len(bad)} of {F * 9} decisions differ from the oracle sampler, first (frame, decision) {bad[0]}: gpu {got[tuple(bad[0])]} oracle {exp[tuple(bad[0])]}"
    if rep_pen != 1.0:  # the repetition penalty really was applied to what the sampler saw: a penalised entry is logit / 1.2
        assert not np.array_equal(cap[1, 1:, :1024], cap[0, 1:, :1024])
    print(f"{kw} rep_pen={rep_pen}: {F * 9}/{F * 9} decisions identical to the oracle sampler on the captured logits")
