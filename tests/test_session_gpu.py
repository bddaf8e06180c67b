"""GPU: continuous batching (fs_lm_session_*, include/fishrt.h) -- the rows of the static-batch step as independent request slots.
A slot must generate what a ONE-prompt generate_static_batch generates (oracle: OracleLM.generate_batch([prompt])), whatever the
other slots do: requests that join mid-flight, finish at different frames, slots that are released and taken again.
Greedy: token-identical to the oracle except where the oracle's own top-2 margin is a near-tie (bf16 row path, see test_lm_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc

SEED = 0xF15E5EED
NEAR_TIE = 5e-3
MID = dict(fcfg.TINY, dim=256, n_head=4, n_local_heads=2, head_dim=64, intermediate_size=1024)


def _prompt(rng, L):
    p = np.zeros((9, L), np.uint32)
    p[0] = rng.randint(0, 400, L)
    k = min(L - 1, 5)
    if k > 0:  # a VQ span so that codebook embeddings take part
        p[0, 1 : 1 + k] = fcfg.TINY_TOKENS["semantic_start_id"] + rng.randint(0, 64, k)
        p[1:, 1 : 1 + k] = rng.randint(0, 64, (8, k))
    return p


def _check_vs_oracle(o, prompt, max_new, got, what, **kw):
    exp = o.generate_batch([prompt], max_new, seed=42, temp=0.0, top_p=1.0, top_k=0, **kw)[0]
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    if np.array_equal(got, exp):
        return 0
    m = o.last_batch_margins
    f = int(np.argmax((got != exp).any(0)))
    mf = float(min(m[f, 0], m[f - 1, 0])) if f > 0 else float(m[f, 0])
    assert mf < NEAR_TIE, f"{what} left the oracle's stream at frame {f} on a margin of {mf:.2e}"
    return 1


@pytest.mark.parametrize("cfg,dtype", [(fcfg.TINY, "bf16"), (MID, "bf16"), (MID, "fp8")], ids=["hd32", "hd64", "hd64-fp8"])
def test_slots_equal_one_prompt_static_batches_whatever_the_neighbours_do(cfg, dtype):
    lm = fishrt.DualARTransformer(cfg, fcfg.TINY_TOKENS, 0, dtype, max_batch=4).load_synthetic(SEED)
    o = orc.OracleLM(orc.TINY | {k: cfg[k] for k in ("dim", "n_head", "n_local_heads", "head_dim", "intermediate_size")})
    o.load_synthetic(SEED, bf16=dtype == "bf16", fp8=dtype == "fp8")  # fp8: the oracle's FS_FP8 weight mode (same bytes and row scales)
    o.set_kv_round_bf16(True)
    rng = np.random.RandomState(7)
    lens = [9, 1, 40, 17, 23, 5, 64, 12, 31]
    prompts = [_prompt(rng, L) for L in lens]
    budgets = [L + int(rng.randint(10, 60)) for L in lens]  # max_new_tokens: 1 + max(0, max_new - L + 1) iterations each
    results, flips = {}, 0
    with lm.session(temp=0.0, top_p=1.0, top_k=0, seed=42, ignore_eos=True) as s:
        pending, live = list(range(len(prompts))), {}
        with pytest.raises(RuntimeError):  # the handle's other entry points are closed while the session is open
            lm.generate_blocking(prompts[0], 20)
        steps = 0
        while pending or live:
            while pending:  # admit as many as fit; the fifth finds the 4 slots busy
                slot = s.add(prompts[pending[0]], budgets[pending[0]])
                if slot is None:
                    assert len(live) == 4
                    break
                assert slot not in live
                live[slot] = pending.pop(0)
            s.step(int(rng.randint(1, 9)))  # odd step sizes: budgets end in the middle of a step call
            steps += 1
            for slot in list(live):
                n, done = s.poll(slot, codes=False)
                if done:
                    codes, _ = s.poll(slot)
                    results[live.pop(slot)] = codes
                    s.release(slot)  # the freed slot is taken by a waiting request in the next round
        assert steps > 8
    for i, p in enumerate(prompts):
        assert results[i].shape == (8, 1 + max(0, budgets[i] - lens[i] + 1)), i
        flips += _check_vs_oracle(o, p, budgets[i], results[i], f"request {i}", ignore_eos=True)
    print(f"session: {len(prompts) - flips}/{len(prompts)} requests identical to their one-prompt static batch, {flips} left it at a near-tie")
    assert flips <= 3
    # the handle works normally again after the session, with every KV page back in the pool
    lm.clear_slow_layer_caches()
    a = lm.generate_static_batch(prompts[:4], 60, temp=0.0, top_p=1.0, top_k=0, ignore_eos=True)
    assert all(x.shape[1] > 0 for x in a)
    lm.close()


def test_eos_ends_a_slot_and_frames_stop():
    """without ignore_eos: a slot that samples <|im_end|> freezes on the device (no further frames, position fixed) while the others go on"""
    lm = fishrt.DualARTransformer(fcfg.TINY, fcfg.TINY_TOKENS, 0, "bf16", max_batch=4).load_synthetic(SEED)
    o = orc.OracleLM(orc.TINY).load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    rng = np.random.RandomState(11)
    prompts = [_prompt(rng, L) for L in (6, 14, 22, 9, 30, 11, 8, 19)]
    exp = [o.generate_batch([p], 150, seed=42, temp=0.0, top_p=1.0, top_k=0)[0] for p in prompts]
    early = [i for i, e in enumerate(exp) if e.shape[1] < 150 - prompts[i].shape[1] + 2]
    assert early, "no fixture request reaches <|im_end|>: EOS path untested"
    got = {}
    with lm.session(temp=0.0, top_p=1.0, top_k=0) as s:
        for base in (0, 4):
            slots = {s.add(prompts[base + j], 150): base + j for j in range(4)}
            while s.step(16):
                pass
            for slot, i in slots.items():
                codes, done = s.poll(slot)
                assert done
                n0 = codes.shape[1]
                got[i] = codes
                s.step(4)  # nothing is live: no launches, nothing changes
                assert s.poll(slot, codes=False) == (n0, True)
                s.release(slot)
    same = sum(int(g.shape == e.shape and np.array_equal(g, e)) for g, e in ((got[i], exp[i]) for i in range(len(prompts))))
    for i in range(len(prompts)):
        n = min(got[i].shape[1], exp[i].shape[1], 6)
        assert np.array_equal(got[i][:, :n], exp[i][:, :n]), i
    print(f"session EOS: {same}/{len(prompts)} requests identical incl. where <|im_end|> falls; early stops in the oracle: {early}")
    assert same >= len(prompts) - 3
    lm.close()


def test_sampled_session_full_size_shapes():
    """Fish-1.5 shapes, 8 slots, top-k 256 / top-p 0.8 sampling, more requests than slots: every request gets its budget of valid codes"""
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(SEED)
    rng = np.random.RandomState(3)
    im_end = fcfg.FISH_1_5_TOKENS["im_end_id"]
    reqs = []
    for i in range(20):
        L = int(rng.randint(16, 200))
        p = np.zeros((9, L), np.uint32)
        p[0] = rng.randint(0, im_end, L)
        reqs.append((p, L + int(rng.randint(8, 48))))
    done_codes = {}
    with lm.session(temp=0.7, top_p=0.8, top_k=256, seed=1, ignore_eos=True) as s:
        pending, live = list(range(len(reqs))), {}
        while pending or live:
            while pending and (slot := s.add(*reqs[pending[0]])) is not None:
                live[slot] = pending.pop(0)
            s.step(8)
            for slot in list(live):
                if s.poll(slot, codes=False)[1]:
                    done_codes[live.pop(slot)] = s.poll(slot)[0]
                    s.release(slot)
        st = lm.last_stats()
    for i, (p, mx) in enumerate(reqs):
        assert done_codes[i].shape == (8, mx - p.shape[1] + 2) and done_codes[i].max() < 1024
    assert st["frames"] == sum(c.shape[1] for c in done_codes.values())
    print(f"sampled session: {st['frames']} frames in {st['decode_ms']:.1f} ms of decode steps ({st['graph_launches']} launches)")
    lm.close()


def test_fullsize_greedy_session_vs_oracle_one_prompt_batches():
    """Fish-1.5 shapes (dim 1024, 24 + 4 layers, audio range 2037, codebooks 1024), bf16, 8 slots, 5 requests joining at different steps:
    every slot == the oracle's one-prompt generate_static_batch on the same bf16-rounded weights, except at oracle near-ties"""
    SEEDW = 0xF15E5EED
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16", max_batch=8).load_synthetic(SEEDW)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEEDW, bf16=True)
    o.set_kv_round_bf16(True)
    rng = np.random.RandomState(21)
    im_end, sem0 = fcfg.FISH_1_5_TOKENS["im_end_id"], fcfg.FISH_1_5_TOKENS["semantic_start_id"]
    reqs = []
    for L, frames in ((12, 14), (31, 10), (7, 16), (20, 12), (1, 9)):
        p = np.zeros((9, L), np.uint32)
        p[0] = rng.randint(0, im_end, L)
        k = min(L - 1, 4)
        if k > 0:
            p[0, 1 : 1 + k] = sem0 + rng.randint(0, 1024, k)
            p[1:, 1 : 1 + k] = rng.randint(0, 1024, (8, k))
        reqs.append((p, frames + L - 2))
    got = {}
    with lm.session(temp=0.0, top_p=1.0, top_k=0, ignore_eos=True) as s:
        live, nxt = {}, 0
        while nxt < len(reqs) or live:
            if nxt < len(reqs):  # one new request per round: each joins while the earlier ones are mid-flight
                live[s.add(*reqs[nxt])] = nxt
                nxt += 1
            s.step(3)
            for slot in list(live):
                if s.poll(slot, codes=False)[1]:
                    got[live.pop(slot)] = s.poll(slot)[0]
                    s.release(slot)
    flips = 0
    for i, (p, mx) in enumerate(reqs):
        flips += _check_vs_oracle(o, p, mx, got[i], f"full-size request {i}", ignore_eos=True)
    print(f"full-size session: {len(reqs) - flips}/{len(reqs)} requests identical to the oracle's one-prompt static batch")
    assert flips <= 2
    lm.close()


def test_budget_exhausted_slot_on_a_page_boundary_leaves_its_neighbours_alone():
    """A slot whose budget ends with pos == a multiple of the KV page size (64) keeps riding the step graphs, frozen, until it is released;
    its K/V write must be parked (scratch page) -- its own page table has no entry for that position, and a stale entry would point into
    a page that now belongs to a neighbour.  Slots: A ends exactly on the boundary and is released late, B / C / D keep
    running meanwhile, D re-uses pages freed earlier; every request must still equal its one-prompt static batch."""
    lm = fishrt.DualARTransformer(MID, fcfg.TINY_TOKENS, 0, "bf16", max_batch=4).load_synthetic(SEED)
    o = orc.OracleLM(orc.TINY | {k: MID[k] for k in ("dim", "n_head", "n_local_heads", "head_dim", "intermediate_size")})
    o.load_synthetic(SEED, bf16=True)
    o.set_kv_round_bf16(True)
    rng = np.random.RandomState(11)
    # request: (L, max_new_tokens); iterations = 1 + max(0, M - L + 1); final pos = L - 1 + iterations
    reqs = {"E": (33, 33 + 20), "A": (20, 63), "B": (30, 30 + 150), "C": (7, 7 + 140), "D": (50, 50 + 60)}
    assert reqs["A"][0] - 1 + (1 + reqs["A"][1] - reqs["A"][0] + 1) == 64
    prompts = {k: _prompt(rng, L) for k, (L, _) in reqs.items()}
    out = {}
    with lm.session(temp=0.0, top_p=1.0, top_k=0, seed=42, ignore_eos=True) as s:
        slot = {}
        slot["E"] = s.add(prompts["E"], reqs["E"][1])  # E runs first and is released: its pages go back to the pool (LIFO)
        while not s.poll(slot["E"], codes=False)[1]:
            s.step(4)
        out["E"] = s.poll(slot["E"])[0]
        s.release(slot["E"])
        for k in ("A", "B", "C"):
            slot[k] = s.add(prompts[k], reqs[k][1])
        released_a = False
        d_added = False
        for _ in range(80):
            s.step(3)
            if not d_added and s.poll(slot["A"], codes=False)[1]:  # A is done (frozen on the boundary) but NOT released yet
                slot["D"] = s.add(prompts["D"], reqs["D"][1])
                d_added = True
            if d_added and not released_a and s.poll(slot["D"], codes=False)[0] > 30:
                out["A"] = s.poll(slot["A"])[0]
                s.release(slot["A"])
                released_a = True
            if all(s.poll(slot[k], codes=False)[1] for k in ("B", "C")) and d_added and s.poll(slot["D"], codes=False)[1]:
                break
        for k in ("B", "C", "D"):
            out[k] = s.poll(slot[k])[0]
        assert released_a and d_added
    flips = sum(_check_vs_oracle(o, prompts[k], reqs[k][1], out[k], f"request {k}", ignore_eos=True) for k in reqs)
    print(f"page-boundary session: {len(reqs) - flips}/{len(reqs)} requests identical to their one-prompt static batch")
    lm.close()
