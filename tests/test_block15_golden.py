"""Production-shape pins by the independent PyTorch restatement (tests/golden/make_golden.py block15 / codecfull; SURVEY.md 8c "full-size
single layer ... tiles with synthetic weights"): ONE Fish-1.5 block -- dim 1024, 16 query / 2 kv heads x 64, SwiGLU 4096 -- as a
one-slow-layer / one-fast-layer model, at cached lengths 1 / 130 / 600, and the Firefly vocoder at its real width on 4 frames.
Checked: the CPU oracle (here) and the HIP kernels (f32 handle: per-node GEMV / attention kernels; bf16 handle: MFMA prefill GEMMs + flash
attention, paged bf16 KV, decode kernels) against fixtures neither of them produced."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lm_block15.npz"))
CF = np.load(os.path.join(os.path.dirname(__file__), "golden", "codec_full.npz"))
CFG = json.loads(bytes(G["cfg_json"]).decode())
MODEL = {k: CFG[k] for k in ("dim", "n_layer", "n_fast_layer", "n_head", "n_local_heads", "head_dim", "intermediate_size", "num_codebooks",
                             "codebook_size", "vocab_size", "max_seq_len", "norm_eps", "rope_base")}
MODEL["tie_word_embeddings"] = 0
TOK = {k: CFG[k] for k in ("im_end_id", "pad_id", "semantic_start_id", "semantic_end_id", "has_semantic_end")}
SEED = int(G["seed"])
PROMPT = G["prompt"]
F32_TOL = 5e-5   # two f32 implementations, different summation orders, logits of magnitude ~2.5
BF16_TOL = 1e-2  # bf16 weights + bf16 K/V in the cache (the golden keeps K/V in f32): same tolerance as the oracle comparisons at this size


def _walk(forward_generate, forward_generate_fast, clear_slow, clear_fast, fast_emb, tag, tol):
    worst = 0.0
    for T in (1, 130, 600):
        clear_slow()
        lg, hd = forward_generate(PROMPT[:, :T], 0)
        lg2, hd2 = forward_generate(PROMPT[:, T:T + 1], T)
        for name, got in (("prefill_logits", lg), ("prefill_hidden", hd), ("decode_logits", lg2), ("decode_hidden", hd2)):
            exp = G[f"{tag}_T{T}_{name}"]
            err = float(np.abs(np.asarray(got).reshape(exp.shape) - exp).max())
            worst = max(worst, err)
            assert err < tol, (tag, T, name, err)
        clear_fast()
        x = G[f"{tag}_T{T}_decode_hidden"]  # teacher-forced on the fixture's hidden state: the fast block is pinned on its own
        for pos in range(4):
            fl = np.asarray(forward_generate_fast(x, pos)).reshape(-1)
            err = float(np.abs(fl - G[f"{tag}_T{T}_fast_logits"][pos]).max())
            worst = max(worst, err)
            assert err < tol, (tag, T, "fast", pos, err)
            x = fast_emb(int(PROMPT[1 + pos, T]))
    return worst


@pytest.mark.parametrize("tag,bf16", [("f32", False), ("bf16w", True)])
def test_oracle_block15_vs_independent_restatement(tag, bf16):
    o = orc.OracleLM(CFG).load_synthetic(SEED, bf16=bf16)
    o.set_kv_round_bf16(False)  # the fixture keeps K / V in f32
    worst = _walk(lambda t, p: tuple(a[0] for a in o.forward_generate(t, p)), lambda x, p: o.forward_generate_fast(x, p)[0], o.clear_slow, o.clear_fast,
                  lambda c: o.fast_embeddings()[c], tag, F32_TOL)
    print(f"oracle vs fixture ({tag}): max |diff| {worst:.2e}")


def test_oracle_full_width_vocoder_vs_independent_restatement():
    o = orc.OracleCodec(tiny=False).load_synthetic(int(CF["seed"]))
    pcm = o.decode(CF["codes"])
    rms = float(np.sqrt(np.mean((pcm - CF["pcm"]) ** 2)))
    print(f"oracle vocoder (full width, 4 frames) vs fixture: rms {rms:.2e}, max {np.abs(pcm - CF['pcm']).max():.2e} at signal rms {np.sqrt(np.mean(CF['pcm'] ** 2)):.3f}")
    assert pcm.shape == CF["pcm"].shape == (8192,) and rms < 2e-6 and np.abs(pcm - CF["pcm"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dtype,tol", [("f32", "f32", F32_TOL), ("bf16w", "bf16", BF16_TOL)])
def test_hip_block15_vs_independent_restatement(tag, dtype, tol):
    import fishrt
    lm = fishrt.DualARTransformer(MODEL, TOK, 0, dtype).load_synthetic(SEED)
    worst = _walk(lambda t, p: tuple(a[0] for a in lm.forward_generate(t, p)), lambda x, p: lm.forward_generate_fast(x, p)[0], lm.clear_slow_layer_caches,
                  lm.clear_fast_layer_caches, lambda c: lm.fast_embeddings([c])[0], tag, tol)
    print(f"HIP {dtype} vs fixture ({tag}): max |diff| {worst:.2e} (tolerance {tol:.0e})")
    lm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,rms_tol", [("f32", 2e-6), ("bf16x3", 2.5e-5), ("f16", 4e-5)])
def test_hip_full_width_vocoder_vs_independent_restatement(precision, rms_tol):
    import fishrt
    c = fishrt.FireflyCodec(0, precision=precision).load_synthetic(int(CF["seed"]))
    pcm = c.decode(np.ascontiguousarray(CF["codes"][None]))[0, 0]
    rms = float(np.sqrt(np.mean((pcm - CF["pcm"]) ** 2)))
    print(f"HIP vocoder {precision} (full width, 4 frames) vs fixture: rms {rms:.2e}")
    assert pcm.shape == (8192,) and rms < rms_tol
    c.close()
