#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run IN THE BUILD CONTAINER ONLY (needs torch CPU; nothing here reads /root/reference).

This is an INDEPENDENT PyTorch (CPU, f32) restatement of the reference hot path, written tensor-op by tensor-op
from the reference sources (cited per function) and deliberately structured differently from oracle/*.cpp
(torch matmul / softmax / F.conv1d / F.conv_transpose1d / F.layer_norm instead of hand-written loops), so that
an agreement between the two pins the C++ restatement.  The reference itself (Rust + candle 0.8.3) cannot be
executed here, so these are "second-implementation" goldens, not reference outputs (DESIGN.md "Oracle").

Fixtures are data only: seeded inputs + expected outputs at a tiny configuration, plus weight-free known answers.

NOT made by this script: tests/golden/default_voice_codes.npy.  It is a byte copy of the reference's voice fixture
`voices-template/default.npy` ((8, 274) int64 code indices in [3, 999]: a DATA file the reference ships, not source), taken once in the
build container.  The tests use it as a realistic code sequence for the vocoder and the prompt formats; bench.py builds the VQ span of the
configs[1] "default voice" prompt from it -- it is the one piece of reference-held data that travels to the GPU box.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1
torch.set_num_threads(4)
torch.manual_seed(0)


# ----------------------------------------------------------------------------- synthetic weights (spec: oracle/fsgen.h)
def fnv1a64(name):
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & M64
    return h


def synth(name, shape, seed, mean=0.0, std=0.02, bf16=False):
    n = int(np.prod(shape))
    key = np.uint64(fnv1a64(name) ^ seed)
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = key + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    f = np.uint64(0xFFFF)
    s = ((z & f) + ((z >> np.uint64(16)) & f) + ((z >> np.uint64(32)) & f) + (z >> np.uint64(48))).astype(np.int64) - 131070
    v = s.astype(np.float32) * np.float32(std / 37837.2272)
    v = np.float32(mean) + v
    if bf16:
        u = v.view(np.uint32)
        r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
        v = r.view(np.float32)
    return torch.from_numpy(v.reshape(shape).copy())


# ----------------------------------------------------------------------------- dual-AR LM (dual_ar.rs)
TINY = dict(dim=128, n_layer=2, n_fast_layer=1, n_head=4, n_local_heads=2, head_dim=32,
            intermediate_size=256, num_codebooks=8, codebook_size=64, vocab_size=512, max_seq_len=256,
            norm_eps=1e-6, rope_base=1e6,
            im_end_id=400, pad_id=5, semantic_start_id=401, semantic_end_id=464, has_semantic_end=1)


class Block:
    def __init__(self, cfg, prefix, seed, bf16):
        D, I = cfg["dim"], cfg["intermediate_size"]
        qkv = (cfg["n_head"] + 2 * cfg["n_local_heads"]) * cfg["head_dim"]
        g = lambda n, shp: synth(prefix + n, shp, seed, 0.0, 0.02, bf16)
        o = lambda n, shp: synth(prefix + n, shp, seed, 1.0, 0.1, bf16)
        self.wqkv = g("attention.wqkv.weight", (qkv, D))
        self.wo = g("attention.wo.weight", (D, D))
        self.w1 = g("feed_forward.w1.weight", (I, D))
        self.w2 = g("feed_forward.w2.weight", (D, I))
        self.w3 = g("feed_forward.w3.weight", (I, D))
        self.ffn_norm = o("ffn_norm.weight", (D,))
        self.attention_norm = o("attention_norm.weight", (D,))
        self.kv = None


def rms_norm(x, w, eps):  # candle_nn::RmsNorm
    return x / torch.sqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def rope_i(x, cos, sin):  # candle_nn::rotary_emb::rope_i -- interleaved pairs; x: (B,H,L,D), cos/sin: (L, D/2)
    x0, x1 = x[..., 0::2], x[..., 1::2]
    o0 = x0 * cos - x1 * sin
    o1 = x0 * sin + x1 * cos
    return torch.stack([o0, o1], -1).flatten(-2)


class TorchLM:
    def __init__(self, cfg, seed, bf16=False):
        self.cfg = cfg
        D, V = cfg["dim"], cfg["vocab_size"]
        g = lambda n, shp: synth(n, shp, seed, 0.0, 0.02, bf16)
        o = lambda n, shp: synth(n, shp, seed, 1.0, 0.1, bf16)
        self.embeddings = g("embeddings.weight", (V, D))
        self.codebook_embeddings = g("codebook_embeddings.weight", (cfg["codebook_size"] * cfg["num_codebooks"], D))
        self.layers = [Block(cfg, f"layers.{l}.", seed, bf16) for l in range(cfg["n_layer"])]
        self.norm = o("norm.weight", (D,))
        self.output = g("output.weight", (V, D))
        self.fast_embeddings = g("fast_embeddings.weight", (cfg["codebook_size"], D))
        self.fast_layers = [Block(cfg, f"fast_layers.{l}.", seed, bf16) for l in range(cfg["n_fast_layer"])]
        self.fast_norm = o("fast_norm.weight", (D,))
        self.fast_output = g("fast_output.weight", (cfg["codebook_size"], D))
        # precompute_freqs_cis (dual_ar.rs:168-186)
        n_elem = D // cfg["n_head"]
        base = np.float32(cfg["rope_base"])
        theta = torch.tensor([np.float32(1.0) / np.power(base, np.float32(i) / np.float32(n_elem), dtype=np.float32)
                              for i in range(0, n_elem, 2)], dtype=torch.float32)
        idx = torch.arange(cfg["max_seq_len"], dtype=torch.float32)[:, None] * theta[None, :]
        self.cos, self.sin = idx.cos(), idx.sin()

    # dual_ar.rs:532-567
    def embed(self, x):
        cfg = self.cfg
        sem = x[:, 0, :].long()
        codes = x[:, 1:, :].long()
        sem_e = self.embeddings[sem].unsqueeze(1)
        shift = torch.arange(0, cfg["num_codebooks"] * cfg["codebook_size"], cfg["codebook_size"]).view(1, -1, 1)
        cb_e = self.codebook_embeddings[codes + shift]
        if cfg["has_semantic_end"]:
            mask = (sem <= cfg["semantic_end_id"]) & (sem >= cfg["semantic_start_id"])
        else:
            mask = sem == cfg["semantic_start_id"]
        cb_e = cb_e * mask.unsqueeze(1).unsqueeze(-1).float()
        return torch.cat([sem_e, cb_e], 1).sum(1)

    # dual_ar.rs:702-712
    def mask_abs(self, s1, s2):
        ctx = self.cfg["max_seq_len"]
        i = torch.arange(s1).view(-1, 1)
        j = torch.arange(s2).view(1, -1)
        return (s1 + j > s2 + i) | (s1 + j + ctx < s2 + i)

    # dual_ar.rs:281-384 + 429-440
    def block(self, blk, x, pos):
        cfg = self.cfg
        B, L, D = x.shape
        H, Hk, Dh = cfg["n_head"], cfg["n_local_heads"], cfg["head_dim"]
        h = rms_norm(x, blk.attention_norm, cfg["norm_eps"])
        qkv = h @ blk.wqkv.t()
        q, k, v = qkv.split([H * Dh, Hk * Dh, Hk * Dh], -1)
        q = q.view(B, L, H, Dh).transpose(1, 2)
        k = k.view(B, L, Hk, Dh).transpose(1, 2)
        v = v.view(B, L, Hk, Dh).transpose(1, 2)
        cos, sin = self.cos[pos:pos + L], self.sin[pos:pos + L]
        q, k = rope_i(q, cos, sin), rope_i(k, cos, sin)
        if blk.kv is not None:
            k = torch.cat([blk.kv[0], k], 2)
            v = torch.cat([blk.kv[1], v], 2)
        blk.kv = (k, v)
        T = k.shape[2]
        n_rep = H // Hk
        ke = k.unsqueeze(2).expand(B, Hk, n_rep, T, Dh).reshape(B, H, T, Dh)
        ve = v.unsqueeze(2).expand(B, Hk, n_rep, T, Dh).reshape(B, H, T, Dh)
        att = q @ (ke.transpose(-1, -2) * (1.0 / math.sqrt(Dh)))
        if L > 1:
            att = att.masked_fill(self.mask_abs(L, T).view(1, 1, L, T), float("-inf"))
        att = torch.softmax(att, -1)
        y = (att @ ve).transpose(1, 2).reshape(B, L, D)
        x = x + y @ blk.wo.t()
        h = rms_norm(x, blk.ffn_norm, cfg["norm_eps"])
        return x + (F.silu(h @ blk.w1.t()) * (h @ blk.w3.t())) @ blk.w2.t()

    def forward_generate(self, toks, pos):
        x = self.embed(toks)
        per_layer = []
        for blk in self.layers:
            x = self.block(blk, x, pos)
            per_layer.append(x[:, -1, :].clone())
        last = x[:, -1:, :]
        logits = rms_norm(last, self.norm, self.cfg["norm_eps"]) @ self.output.t()
        return logits[:, 0], last[:, 0], per_layer

    def forward_generate_fast(self, x, pos):
        x = x.view(-1, 1, self.cfg["dim"])
        for blk in self.fast_layers:
            x = self.block(blk, x, pos)
        return (rms_norm(x, self.fast_norm, self.cfg["norm_eps"]) @ self.fast_output.t())[:, 0]

    def clear_fast(self):
        for b in self.fast_layers:
            b.kv = None

    def clear_slow(self):
        for b in self.layers:
            b.kv = None

    def kv_len(self):
        return 0 if self.layers[0].kv is None else self.layers[0].kv[0].shape[2]


class RepPen:  # rep_pen.rs:4-72, python dict/deque transliteration of the HashMap/VecDeque logic
    def __init__(self, vocab, ctx, amt):
        self.mask = torch.ones(vocab)
        self.ctx, self.amt = ctx, amt
        self.context, self.seen = [], {}

    def apply(self, logits, last):
        if last not in self.seen:
            self.seen[last] = 1
        if self.seen[last] == 1:
            self.mask[last] = self.amt
        self.context.insert(0, last)
        if len(self.context) > self.ctx:
            d = self.context.pop()
            if d in self.seen:
                self.seen[d] -= 1
                if self.seen[d] == 0:
                    del self.seen[d]
                    self.mask[d] = 1.0
        return logits / self.mask


def argmax_last(v):  # host ArgMax: iter().enumerate().max_by(total_cmp) -> last maximal index
    v = v.numpy()
    m = v.max()
    return int(np.nonzero(v == m)[0][-1])


class PySampler:
    """candle_transformers LogitsProcessor with Sampling::TopKThenTopP (call sites single_batch.rs:38-46,133,169; vendored twin
    sampling/mod.rs:51-132) on top of tests/golden/rng_ref.py (StdRng, WeightedIndex).  Conventions shared with oracle/ and the HIP
    sampler, both documented there: the top-k set is kept in ascending token order (`select_nth_unstable_by` leaves it unspecified)
    and the softmax denominator is accumulated in f64."""

    def __init__(self, seed, temp, top_p, top_k):
        import rng_ref
        self.R = rng_ref
        self.rng = rng_ref.StdRng(seed)
        self.temp, self.top_p, self.top_k = temp, np.float32(top_p), top_k

    def _multinomial(self, probs):
        return self.R.weighted_index_sample(self.rng, [float(v) for v in probs])

    def _topp(self, probs):
        order = sorted(range(len(probs)), key=lambda i: -float(probs[i]))  # stable: ties stay in index order (slice::sort_by)
        cumsum = np.float32(0.0)
        for i in order:
            if cumsum >= self.top_p:
                probs[i] = np.float32(0.0)
            cumsum = np.float32(cumsum + probs[i])
        return self._multinomial(probs)

    def sample(self, logits):
        import math
        lg = np.asarray(logits, np.float32)
        if self.temp == 0.0:
            return int(np.nonzero(lg == lg.max())[0][-1])
        z = lg * np.float32(1.0 / self.temp)
        mx = z.max()
        e = np.array([np.float32(math.exp(float(np.float32(v - mx)))) for v in z], np.float32)
        p = e / np.float32(e.astype(np.float64).sum())
        n = len(p)
        if self.top_k == 0 or self.top_k >= n:
            return self._topp(p.copy())
        order = sorted(range(n), key=lambda i: -float(p[i]))
        keep = sorted(order[: self.top_k])
        tk = np.array([p[i] for i in keep], np.float32)
        sum_p = np.float32(0.0)
        for v in tk:
            sum_p = np.float32(sum_p + v)
        j = self._multinomial(tk) if (self.top_p <= 0 or self.top_p >= sum_p) else self._topp(tk)
        return keep[j]


def generate(lm, prompt, max_new_tokens, rep_pen, ignore_eos=False, sampler=None):
    """single_batch.rs:31-214 + 217-306; greedy (temp == 0) unless a PySampler is given."""
    pick = (lambda v: sampler.sample(v.numpy())) if sampler is not None else argmax_last
    cfg = lm.cfg
    C = cfg["num_codebooks"]
    im_end = cfg["im_end_id"]
    rps = [RepPen(cfg["codebook_size"], 16, rep_pen) for _ in range(C)]
    input_pos = lm.kv_len()
    max_pos = max_new_tokens + lm.kv_len()
    cur = prompt.clone()
    prev = None
    frames, margins = [], []
    it = 0
    while cur is not None and input_pos <= max_pos:
        logits, hidden, _ = lm.forward_generate(cur.unsqueeze(0), input_pos)
        sl = logits[0, im_end:].clone()
        if ignore_eos:
            sl[0] = float("-inf")
        top2 = torch.topk(sl, 2).values
        margins.append(float(top2[0] - top2[1]))
        sem = pick(sl) + im_end
        cb = [sem]
        lm.clear_fast()
        x = hidden
        for ci in range(C):
            if sem == im_end:
                cb.append(0)
                continue
            fl = lm.forward_generate_fast(x, ci)[0]
            if prev is not None:
                fl = rps[ci].apply(fl, prev[ci + 1])
            top2 = torch.topk(fl, 2).values
            margins.append(float(top2[0] - top2[1]))
            a = pick(fl)
            if ci != C - 1:
                x = lm.fast_embeddings[a].view(1, -1)
            cb.append(a)
        input_pos += cur.shape[-1] if prev is None else 1
        prev = cb
        cur = None if sem == im_end else torch.tensor(cb, dtype=torch.int64).view(-1, 1)
        if it == 0 or cb[0] != im_end:
            frames.append(cb)
        it += 1
    out = np.array(frames, np.uint32).T[1:]  # drop row 0
    return out, np.array(frames, np.uint32), min(margins)


def make_lm_golden():
    cfg = TINY
    seed = 0xF15E5EED
    rng = np.random.RandomState(1234)
    L = 12
    prompt = np.zeros((9, L), np.int64)
    prompt[0] = rng.randint(0, cfg["im_end_id"], L)
    # a VQ span inside the prompt (prompt.rs:53-93): row0 = semantic_start + code0, rows 1.. = codes
    codes = rng.randint(0, cfg["codebook_size"], (8, 5))
    prompt[0, 4:9] = cfg["semantic_start_id"] + codes[0]
    prompt[1:, 4:9] = codes
    out = dict(prompt=prompt.astype(np.uint32), seed=np.uint64(seed))
    for bf16 in (False, True):
        tag = "bf16w" if bf16 else "f32w"
        lm = TorchLM(cfg, seed, bf16)
        # teacher-forced stages: prefill, one decode step, one batched (B=2) prefill, fast steps
        logits, hidden, per_layer = lm.forward_generate(torch.from_numpy(prompt).unsqueeze(0), 0)
        out[f"{tag}_prefill_logits"] = logits.numpy()
        out[f"{tag}_prefill_hidden"] = hidden.numpy()
        out[f"{tag}_prefill_layers"] = torch.stack(per_layer).numpy()
        step = np.array([[cfg["semantic_start_id"] + 7], [7], [1], [2], [3], [4], [5], [6], [63]], np.int64)
        l2, h2, _ = lm.forward_generate(torch.from_numpy(step).unsqueeze(0), L)
        out["decode_step_tokens"] = step.astype(np.uint32)
        out[f"{tag}_decode_logits"] = l2.numpy()
        out[f"{tag}_decode_hidden"] = h2.numpy()
        lm.clear_fast()
        fl0 = lm.forward_generate_fast(h2, 0)
        fl1 = lm.forward_generate_fast(lm.fast_embeddings[11].view(1, -1), 1)
        fl2 = lm.forward_generate_fast(lm.fast_embeddings[50].view(1, -1), 2)
        out[f"{tag}_fast_logits"] = torch.cat([fl0, fl1, fl2]).numpy()
        # chunked prefill with a cached prefix (dual_ar.rs:585-588: mask (L, T_cached + L))
        lm.clear_slow()
        lm.forward_generate(torch.from_numpy(prompt[:, :5]).unsqueeze(0), 0)
        l3, h3, _ = lm.forward_generate(torch.from_numpy(prompt[:, 5:]).unsqueeze(0), 5)
        out[f"{tag}_chunked_logits"] = l3.numpy()
        # free-running greedy rollouts, rep-pen 1.0 and 1.2, max_new_tokens chosen for 24 frames: M - L + 2 = 24
        for rp in (1.0, 1.2):
            lm.clear_slow()
            codes_out, frames, margin = generate(lm, torch.from_numpy(prompt), 24 + L - 2, rp, ignore_eos=True)
            assert codes_out.shape == (8, 24), codes_out.shape
            out[f"{tag}_rollout_rp{int(rp * 10)}"] = codes_out
            out[f"{tag}_rollout_rp{int(rp * 10)}_frames"] = frames
            out[f"{tag}_rollout_rp{int(rp * 10)}_min_margin"] = np.float32(margin)
            print(f"  LM {tag} rp={rp}: 24 frames, min top-2 margin {margin:.3e}")
        if not bf16:
            # sampled rollouts (temp / top-k / top-p / WeightedIndex / StdRng): the independent pure-Python RNG + sampler
            for name, kw in (("s1", dict(seed=42, temp=0.7, top_p=0.8, top_k=20)), ("s2", dict(seed=7, temp=1.0, top_p=0.95, top_k=50)),
                             ("s3", dict(seed=123456789, temp=0.7, top_p=1.0, top_k=16))):
                lm.clear_slow()
                codes_out, frames, _ = generate(lm, torch.from_numpy(prompt), 24 + L - 2, 1.2, ignore_eos=True, sampler=PySampler(**kw))
                assert codes_out.shape == (8, 24), codes_out.shape
                out[f"sampled_{name}_cfg"] = np.array([kw["seed"], kw["temp"], kw["top_p"], kw["top_k"]], np.float64)
                out[f"sampled_{name}_rollout"] = codes_out
                out[f"sampled_{name}_frames"] = frames
                print(f"  LM sampled rollout {name} {kw}: 24 frames")
        # batched prefill B=2 with left padding (static_batch.rs:68-111): pad mask is NOT applied (dual_ar.rs:589-615)
        lm.clear_slow()
        p2 = np.zeros((2, 9, L), np.int64)
        p2[0] = prompt
        p2[1, 0, :4] = cfg["im_end_id"]
        p2[1, :, 4:] = prompt[:, : L - 4]
        lb, hb, _ = lm.forward_generate(torch.from_numpy(p2), 0)
        out["batch2_prompt"] = p2.astype(np.uint32)
        out[f"{tag}_batch2_logits"] = lb.numpy()
        out[f"{tag}_batch2_hidden"] = hb.numpy()
    np.savez_compressed(os.path.join(HERE, "lm_tiny.npz"), **out)


# ----------------------------------------------------------------------------- Firefly vocoder (codec/*.rs), tiny = channels / 8
def conv_w(name, cout, cin_g, k, seed):
    w = synth(name + ".conv.weight", (cout, cin_g, k), seed, 0.0, 1.0 / math.sqrt(cin_g * k))
    b = synth(name + ".conv.bias", (cout,), seed, 0.0, 0.02)
    return w, b


def fish_conv(x, w, b, dil=1, groups=1):  # utils/mod.rs:53-62 (stride 1): left pad (k-1)*dil
    k = w.shape[-1]
    return F.conv1d(F.pad(x, ((k - 1) * dil, 0)), w, b, dilation=dil, groups=groups)


def fish_tconv(x, w, b, stride):  # utils/mod.rs:110-122
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pad = max(k - stride, 0)
    return y[..., : y.shape[-1] - pad]


def torch_codec_decode(codes, seed, C):
    """decoder.rs:37-68 for b=1.  codes: (8, T) int."""
    G, T = codes.shape
    dg = C // G
    levels = [8, 5, 5, 5]
    # implicit codebook (fsq.rs:137-159) in float arithmetic as the reference does
    idx = torch.arange(1000, dtype=torch.float32).unsqueeze(-1)
    basis = torch.tensor([1.0, 8.0, 40.0, 200.0])
    lv = torch.tensor([8.0, 5.0, 5.0, 5.0])
    nc = torch.floor(idx / basis)
    nc = nc - torch.floor(nc / lv) * lv
    hw = torch.floor(lv / 2)
    codebook = (nc - hw) / hw
    stages = []
    zs = []
    for g in range(G):
        p = f"quantizer.residual_fsq.rvqs.{g}.project_out"
        w = synth(p + ".weight", (dg, 4), seed, 0.0, 0.5)
        b = synth(p + ".bias", (dg,), seed, 0.0, 0.02)
        zs.append(codebook[torch.from_numpy(codes[g].astype(np.int64))] @ w.t() + b)
    z = torch.cat(zs, -1).t().unsqueeze(0)  # (1, C, T)
    stages.append(z[0].clone())
    for i in range(2):
        p = f"quantizer.upsample.{i}"
        w = synth(p + ".0.conv.weight", (C, C, 2), seed, 0.0, 1.0 / math.sqrt(C))
        b = synth(p + ".0.conv.bias", (C,), seed, 0.0, 0.02)
        z = fish_tconv(z, w, b, 2)
        q = p + ".1"
        dw, db = conv_w(q + ".dwconv", C, 1, 7, seed)
        h = fish_conv(z, dw, db, 1, C).permute(0, 2, 1)
        h = F.layer_norm(h, (C,), synth(q + ".norm.weight", (C,), seed, 1.0, 0.1), synth(q + ".norm.bias", (C,), seed, 0.0, 0.02), 1e-6)
        h = F.gelu(h @ synth(q + ".pwconv1.weight", (4 * C, C), seed, 0.0, 1.0 / math.sqrt(C)).t()
                   + synth(q + ".pwconv1.bias", (4 * C,), seed, 0.0, 0.02), approximate="tanh")
        h = h @ synth(q + ".pwconv2.weight", (C, 4 * C), seed, 0.0, 1.0 / math.sqrt(4.0 * C)).t() + synth(q + ".pwconv2.bias", (C,), seed, 0.0, 0.02)
        h = synth(q + ".gamma", (C,), seed, 0.1, 0.02) * h
        z = z + h.permute(0, 2, 1)
        stages.append(z[0].clone())
    w, b = conv_w("head.conv_pre", C, C, 13, seed)
    x = fish_conv(z, w, b)
    stages.append(x[0].clone())
    rates, ks = [8, 8, 2, 2, 2], [16, 16, 4, 4, 4]
    ch = C
    for i in range(5):
        cin, cout = ch, ch // 2
        w = synth(f"head.ups.{i}.conv.weight", (cin, cout, ks[i]), seed, 0.0, 1.0 / math.sqrt(cin * ks[i] / rates[i]))
        b = synth(f"head.ups.{i}.conv.bias", (cout,), seed, 0.0, 0.02)
        x = fish_tconv(F.silu(x), w, b, rates[i])
        ch = cout
        outs = []
        for j, k in enumerate([3, 7, 11]):
            r = x
            for m, d in enumerate([1, 3, 5]):
                q = f"head.resblocks.{i}.blocks.{j}"
                w1, b1 = conv_w(f"{q}.convs1.{m}", ch, ch, k, seed)
                w2, b2 = conv_w(f"{q}.convs2.{m}", ch, ch, k, seed)
                t = fish_conv(F.silu(r), w1, b1, d)
                t = fish_conv(F.silu(t), w2, b2, d)
                r = r + t
            outs.append(r)
        x = torch.stack(outs, 0).mean(0)
        stages.append(x[0].clone())
    w, b = conv_w("head.conv_post", 1, ch, 13, seed)
    pcm = torch.tanh(fish_conv(F.silu(x), w, b))
    return pcm[0, 0], stages


def make_codec_golden():
    seed = 0xC0DEC
    rng = np.random.RandomState(7)
    T = 6
    codes = rng.randint(0, 1000, (8, T)).astype(np.uint32)
    codes[0, 0], codes[1, 0] = 0, 999
    pcm, stages = torch_codec_decode(codes, seed, 64)
    out = dict(codes=codes, seed=np.uint64(seed), pcm=pcm.numpy())
    for i, s in enumerate(stages):
        out[f"stage{i}"] = s.numpy()
    print("  codec tiny: pcm", pcm.shape, "rms", float(pcm.pow(2).mean().sqrt()), "stage rms",
          [round(float(s.pow(2).mean().sqrt()), 3) for s in stages])
    np.savez_compressed(os.path.join(HERE, "codec_tiny.npz"), **out)



# ----------------------------------------------------------------------------- production-shape fixtures ("full-size single layer")
# One Fish-1.5 BLOCK: dim 1024, 16 query / 2 kv heads x 64, SwiGLU 4096 -- the geometry every production kernel is specialised for (head_dim
# 64 RoPE, GQA 16:2, K = 1024 / 4096 GEMVs) -- as a one-slow-layer / one-fast-layer model with a small vocabulary (the embedding / head tables
# would be 400 MB each at 102k tokens and pin nothing new).  Cached lengths 1, 130 and 600: one token, just past one 128-token attention chunk /
# two KV pages, several chunks.  Per length: prefill (causal mask) logits + hidden of the last position, one decode step on top (L == 1, no
# mask), and four fast-decoder passes (positions 0..3) fed with hidden state / fast embeddings.
B15 = dict(dim=1024, n_layer=1, n_fast_layer=1, n_head=16, n_local_heads=2, head_dim=64, intermediate_size=4096, num_codebooks=8,
           codebook_size=1024, vocab_size=4096, max_seq_len=1024, norm_eps=1e-6, rope_base=1e6,
           im_end_id=2000, pad_id=5, semantic_start_id=2001, semantic_end_id=3024, has_semantic_end=1)


def make_block15_golden():
    cfg, seed = B15, 0xF15E5EED
    import json
    out = dict(seed=np.uint64(seed), cfg_json=np.frombuffer(json.dumps(cfg).encode(), np.uint8))
    rng = np.random.RandomState(15)
    Tmax = 600
    prompt = np.zeros((9, Tmax + 1), np.int64)
    prompt[0] = rng.randint(0, cfg["im_end_id"], Tmax + 1)
    vq = rng.rand(Tmax + 1) < 0.5  # half of the positions are VQ frames: semantic token + 8 codebook rows
    prompt[0, vq] = cfg["semantic_start_id"] + rng.randint(0, 1024, int(vq.sum()))
    prompt[1:, vq] = rng.randint(0, 1024, (8, int(vq.sum())))
    out["prompt"] = prompt.astype(np.uint32)
    for tag, bf16 in (("f32", False), ("bf16w", True)):
        lm = TorchLM(cfg, seed, bf16=bf16)
        for T in (1, 130, 600):
            lm.clear_slow()
            lg, hd, _ = lm.forward_generate(torch.from_numpy(prompt[:, :T]).unsqueeze(0), 0)
            lg2, hd2, _ = lm.forward_generate(torch.from_numpy(prompt[:, T:T + 1]).unsqueeze(0), T)
            out[f"{tag}_T{T}_prefill_logits"], out[f"{tag}_T{T}_prefill_hidden"] = lg.numpy()[0], hd.numpy()[0]
            out[f"{tag}_T{T}_decode_logits"], out[f"{tag}_T{T}_decode_hidden"] = lg2.numpy()[0], hd2.numpy()[0]
            lm.clear_fast()
            x, fl = hd2, []
            for pos in range(4):
                f = lm.forward_generate_fast(x, pos)
                fl.append(f.numpy()[0])
                x = lm.fast_embeddings[int(prompt[1 + pos, T])].view(1, -1)  # a fixed code sequence, not the argmax: no tie sensitivity
            out[f"{tag}_T{T}_fast_logits"] = np.stack(fl)
            print(f"  block15 {tag} T={T}: |logits| {float(lg2.abs().max()):.3f}, |hidden| {float(hd2.abs().max()):.3f}, fast |logits| {np.abs(fl[-1]).max():.3f}")
    np.savez_compressed(os.path.join(HERE, "lm_block15.npz"), **out)


def make_codec_full_golden():
    """The vocoder at its REAL width (512 -> 256 -> ... -> 16 channels, every HiFi-GAN stage full size) on 4 frames: 8192 PCM samples."""
    seed = 0xC0DEC
    rng = np.random.RandomState(9)
    codes = rng.randint(0, 1000, (8, 4)).astype(np.uint32)
    pcm, stages = torch_codec_decode(codes, seed, 512)
    out = dict(codes=codes, seed=np.uint64(seed), pcm=pcm.numpy())
    for i in (0, len(stages) // 2, len(stages) - 1):  # a few stage outputs (first, middle, last before conv_post), 64-sample tails
        out[f"stage{i}_tail"] = stages[i][:, -64:].numpy()
    print("  codec full width: pcm", tuple(pcm.shape), "rms", float(pcm.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(HERE, "codec_full.npz"), **out)


# ----------------------------------------------------------------------------- Firefly encoder (tiny: dims / 8, depths 1,1,2,1)
REF_MEL = "/root/reference/fish_speech_core/lib/audio/melfilters160.bytes"


def torch_log_mel(pcm, mel_table):
    """spectrogram.rs:29-158 + stft.rs:52-90 with numpy's f64 FFT: reflect pad that REPEATS the edge, frame f = padded[512 f,
    512 f + 2048) (tail zero-filled), periodic Hann, |.| -> f32, + 1e-6, (frames, 1025) @ (1025, 160), clamp(1e-5, 100).log()"""
    pad = 768
    x = np.concatenate([pcm[:pad][::-1], pcm, pcm[-pad:][::-1]]).astype(np.float64)
    full, rem = divmod(len(x), 512)
    n_frames = max(0, full - 3) + (1 if rem > 0 and len(x) >= 2048 else 0)
    win = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(2048) / 2048.0))
    frames = np.stack([np.pad(x[f * 512:f * 512 + 2048], (0, max(0, f * 512 + 2048 - len(x)))) * win for f in range(n_frames)])
    lin = np.abs(np.fft.fft(frames, axis=1)[:, :1025]).astype(np.float32) + np.float32(1e-6)
    mel = torch.from_numpy(lin) @ torch.from_numpy(mel_table)
    return torch.log(torch.clamp(mel, 1e-5, 100.0)).t().contiguous()  # (160, frames)


def ln_cf(x, w, b):  # LayerNormChannelsFirst (convnext.rs:144-154); x (1, C, T)
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return (x - u) / torch.sqrt(s + 1e-6) * w[None, :, None] + b[None, :, None]


def cnx_block(z, q, C, seed):  # convnext.rs:110-126
    dw, db = conv_w(q + ".dwconv", C, 1, 7, seed)
    h = fish_conv(z, dw, db, 1, C).permute(0, 2, 1)
    h = F.layer_norm(h, (C,), synth(q + ".norm.weight", (C,), seed, 1.0, 0.1), synth(q + ".norm.bias", (C,), seed, 0.0, 0.02), 1e-6)
    h = F.gelu(h @ synth(q + ".pwconv1.weight", (4 * C, C), seed, 0.0, 1.0 / math.sqrt(C)).t()
               + synth(q + ".pwconv1.bias", (4 * C,), seed, 0.0, 0.02), approximate="tanh")
    h = h @ synth(q + ".pwconv2.weight", (C, 4 * C), seed, 0.0, 1.0 / math.sqrt(4.0 * C)).t() + synth(q + ".pwconv2.bias", (C,), seed, 0.0, 0.02)
    return z + (synth(q + ".gamma", (C,), seed, 0.1, 0.02) * h).permute(0, 2, 1)


def torch_codec_encode_mel(mel, seed, dims, depths):
    """encoder.rs:38-42: ConvNeXtEncoder (convnext.rs:319-331) -> quantizer.encode (quantizer.rs:104-124, FSQ fsq.rs:68-118)."""
    stages = []
    x = mel.unsqueeze(0)
    w, b = conv_w("backbone.downsample_layers.0.0", dims[0], 160, 7, seed)
    x = fish_conv(x, w, b)
    x = ln_cf(x, synth("backbone.downsample_layers.0.1.weight", (dims[0],), seed, 1.0, 0.1), synth("backbone.downsample_layers.0.1.bias", (dims[0],), seed, 0.0, 0.02))
    for j in range(depths[0]):
        x = cnx_block(x, f"backbone.stages.0.{j}", dims[0], seed)
    stages.append(x[0].clone())
    for i in range(1, 4):
        q = f"backbone.downsample_layers.{i}"
        x = ln_cf(x, synth(q + ".0.weight", (dims[i - 1],), seed, 1.0, 0.1), synth(q + ".0.bias", (dims[i - 1],), seed, 0.0, 0.02))
        x = F.conv1d(x, synth(q + ".1.weight", (dims[i], dims[i - 1], 1), seed, 0.0, 1.0 / math.sqrt(dims[i - 1])), synth(q + ".1.bias", (dims[i],), seed, 0.0, 0.02))
        for j in range(depths[i]):
            x = cnx_block(x, f"backbone.stages.{i}.{j}", dims[i], seed)
        stages.append(x[0].clone())
    x = ln_cf(x, synth("backbone.norm.weight", (dims[3],), seed, 1.0, 0.1), synth("backbone.norm.bias", (dims[3],), seed, 0.0, 0.02))
    stages.append(x[0].clone())
    C = dims[3]
    for i in range(2):
        q = f"quantizer.downsample.{i}"
        w, b = conv_w(q + ".0", C, C, 2, seed)
        x = F.conv1d(x, w, b, stride=2)  # FishConvNet: pad = k - stride = 0
        x = cnx_block(x, q + ".1", C, seed)
        stages.append(x[0].clone())
    zt = x[0].t()  # (L, C)
    G, dg = 8, C // 8
    lv = torch.tensor([8.0, 5.0, 5.0, 5.0])
    basis = torch.tensor([1.0, 8.0, 40.0, 200.0])
    half_l = (lv - 1.0) * 1.001 / 2.0
    offset = torch.where(lv % 2 == 0, torch.tensor(0.5), torch.tensor(0.0))
    qv = offset / half_l
    shift = torch.log((1.0 + qv) / (1.0 - qv)) * 0.5

    def bound(z):
        return torch.tanh(z + shift) * half_l - offset

    hw = torch.floor(lv / 2.0)
    out = []
    for g in range(G):
        p = f"quantizer.residual_fsq.rvqs.{g}.project_in"
        z = zt[:, g * dg:(g + 1) * dg] @ synth(p + ".weight", (4, dg), seed, 0.0, 1.0 / math.sqrt(dg)).t() + synth(p + ".bias", (4,), seed, 0.0, 0.02)
        residual = bound(z)
        codes = torch.round(bound(residual / 1.0)) / hw      # torch.round is half-to-even; exact .5 does not occur after tanh
        idx = ((codes * hw + hw) * basis).sum(-1).to(torch.int64)
        out.append(idx)
    return torch.stack(out, 0).numpy().astype(np.uint32), stages


def make_codec_enc_golden():
    seed = 0xC0DEC
    rng = np.random.RandomState(11)
    n = 512 * 30 + 123
    t = np.arange(n) / 44100.0
    pcm = (0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 1870.0 * t + 1.0) + 0.05 * rng.randn(n)).astype(np.float32)
    table = np.fromfile(REF_MEL, dtype="<f4").reshape(1025, 160)
    mel = torch_log_mel(pcm, table)
    codes, stages = torch_codec_encode_mel(mel, seed, [16, 32, 48, 64], [1, 1, 2, 1])
    out = dict(pcm=pcm, seed=np.uint64(seed), mel=mel.numpy(), codes=codes)
    for i, s in enumerate(stages):
        out[f"stage{i}"] = s.numpy()
    # a sample of the reference's embedded mel table (the build regenerates the table from the slaney formula)
    flat = table.reshape(-1)
    nz = np.nonzero(flat)[0]
    pick = np.concatenate([nz[rng.choice(len(nz), 400, replace=False)], rng.choice(flat.size, 112, replace=False)]).astype(np.int64)
    out["mel_table_idx"], out["mel_table_val"] = pick, flat[pick]
    print("  codec enc tiny: mel", tuple(mel.shape), "codes", codes.shape, "stage rms", [round(float(s.pow(2).mean().sqrt()), 3) for s in stages])
    np.savez_compressed(os.path.join(HERE, "codec_enc_tiny.npz"), **out)


def make_known_answers():
    """Weight-free known answers derived by hand from the reference sources."""
    out = {}
    # FSQ implicit codebook: idx 0 -> (-1,-1,-1,-1); idx 999 -> (0.75, 1, 1, 1)  (fsq.rs:53-58,119-144)
    cb = np.zeros((1000, 4), np.float32)
    for idx in range(1000):
        for k, (basis, lv) in enumerate(zip([1, 8, 40, 200], [8, 5, 5, 5])):
            hw = lv // 2
            cb[idx, k] = ((idx // basis) % lv - hw) / hw
    assert tuple(cb[0]) == (-1, -1, -1, -1) and tuple(cb[999]) == (0.75, 1, 1, 1)
    out["fsq_codebook"] = cb
    # causal mask (dual_ar.rs:702-712) for (3, 5), context 8192
    out["mask_3_5"] = np.array([[0, 0, 0, 1, 1], [0, 0, 0, 0, 1], [0, 0, 0, 0, 0]], np.uint8)
    # rep-pen quirk trace (rep_pen.rs:37-65), window 3, penalty 2.0, tokens 5,5,7,8,5:
    #  after 5: {5}; after 5: {5}; after 7: {5,7}; after 8 (window 5,5,7,8 -> drop oldest 5 -> 5 un-penalised though
    #  another 5 is still inside the window): {7,8}; after 5: 5 is re-inserted, window 5,8,7,5 -> the dropped token is
    #  the older 5, which IS in the map again (count 1 -> 0) -> removed: {7,8} (the token just seen is un-penalised)
    out["reppen_tokens"] = np.array([5, 5, 7, 8, 5], np.int32)
    out["reppen_penalised"] = np.array([[5, -1, -1], [5, -1, -1], [5, 7, -1], [7, 8, -1], [7, 8, -1]], np.int32)
    # RoPE table spot values (dual_ar.rs:174-185), f64 reference for tolerance checks
    out["rope_theta_fish15"] = np.array([1.0 / (1e6 ** (i / 64.0)) for i in range(0, 64, 2)], np.float64)
    # frame budget (single_batch.rs:61,77,193-197): frames = M - L - p0 + 2 when no EOS
    out["budget_cases"] = np.array([[16, 256, 0, 242], [12, 34, 0, 24], [1, 10, 0, 11]], np.int64)
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["known", "lm", "codec", "enc", "block15", "codecfull"]
    if "known" in which:
        make_known_answers()
    if "lm" in which:
        make_lm_golden()
    if "codec" in which:
        make_codec_golden()
    if "enc" in which:
        make_codec_enc_golden()  # needs /root/reference (mel table): build container only
    if "block15" in which:
        make_block15_golden()
    if "codecfull" in which:
        make_codec_full_golden()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
