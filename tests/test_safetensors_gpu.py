"""GPU: the safetensors loaders (fs_lm_load_safetensors / fs_codec_load_safetensors) bind tensors by the reference's names
(dual_ar.rs:125-156,219-223,415-419,466-511; codec/utils/mod.rs:28-40,84-96) and re-lay them for the kernels.
A checkpoint is written with the `safetensors` package from the oracle's synthetic weights (f32 and bf16 storage); a handle
loaded from the file must behave exactly like the one initialised with fs_*_load_synthetic."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import config as fcfg
from oracle import oracle as orc

SEED = 99


def _lm_tensors(cfg, bf16):
    D, I, V = cfg["dim"], cfg["intermediate_size"], cfg["vocab_size"]
    qkv = (cfg["n_head"] + 2 * cfg["n_local_heads"]) * cfg["head_dim"]
    t = {}

    def mat(name, shape):
        t[name] = orc.synth(name, int(np.prod(shape)), SEED, 0.0, 0.02, bf16).reshape(shape)

    def nrm(name):
        t[name] = orc.synth(name, D, SEED, 1.0, 0.1, bf16)

    mat("embeddings.weight", (V, D))
    mat("codebook_embeddings.weight", (cfg["codebook_size"] * cfg["num_codebooks"], D))
    for pre, n in (("layers.", cfg["n_layer"]), ("fast_layers.", cfg["n_fast_layer"])):
        for l in range(n):
            p = f"{pre}{l}."
            mat(p + "attention.wqkv.weight", (qkv, D)); mat(p + "attention.wo.weight", (D, D))
            mat(p + "feed_forward.w1.weight", (I, D)); mat(p + "feed_forward.w2.weight", (D, I)); mat(p + "feed_forward.w3.weight", (I, D))
            nrm(p + "ffn_norm.weight"); nrm(p + "attention_norm.weight")
    nrm("norm.weight"); mat("output.weight", (V, D)); mat("fast_embeddings.weight", (cfg["codebook_size"], D))
    nrm("fast_norm.weight"); mat("fast_output.weight", (cfg["codebook_size"], D))
    return t


def _save(tensors, path, as_bf16):
    from safetensors.numpy import save_file
    if not as_bf16:
        save_file({k: np.ascontiguousarray(v, np.float32) for k, v in tensors.items()}, path)
        return
    # bf16 storage: write the raw file by hand (numpy has no bfloat16): header JSON + u16 payload
    import json, struct
    hdr, blobs, off = {}, [], 0
    for k, v in tensors.items():
        u = (np.ascontiguousarray(v, np.float32).view(np.uint32) >> 16).astype(np.uint16)  # values are already bf16-representable
        b = u.tobytes()
        hdr[k] = {"dtype": "BF16", "shape": list(v.shape), "data_offsets": [off, off + len(b)]}
        off += len(b); blobs.append(b)
    hdr["__metadata__"] = {"format": "pt"}
    hj = json.dumps(hdr).encode()
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj))); f.write(hj)
        for b in blobs:
            f.write(b)


@pytest.mark.parametrize("dtype,store_bf16", [("f32", False), ("bf16", True), ("bf16", False)])
def test_lm_checkpoint_equals_synthetic(tmp_path, dtype, store_bf16):
    cfg, tok = fcfg.TINY, fcfg.TINY_TOKENS
    path = str(tmp_path / "model.safetensors")
    _save(_lm_tensors(cfg, bf16=(dtype == "bf16")), path, store_bf16)
    a = fishrt.DualARTransformer(cfg, tok, 0, dtype).load_safetensors(path)
    b = fishrt.DualARTransformer(cfg, tok, 0, dtype).load_synthetic(SEED)
    p = np.zeros((9, 7), np.uint32)
    p[0] = [3, 401, 17, 464, 399, 12, 250]
    p[1:, 1] = np.arange(8); p[1:, 3] = 63 - np.arange(8)
    la, ha = a.forward_generate(p, 0)
    lb, hb = b.forward_generate(p, 0)
    assert np.array_equal(la, lb) and np.array_equal(ha, hb)
    ga = a.generate_blocking(p[:, :3], 20, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    gb = b.generate_blocking(p[:, :3], 20, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    assert np.array_equal(ga, gb)
    # missing tensor / wrong shape -> error naming the tensor (candle VarBuilder behaviour)
    bad = _lm_tensors(cfg, False)
    del bad["layers.1.feed_forward.w3.weight"]
    _save(bad, path, False)
    with pytest.raises(RuntimeError, match="layers.1.feed_forward.w3.weight"):
        fishrt.DualARTransformer(cfg, tok, 0, dtype).load_safetensors(path)


def test_malformed_checkpoints_are_errors_not_crashes(tmp_path):
    """header / offset validation of csrc/safetensors.h: a truncated or inconsistent file must come back as an error string (the
    reference's safetensors crate rejects the same files), never as a read past the mapping; a transposed tensor of the right element
    count is a shape error as with candle's VarBuilder::get"""
    import json, struct
    cfg, tok = fcfg.TINY, fcfg.TINY_TOKENS
    good = _lm_tensors(cfg, False)
    path = str(tmp_path / "bad.safetensors")

    def write(hdr_bytes, payload, hlen=None):
        with open(path, "wb") as f:
            f.write(struct.pack("<Q", len(hdr_bytes) if hlen is None else hlen)); f.write(hdr_bytes); f.write(payload)

    def header_and_payload(tensors, patch=None):
        hdr, blobs, off = {}, [], 0
        for k, v in tensors.items():
            b = np.ascontiguousarray(v, np.float32).tobytes()
            hdr[k] = {"dtype": "F32", "shape": list(v.shape), "data_offsets": [off, off + len(b)]}
            off += len(b); blobs.append(b)
        if patch:
            patch(hdr)
        return json.dumps(hdr).encode(), b"".join(blobs)

    def expect_error(match):
        with pytest.raises(RuntimeError, match=match):
            fishrt.DualARTransformer(cfg, tok, 0, "f32").load_safetensors(path)

    # 1. transposed matrix: same numel, wrong shape
    t = dict(good); t["layers.0.feed_forward.w2.weight"] = np.ascontiguousarray(t["layers.0.feed_forward.w2.weight"].T)
    write(*header_and_payload(t)); expect_error("shape mismatch for layers.0.feed_forward.w2.weight")
    # 2. data_offsets [0, 0] with a full shape (would read past the mapping)
    write(*header_and_payload(good, lambda h: h["norm.weight"].update(data_offsets=[0, 0]))); expect_error("norm.weight")
    # 3. byte range shorter than shape x dtype
    def short(h):
        b, e = h["fast_norm.weight"]["data_offsets"]; h["fast_norm.weight"]["data_offsets"] = [b, e - 4]
    write(*header_and_payload(good, short)); expect_error("fast_norm.weight")
    # 4. offsets beyond the file
    hj, pl = header_and_payload(good)
    write(hj, pl[: len(pl) // 2]); expect_error("bad offsets")
    # 5. header length larger than the file / absurd (8 + hlen wraps)
    write(hj, pl, hlen=len(hj) + len(pl) + 100); expect_error("corrupt safetensors header")
    write(hj, pl, hlen=2 ** 64 - 4); expect_error("corrupt safetensors header")
    # 6. truncated header JSON
    write(hj[: len(hj) // 2], b""); expect_error("safetensors")           # the first tensor's offsets already fail (no payload)
    write(hj[: len(hj) // 2], pl); expect_error("safetensors header")      # with the payload present the scanner runs off the header
    # 7. unsupported dtype
    write(*header_and_payload(good, lambda h: h["norm.weight"].update(dtype="I64"))); expect_error("unsupported safetensors dtype")
    # the handle machinery still works afterwards
    write(*header_and_payload(good))
    fishrt.DualARTransformer(cfg, tok, 0, "f32").load_safetensors(path).close()


def _codec_tensors(C=64, seed=1234):
    """the tiny codec's tensors by the reference's names (channel_div 8), from the oracle's synthetic generator"""
    import math
    t = {}

    def put(name, shape, mean, std):
        t[name] = orc.synth(name, int(np.prod(shape)), seed, mean, std).reshape(shape)

    for g in range(8):
        put(f"quantizer.residual_fsq.rvqs.{g}.project_out.weight", (C // 8, 4), 0.0, 0.5)
        put(f"quantizer.residual_fsq.rvqs.{g}.project_out.bias", (C // 8,), 0.0, 0.02)
    for i in range(2):
        p = f"quantizer.upsample.{i}"
        put(p + ".0.conv.weight", (C, C, 2), 0.0, 1 / math.sqrt(C)); put(p + ".0.conv.bias", (C,), 0.0, 0.02)
        q = p + ".1"
        put(q + ".dwconv.conv.weight", (C, 1, 7), 0.0, 1 / math.sqrt(7)); put(q + ".dwconv.conv.bias", (C,), 0.0, 0.02)
        put(q + ".norm.weight", (C,), 1.0, 0.1); put(q + ".norm.bias", (C,), 0.0, 0.02)
        put(q + ".pwconv1.weight", (4 * C, C), 0.0, 1 / math.sqrt(C)); put(q + ".pwconv1.bias", (4 * C,), 0.0, 0.02)
        put(q + ".pwconv2.weight", (C, 4 * C), 0.0, 1 / math.sqrt(4 * C)); put(q + ".pwconv2.bias", (C,), 0.0, 0.02)
        put(q + ".gamma", (C,), 0.1, 0.02)
    put("head.conv_pre.conv.weight", (C, C, 13), 0.0, 1 / math.sqrt(C * 13)); put("head.conv_pre.conv.bias", (C,), 0.0, 0.02)
    rates, ks = [8, 8, 2, 2, 2], [16, 16, 4, 4, 4]
    for s in range(5):
        cin, cout = C >> s, C >> (s + 1)
        put(f"head.ups.{s}.conv.weight", (cin, cout, ks[s]), 0.0, 1 / math.sqrt(cin * ks[s] / rates[s])); put(f"head.ups.{s}.conv.bias", (cout,), 0.0, 0.02)
        for j, k in enumerate((3, 7, 11)):
            for m in range(3):
                for cv in ("convs1", "convs2"):
                    q = f"head.resblocks.{s}.blocks.{j}.{cv}.{m}"
                    put(q + ".conv.weight", (cout, cout, k), 0.0, 1 / math.sqrt(cout * k)); put(q + ".conv.bias", (cout,), 0.0, 0.02)
    put("head.conv_post.conv.weight", (1, C >> 5, 13), 0.0, 1 / math.sqrt((C >> 5) * 13)); put("head.conv_post.conv.bias", (1,), 0.0, 0.02)

    # encoder side (convnext.rs:186-271, quantizer.rs:44-66, grouped_residual_fsq.rs:52-56); channel_div 8 -> dims / 8, depths 1,1,2,1
    def block(q, Cb):
        put(q + ".dwconv.conv.weight", (Cb, 1, 7), 0.0, 1 / math.sqrt(7)); put(q + ".dwconv.conv.bias", (Cb,), 0.0, 0.02)
        put(q + ".norm.weight", (Cb,), 1.0, 0.1); put(q + ".norm.bias", (Cb,), 0.0, 0.02)
        put(q + ".pwconv1.weight", (4 * Cb, Cb), 0.0, 1 / math.sqrt(Cb)); put(q + ".pwconv1.bias", (4 * Cb,), 0.0, 0.02)
        put(q + ".pwconv2.weight", (Cb, 4 * Cb), 0.0, 1 / math.sqrt(4 * Cb)); put(q + ".pwconv2.bias", (Cb,), 0.0, 0.02)
        put(q + ".gamma", (Cb,), 0.1, 0.02)

    dims, depths = [16, 32, 48, 64], [1, 1, 2, 1]
    put("backbone.downsample_layers.0.0.conv.weight", (dims[0], 160, 7), 0.0, 1 / math.sqrt(160 * 7)); put("backbone.downsample_layers.0.0.conv.bias", (dims[0],), 0.0, 0.02)
    put("backbone.downsample_layers.0.1.weight", (dims[0],), 1.0, 0.1); put("backbone.downsample_layers.0.1.bias", (dims[0],), 0.0, 0.02)
    for i in range(4):
        if i > 0:
            q = f"backbone.downsample_layers.{i}"
            put(q + ".0.weight", (dims[i - 1],), 1.0, 0.1); put(q + ".0.bias", (dims[i - 1],), 0.0, 0.02)
            put(q + ".1.weight", (dims[i], dims[i - 1], 1), 0.0, 1 / math.sqrt(dims[i - 1])); put(q + ".1.bias", (dims[i],), 0.0, 0.02)
        for j in range(depths[i]):
            block(f"backbone.stages.{i}.{j}", dims[i])
    put("backbone.norm.weight", (dims[3],), 1.0, 0.1); put("backbone.norm.bias", (dims[3],), 0.0, 0.02)
    for i in range(2):
        q = f"quantizer.downsample.{i}"
        put(q + ".0.conv.weight", (C, C, 2), 0.0, 1 / math.sqrt(C * 2)); put(q + ".0.conv.bias", (C,), 0.0, 0.02)
        block(q + ".1", C)
    for g in range(8):
        put(f"quantizer.residual_fsq.rvqs.{g}.project_in.weight", (4, C // 8), 0.0, 1 / math.sqrt(C // 8))
        put(f"quantizer.residual_fsq.rvqs.{g}.project_in.bias", (4,), 0.0, 0.02)
    return t


def test_codec_checkpoint_equals_synthetic(tmp_path):
    seed = 1234
    t = _codec_tensors(64, seed)
    path = str(tmp_path / "firefly.safetensors")
    _save(t, path, False)
    a = fishrt.FireflyCodec(0, channel_div=8).load_safetensors(path)
    b = fishrt.FireflyCodec(0, channel_div=8).load_synthetic(seed)
    codes = np.random.RandomState(0).randint(0, 1000, (1, 8, 9)).astype(np.uint32)
    assert np.array_equal(a.decode(codes), b.decode(codes))
    o = orc.OracleCodec(tiny=True).load_synthetic(seed)
    ref = o.decode(codes[0])
    assert float(np.sqrt(np.mean((a.decode(codes)[0, 0] - ref) ** 2))) < 4e-5  # default precision mode (f16 operands)
    a32 = fishrt.FireflyCodec(0, channel_div=8, precision="f32").load_safetensors(path)
    assert float(np.sqrt(np.mean((a32.decode(codes)[0, 0] - ref) ** 2))) < 1e-6
    clip = (0.2 * np.sin(np.arange(20000) * 0.05) + 0.02 * np.random.RandomState(1).randn(20000)).astype(np.float32)[None, None]
    assert np.array_equal(a.encode(clip), b.encode(clip))
