"""Model hyper-parameters.  The reference reads them from the checkpoint's config.json at run time
(fish_speech_core/src/bin/llama_generate.rs:176; BaseModelArgs::from_file, dual_ar.rs:110); the Fish-1.5 values
below are the public checkpoint's and are only defaults for synthetic-weight runs."""
import json

FISH_1_5 = dict(dim=1024, n_layer=24, n_fast_layer=4, n_head=16, n_local_heads=2, head_dim=64,
                intermediate_size=4096, num_codebooks=8, codebook_size=1024, vocab_size=102048, max_seq_len=8192,
                norm_eps=1e-6, rope_base=1e6, tie_word_embeddings=0)
FISH_1_5_TOKENS = dict(im_end_id=100011, pad_id=5, semantic_start_id=100012, semantic_end_id=101035, has_semantic_end=1)

# tiny configuration used by the parity tests: im_end == semantic_start - 1 preserved
TINY = dict(dim=128, n_layer=2, n_fast_layer=1, n_head=4, n_local_heads=2, head_dim=32, intermediate_size=256,
            num_codebooks=8, codebook_size=64, vocab_size=512, max_seq_len=256, norm_eps=1e-6, rope_base=1e6,
            tie_word_embeddings=0)
TINY_TOKENS = dict(im_end_id=400, pad_id=5, semantic_start_id=401, semantic_end_id=464, has_semantic_end=1)

FRAME_RATE_HZ = 21.535  # generate/single_batch.rs:292-295


def from_config_json(path):
    """BaseModelArgs::from_file (dual_ar.rs:110-115)."""
    with open(path) as f:
        c = json.load(f)
    out = dict(FISH_1_5)
    for k in out:
        if k in c and c[k] is not None:
            out[k] = c[k]
    if c.get("intermediate_size") is None:
        out["intermediate_size"] = out["dim"] * 4  # dual_ar.rs:129
    out["tie_word_embeddings"] = int(bool(c.get("tie_word_embeddings", False)))
    return out

# Fish-Speech 1.4 (BASELINE configs[4] shapes): same backbone, 32k vocabulary, single <|semantic|> id, no semantic range
# (TokenConfig: semantic_end_id = None, pad_id == semantic id; dual_ar.rs:34-46)
FISH_1_4 = dict(FISH_1_5, vocab_size=32000, max_seq_len=8192)  # RoPE table 8192 for the 4096-frame long-form run
FISH_1_4_TOKENS = dict(im_end_id=4, pad_id=5, semantic_start_id=5, semantic_end_id=0, has_semantic_end=0)
TINY_1_4_TOKENS = dict(im_end_id=4, pad_id=5, semantic_start_id=5, semantic_end_id=0, has_semantic_end=0)
