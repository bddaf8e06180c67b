// TEST INFRASTRUCTURE ONLY (oracle/).  CPU f32 restatement of the reference's dual-AR
// transformer hot path.  PARITY UNPINNED against the reference binary: the reference is Rust +
// candle 0.8.3 (Cargo.lock:365), neither of which can be built or imported in this image, and
// the reference holds no golden vectors for this path.  The restatement is
// pinned instead by (a) the weight-free known answers of SURVEY.md §8c (tests/test_oracle_known_answers.py)
// and (b) an independent PyTorch restatement written from the same reference files
// (tests/golden/make_golden.py -> tests/golden/*.npz).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
#pragma once
#include <cstdint>
#include <deque>
#include <set>
#include <string>
#include <vector>

namespace oracle {

// fish_speech_core/lib/lm/dual_ar.rs:57-81 (BaseModelArgs)
struct ModelArgs {
    int dim = 1024, n_layer = 24, n_fast_layer = 4, n_head = 16, n_local_heads = 2, head_dim = 64;
    int intermediate_size = 4096, num_codebooks = 8, codebook_size = 1024, vocab_size = 102048;
    int max_seq_len = 8192;
    float norm_eps = 1e-6f, rope_base = 1e6f;
    int tie_word_embeddings = 0;
};
// fish_speech_core/lib/lm/dual_ar.rs:17-23 (TokenConfig)
struct TokenCfg {
    uint32_t im_end_id = 100011, pad_id = 5, semantic_start_id = 100012, semantic_end_id = 101035;
    int has_semantic_end = 1;  // Fish 1.5 => Some(end); Fish 1.4 => None (dual_ar.rs:40-45)
};
// fish_speech_core/lib/lm/sampling/mod.rs:29-34
struct Sampling {
    double temp = 0.0, top_p = 1.0;
    uint64_t top_k = 0;
    float repetition_penalty = 1.0f;
};

struct Block {
    std::vector<float> wqkv, wo, w1, w2, w3, ffn_norm, attention_norm;
    // growing KV cache (dual_ar.rs:204,316-324), layout (B, Hkv, T, D)
    std::vector<float> k, v;
    int kv_len = 0, kv_b = 0;
    // test hook (orc_lm_force_kv): one-shot replacement of the NEXT single-token step's own K / V row by rows computed elsewhere (the GPU
    // kernel's bf16 entries); force_diff = largest |own - forced| of that step in units of the forced value's bf16 ulp
    std::vector<float> force_k, force_v;
    float force_diff = 0.f;
};

// fish_speech_core/lib/lm/sampling/rep_pen.rs:4-72 (SingleBatchedRepPenProcessor)
struct RepPen {
    std::vector<float> mask;
    std::deque<size_t> context;
    std::set<size_t> seen;  // tokens_seen: the count is `or_insert(1)` and never incremented (rep_pen.rs:43)
    size_t max_ctx = 16;
    float amt = 1.0f;
    void init(size_t vocab, size_t ctx, float penalty) {
        mask.assign(vocab, 1.0f); context.clear(); seen.clear(); max_ctx = ctx; amt = penalty;
    }
    // returns logits / mask after the window update (rep_pen.rs:37-65)
    void apply(std::vector<float>& logits, size_t last_token);
};

struct LM {
    ModelArgs a;
    TokenCfg t;
    bool kv_round_bf16 = false;  // mimic a bf16 KV cache (GPU bf16 mode comparison only)
    int n_threads = 0;
    std::vector<float> embeddings, codebook_embeddings, fast_embeddings, output, fast_output, norm, fast_norm;
    std::vector<Block> layers, fast_layers;
    std::vector<float> cos_t, sin_t;  // (max_seq_len, head_dim/2)  dual_ar.rs:168-186

    void init(const ModelArgs& args, const TokenCfg& tc);
    void load_synthetic(uint64_t seed, int mode);  // 0 f32, 1 bf16 checkpoint, 2 fp8 Linear weights (+ bf16 rest)
    // dual_ar.rs:574-635.  toks: (B, C+1, L) u32.  logits: (B, V) ; hidden: (B, dim) pre-norm.
    void forward_generate(const uint32_t* toks, int B, int L, int input_pos, float* logits, float* hidden,
                          bool full_vocab_head = true);
    // dual_ar.rs:638-673.  x: (B, dim).  logits: (B, codebook_size)
    void forward_generate_fast(const float* x, int B, int input_pos, float* logits);
    void clear_fast();                // dual_ar.rs:675-679
    void clear_slow();                // dual_ar.rs:681-685
    void clear_slow_until(int pos);   // dual_ar.rs:687-693, 392-404
    int kv_len() const { return layers.empty() ? 0 : layers[0].kv_len; }  // dual_ar.rs:695-700
    // generate/single_batch.rs:217-306 (generate_blocking): returns codes (num_codebooks, n) row-major.
    // `ignore_eos` is a bench-only extension (SURVEY.md §8d configs[0]): masks <|im_end|> so length is fixed.
    std::vector<uint32_t> generate(const uint32_t* prompt, int L, int max_new_tokens, const Sampling& s,
                                   uint64_t seed, bool ignore_eos, int* n_frames,
                                   std::vector<float>* hidden_out = nullptr, double* prefill_s = nullptr,
                                   double* decode_s = nullptr, int max_frames = -1,
                                   std::vector<float>* margins = nullptr);

    // generate/static_batch.rs:282-390 (generate_static_batch, audio_only): per-row codes (num_codebooks, n_b)
    std::vector<std::vector<uint32_t>> generate_batch(const std::vector<std::vector<uint32_t>>& prompts, const std::vector<int>& lens,
                                                      int max_new_tokens, const Sampling& s, uint64_t seed, bool ignore_eos,
                                                      std::vector<int>* n_frames,
                                                      std::vector<float>* margins = nullptr /* [iteration][row]: min top-2 margin of the row's 9 decisions */);

    void embed(const uint32_t* toks, int B, int L, float* x);  // dual_ar.rs:532-567
    void block_forward(Block& blk, float* x, int B, int L, int input_pos, int T_cached_expected);
};

void get_mask_abs(int size1, int size2, int context, uint8_t* mask);  // dual_ar.rs:702-712
void precompute_freqs(const ModelArgs& a, std::vector<float>& cos_t, std::vector<float>& sin_t);

// candle_transformers::generation::LogitsProcessor (candle-transformers 0.8.3, not in tree; call sites
// generate/single_batch.rs:38-46,133,169) and its vendored twin sampling/mod.rs:40-132.
struct ChaCha12Rng;  // rand 0.8.5 StdRng
struct LogitsProcessor {
    Sampling s;
    ChaCha12Rng* rng = nullptr;
    LogitsProcessor(uint64_t seed, const Sampling& s);
    ~LogitsProcessor();
    uint32_t sample(const float* logits, size_t n);
};

// test hooks into the sampler's RNG chain (rand 0.8.5 StdRng = ChaCha12, seed_from_u64, BlockRng::next_u64, WeightedIndex<f32>)
void rng_chacha12_block(const uint32_t* key8, uint64_t counter, uint32_t* out16);
void rng_seed_key(uint64_t seed, uint32_t* key8);
void rng_stream(uint64_t seed, int n_u32_first, uint32_t* out32, int n_u64, uint64_t* out64);
void rng_weighted_index(uint64_t seed, const float* w, int n, int draws, uint32_t* out);
void rng_batched_sample(uint64_t seed, const Sampling& s, const float* logits, int B, int n, int call_index, uint32_t* out);

}  // namespace oracle
