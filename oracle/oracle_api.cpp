// TEST INFRASTRUCTURE ONLY (oracle/).  Flat C API over the CPU restatement, for ctypes (oracle/oracle.py).
#include <cstring>
#include <string>

#include "fsgen.h"
#include <stdexcept>
#include "oracle_codec.h"
#include "oracle_lm.h"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

static thread_local std::string g_err;
#define GUARD(stmt) \
    try { stmt; return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// ---- generator
void orc_synth_fill(float* dst, uint64_t n, const char* name, uint64_t seed, float mean, double stdv, int bf16) {
    fsgen::fill(dst, n, name, seed, mean, stdv, bf16 != 0);
}

// fp8 (OCP e4m3fn) helpers of the FS_FP8 storage format
uint8_t orc_f32_to_e4m3(float f) { return fsgen::f32_to_e4m3(f); }
float orc_e4m3_to_f32(uint8_t b) { return fsgen::e4m3_to_f32(b); }
void orc_quant_rows_fp8(float* w, uint64_t rows, uint64_t cols) { fsgen::quant_rows_fp8(w, rows, cols); }

// ---- LM
void* orc_lm_create(const int* iargs /*11*/, const float* fargs /*2*/, const uint32_t* tok /*5*/) {
    ModelArgs a;
    a.dim = iargs[0]; a.n_layer = iargs[1]; a.n_fast_layer = iargs[2]; a.n_head = iargs[3]; a.n_local_heads = iargs[4];
    a.head_dim = iargs[5]; a.intermediate_size = iargs[6]; a.num_codebooks = iargs[7]; a.codebook_size = iargs[8];
    a.vocab_size = iargs[9]; a.max_seq_len = iargs[10];
    a.norm_eps = fargs[0]; a.rope_base = fargs[1];
    TokenCfg t;
    t.im_end_id = tok[0]; t.pad_id = tok[1]; t.semantic_start_id = tok[2]; t.semantic_end_id = tok[3];
    t.has_semantic_end = (int)tok[4];
    LM* lm = new LM();
    lm->init(a, t);
    return lm;
}
void orc_lm_destroy(void* p) { delete (LM*)p; }
int orc_lm_load_synthetic(void* p, uint64_t seed, int mode) { GUARD(((LM*)p)->load_synthetic(seed, mode)) }
void orc_lm_set_kv_round_bf16(void* p, int on) { ((LM*)p)->kv_round_bf16 = on != 0; }
int orc_lm_forward_generate(void* p, const uint32_t* toks, int B, int L, int input_pos, float* logits, float* hidden,
                            int full_head) {
    GUARD(((LM*)p)->forward_generate(toks, B, L, input_pos, logits, hidden, full_head != 0))
}
int orc_lm_forward_generate_fast(void* p, const float* x, int B, int pos, float* logits) {
    GUARD(((LM*)p)->forward_generate_fast(x, B, pos, logits))
}
void orc_lm_clear_fast(void* p) { ((LM*)p)->clear_fast(); }
void orc_lm_clear_slow(void* p) { ((LM*)p)->clear_slow(); }
void orc_lm_clear_slow_until(void* p, int pos) { ((LM*)p)->clear_slow_until(pos); }
int orc_lm_kv_len(void* p) { return ((LM*)p)->kv_len(); }
// test hook: overwrite cached K / V rows [t0, t0 + n) of one slow layer (batch 1) with values computed elsewhere -- the parity tests feed the
// oracle the GPU's own (bf16) cache entries so that K/V rounding-boundary flips stop compounding through the layers and only the
// summation order of the CURRENT step separates the two (tests/test_kv_forced_gpu.py).  k, v: [n][Hkv][D]
int orc_lm_set_kv(void* p, int layer, int t0, int n, const float* k, const float* v) {
    LM* lm = (LM*)p;
    const bool fast = layer >= 1000;  // layer 1000 + l: fast decoder layer l (its per-frame cache, dual_ar.rs:638-673)
    if (fast) layer -= 1000;
    if (layer < 0 || layer >= (int)(fast ? lm->fast_layers.size() : lm->layers.size())) return 1;
    Block& b = fast ? lm->fast_layers[layer] : lm->layers[layer];
    const int Hk = lm->a.n_local_heads, D = lm->a.head_dim, T = b.kv_len;
    if (b.kv_b != 1 || t0 < 0 || n < 0 || t0 + n > T) return 1;
    for (int t = 0; t < n; ++t)
        for (int g = 0; g < Hk; ++g)
            for (int d = 0; d < D; ++d) {
                b.k[((size_t)g * T + t0 + t) * D + d] = k[((size_t)t * Hk + g) * D + d];
                b.v[((size_t)g * T + t0 + t) * D + d] = v[((size_t)t * Hk + g) * D + d];
            }
    return 0;
}
const float* orc_lm_fast_embeddings(void* p) { return ((LM*)p)->fast_embeddings.data(); }
const float* orc_lm_freqs(void* p, int sin) { return sin ? ((LM*)p)->sin_t.data() : ((LM*)p)->cos_t.data(); }
// test hook: the next batch-1 single-token step of this layer (1000 + l: fast layer l) uses these K / V rows [Hkv][D] instead of its own
int orc_lm_force_kv(void* p, int layer, const float* k, const float* v) {
    LM* lm = (LM*)p;
    const bool fast = layer >= 1000;
    if (fast) layer -= 1000;
    if (layer < 0 || layer >= (int)(fast ? lm->fast_layers.size() : lm->layers.size())) return 1;
    Block& b = fast ? lm->fast_layers[layer] : lm->layers[layer];
    const size_t n = (size_t)lm->a.n_local_heads * lm->a.head_dim;
    b.force_k.assign(k, k + n); b.force_v.assign(v, v + n);
    return 0;
}
float orc_lm_force_kv_diff(void* p, int layer) {
    LM* lm = (LM*)p;
    const bool fast = layer >= 1000;
    if (fast) layer -= 1000;
    return (fast ? lm->fast_layers.at(layer) : lm->layers.at(layer)).force_diff;
}
// tensor access for cross-checks
const float* orc_lm_tensor(void* p, const char* name, int layer) {
    LM* lm = (LM*)p;
    std::string n = name;
    if (n == "embeddings") return lm->embeddings.data();
    if (n == "codebook_embeddings") return lm->codebook_embeddings.data();
    if (n == "output") return lm->output.data();
    if (n == "fast_output") return lm->fast_output.data();
    if (n == "norm") return lm->norm.data();
    if (n == "fast_norm") return lm->fast_norm.data();
    bool fast = n.rfind("fast.", 0) == 0;
    if (fast) n = n.substr(5);
    Block& b = fast ? lm->fast_layers.at(layer) : lm->layers.at(layer);
    if (n == "wqkv") return b.wqkv.data();
    if (n == "wo") return b.wo.data();
    if (n == "w1") return b.w1.data();
    if (n == "w2") return b.w2.data();
    if (n == "w3") return b.w3.data();
    if (n == "ffn_norm") return b.ffn_norm.data();
    if (n == "attention_norm") return b.attention_norm.data();
    return nullptr;
}
// generate_blocking.  codes_out: (num_codebooks, cap) row-major with row stride = *n_frames after return
int orc_lm_generate(void* p, const uint32_t* prompt, int L, int max_new_tokens, double temp, double top_p, uint64_t top_k,
                    float rep_pen, uint64_t seed, int ignore_eos, int max_frames, uint32_t* codes_out, int cap,
                    int* n_frames, double* prefill_s, double* decode_s, float* margins_out /*[iterations] or null*/,
                    int* n_iter_out, float* hidden_out /*[hidden_cap, dim] or null*/, int hidden_cap, int* n_hidden) {
    try {
        LM* lm = (LM*)p;
        Sampling s; s.temp = temp; s.top_p = top_p; s.top_k = top_k; s.repetition_penalty = rep_pen;
        int n = 0;
        std::vector<float> margins, hid;
        auto out = lm->generate(prompt, L, max_new_tokens, s, seed, ignore_eos != 0, &n, hidden_out ? &hid : nullptr, prefill_s, decode_s,
                                max_frames, margins_out ? &margins : nullptr);
        if (hidden_out) {
            const int rows = (int)(hid.size() / (size_t)lm->a.dim);
            if (rows > hidden_cap) { g_err = "hidden_out too small"; return 2; }
            std::memcpy(hidden_out, hid.data(), sizeof(float) * hid.size());
            if (n_hidden) *n_hidden = rows;
        }
        if (margins_out) std::memcpy(margins_out, margins.data(), sizeof(float) * margins.size());
        if (n_iter_out) *n_iter_out = (int)margins.size();
        if (n > cap) { g_err = "codes_out too small"; return 2; }
        std::memcpy(codes_out, out.data(), sizeof(uint32_t) * out.size());
        *n_frames = n;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// generate_static_batch.  prompts concatenated [(C+1) x L_i]; codes_out: [n][C][cap] row-major; n_frames[n]
int orc_lm_generate_batch(void* p, const uint32_t* prompts, const int* lens, int n, int max_new_tokens, double temp, double top_p,
                          uint64_t top_k, uint64_t seed, int ignore_eos, uint32_t* codes_out, int cap, int* n_frames,
                          float* margins_out /*[margins_cap][n] or null*/, int margins_cap, int* n_iter) {
    try {
        LM* lm = (LM*)p;
        Sampling s; s.temp = temp; s.top_p = top_p; s.top_k = top_k; s.repetition_penalty = 1.0f;
        const int C1 = lm->a.num_codebooks + 1, C = lm->a.num_codebooks;
        std::vector<std::vector<uint32_t>> ps(n);
        std::vector<int> ls(lens, lens + n);
        size_t off = 0;
        for (int i = 0; i < n; ++i) { ps[i].assign(prompts + off, prompts + off + (size_t)C1 * lens[i]); off += (size_t)C1 * lens[i]; }
        std::vector<int> nf;
        std::vector<float> margins;
        auto out = lm->generate_batch(ps, ls, max_new_tokens, s, seed, ignore_eos != 0, &nf, margins_out ? &margins : nullptr);
        if (margins_out) {
            const int it = (int)(margins.size() / (size_t)n);
            if (it > margins_cap) { g_err = "margins_out too small"; return 2; }
            std::memcpy(margins_out, margins.data(), sizeof(float) * margins.size());
            if (n_iter) *n_iter = it;
        }
        for (int i = 0; i < n; ++i) {
            if (nf[i] > cap) { g_err = "codes_out too small"; return 2; }
            for (int c = 0; c < C; ++c)
                std::memcpy(codes_out + ((size_t)i * C + c) * cap, out[i].data() + (size_t)c * nf[i], sizeof(uint32_t) * nf[i]);
            n_frames[i] = nf[i];
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// ---- weight-free helpers (known-answer tests)
void orc_get_mask_abs(int s1, int s2, int ctx, uint8_t* m) { get_mask_abs(s1, s2, ctx, m); }
void* orc_reppen_create(int vocab, int ctx, float amt) { RepPen* r = new RepPen(); r->init(vocab, ctx, amt); return r; }
void orc_reppen_destroy(void* r) { delete (RepPen*)r; }
int orc_reppen_apply(void* r, float* logits, int n, int last_token) {
    try {
        std::vector<float> l(logits, logits + n);
        ((RepPen*)r)->apply(l, (size_t)last_token);
        std::memcpy(logits, l.data(), sizeof(float) * n);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void orc_reppen_mask(void* r, float* mask_out) {
    RepPen* rp = (RepPen*)r;
    std::memcpy(mask_out, rp->mask.data(), sizeof(float) * rp->mask.size());
}
void orc_rng_chacha12_block(const uint32_t* key8, uint64_t counter, uint32_t* out16) { rng_chacha12_block(key8, counter, out16); }
void orc_rng_seed_key(uint64_t seed, uint32_t* key8) { rng_seed_key(seed, key8); }
void orc_rng_stream(uint64_t seed, int n32, uint32_t* out32, int n64, uint64_t* out64) { rng_stream(seed, n32, out32, n64, out64); }
void orc_rng_weighted_index(uint64_t seed, const float* w, int n, int draws, uint32_t* out) { rng_weighted_index(seed, w, n, draws, out); }
void orc_batched_sample(uint64_t seed, double temp, double top_p, uint64_t top_k, const float* logits, int B, int n, int call_index, uint32_t* out) {
    Sampling s; s.temp = temp; s.top_p = top_p; s.top_k = top_k; s.repetition_penalty = 1.f;
    rng_batched_sample(seed, s, logits, B, n, call_index, out);
}
void* orc_sampler_create(uint64_t seed, double temp, double top_p, uint64_t top_k) {
    Sampling s; s.temp = temp; s.top_p = top_p; s.top_k = top_k;
    return new LogitsProcessor(seed, s);
}
void orc_sampler_destroy(void* s) { delete (LogitsProcessor*)s; }
uint32_t orc_sampler_sample(void* s, const float* logits, uint64_t n) { return ((LogitsProcessor*)s)->sample(logits, n); }

// ---- codec
void* orc_codec_create(int tiny) {
    Codec* c = new Codec();
    if (tiny) c->init_tiny(); else c->init_fish15();
    return c;
}
void orc_codec_destroy(void* c) { delete (Codec*)c; }
int orc_codec_load_synthetic(void* c, uint64_t seed) { GUARD(((Codec*)c)->load_synthetic(seed)) }
int orc_codec_set_tensor(void* c, const char* name, const float* data, uint64_t n) { GUARD(((Codec*)c)->set_tensor(name, data, (size_t)n)) }
int orc_codec_hop(void* c) { return ((Codec*)c)->hop(); }
void orc_codec_fsq_code(void* c, uint32_t idx, float* code4) { ((Codec*)c)->fsq_code(idx, code4); }
// pcm_out: (hop*T).  If stage_out != null, stage `stage_idx` (0 = quantizer output, 1..2 upsample, 3 conv_pre,
// 4..8 HiFiGAN stages) is copied there (caller sizes it).
int orc_codec_decode(void* c, const uint32_t* codes, int T, float* pcm_out, int stage_idx, float* stage_out) {
    try {
        std::vector<std::vector<float>> st;
        auto pcm = ((Codec*)c)->decode(codes, T, stage_out ? &st : nullptr);
        std::memcpy(pcm_out, pcm.data(), sizeof(float) * pcm.size());
        if (stage_out) std::memcpy(stage_out, st.at(stage_idx).data(), sizeof(float) * st.at(stage_idx).size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// ---- encoder side.  mel_out: (n_mels * frames) channel-first; codes_out: (n_groups * L)
int orc_codec_mel_filterbank(int sr, int n_fft, int n_mels, float* out) {
    try { auto fb = Codec::mel_filterbank(sr, n_fft, n_mels); std::memcpy(out, fb.data(), sizeof(float) * fb.size()); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_codec_log_mel(void* c, const float* pcm, int n, float* mel_out, int cap_frames, int* frames) {
    try {
        auto mel = ((Codec*)c)->log_mel(pcm, n, frames);
        if (*frames > cap_frames) throw std::runtime_error("mel_out too small");
        std::memcpy(mel_out, mel.data(), sizeof(float) * mel.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_codec_encode_mel(void* c, const float* mel, int frames, uint32_t* codes_out, int cap_L, int* L, int stage_idx, float* stage_out) {
    try {
        Codec* cc = (Codec*)c;
        std::vector<float> m(mel, mel + (size_t)cc->n_mels * frames);
        std::vector<std::vector<float>> st;
        auto idx = cc->encode_mel(m, frames, L, stage_out ? &st : nullptr);
        if (*L > cap_L) throw std::runtime_error("codes_out too small");
        std::memcpy(codes_out, idx.data(), sizeof(uint32_t) * idx.size());
        if (stage_out) std::memcpy(stage_out, st.at(stage_idx).data(), sizeof(float) * st.at(stage_idx).size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_codec_encode(void* c, const float* pcm, int n, uint32_t* codes_out, int cap_L, int* L) {
    try {
        auto idx = ((Codec*)c)->encode(pcm, n, L);
        if (*L > cap_L) throw std::runtime_error("codes_out too small");
        std::memcpy(codes_out, idx.data(), sizeof(uint32_t) * idx.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
