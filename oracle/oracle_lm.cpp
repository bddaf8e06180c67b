// TEST INFRASTRUCTURE ONLY (oracle/).  See oracle_lm.h for the parity statement.
// CPU f32 restatement, op-for-op, of:
//   fish_speech_core/lib/lm/dual_ar.rs                  (model)
//   fish_speech_core/lib/lm/generate/single_batch.rs    (batch-1 generator)
//   fish_speech_core/lib/lm/generate/utils.rs           (audio-range constraint)
//   fish_speech_core/lib/lm/sampling/{mod,rep_pen}.rs   (sampling, repetition penalty)
// including the reference's deliberate inefficiencies that matter for the "Candle-CPU
// equivalent" timing baseline: the KV cache is re-concatenated every step (dual_ar.rs:316-324)
// and the slow head is computed over the full vocabulary (dual_ar.rs:631).
#include "oracle_lm.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>

#include "fsgen.h"

namespace oracle {

// ---------------------------------------------------------------- basic ops (candle semantics)

// candle_nn::Linear without bias: y = x . W^T   (W row-major [N,K])
static void linear(const float* x, int M, const float* W, int N, int K, float* y) {
#pragma omp parallel for schedule(static) if ((size_t)N * K * M > (1u << 18))
    for (int n = 0; n < N; ++n) {
        const float* w = W + (size_t)n * K;
        for (int m = 0; m < M; ++m) {
            const float* xm = x + (size_t)m * K;
            float acc = 0.f;
#pragma omp simd reduction(+ : acc)
            for (int k = 0; k < K; ++k) acc += xm[k] * w[k];
            y[(size_t)m * N + n] = acc;
        }
    }
}

// candle_nn::RmsNorm (f32): x / sqrt(mean(x^2) + eps) * w
static void rms_norm(const float* x, int M, int D, const float* w, float eps, float* y) {
    for (int m = 0; m < M; ++m) {
        const float* xm = x + (size_t)m * D;
        float s = 0.f;
        for (int i = 0; i < D; ++i) s += xm[i] * xm[i];
        float d = std::sqrt(s / (float)D + eps);
        for (int i = 0; i < D; ++i) y[(size_t)m * D + i] = (xm[i] / d) * w[i];
    }
}

static inline float silu(float x) { return x / (1.f + std::exp(-x)); }

// dual_ar.rs:168-186
void precompute_freqs(const ModelArgs& a, std::vector<float>& cos_t, std::vector<float>& sin_t) {
    const int n_elem = a.dim / a.n_head;
    const int half = n_elem / 2;
    std::vector<float> theta(half);
    for (int j = 0; j < half; ++j) theta[j] = 1.f / std::pow(a.rope_base, (float)(2 * j) / (float)n_elem);
    cos_t.resize((size_t)a.max_seq_len * half);
    sin_t.resize((size_t)a.max_seq_len * half);
    for (int p = 0; p < a.max_seq_len; ++p)
        for (int j = 0; j < half; ++j) {
            float ang = (float)p * theta[j];
            cos_t[(size_t)p * half + j] = std::cos(ang);
            sin_t[(size_t)p * half + j] = std::sin(ang);
        }
}

// dual_ar.rs:702-712.  1 = masked.
void get_mask_abs(int size1, int size2, int context, uint8_t* mask) {
    for (int i = 0; i < size1; ++i)
        for (int j = 0; j < size2; ++j)
            mask[(size_t)i * size2 + j] = (uint8_t)((size1 + j > size2 + i) || (size1 + j + context < size2 + i));
}

// ---------------------------------------------------------------- rep-pen (rep_pen.rs:37-65)
void RepPen::apply(std::vector<float>& logits, size_t last_token) {
    if (last_token >= mask.size()) throw std::runtime_error("Token must be within vocab size");
    // `entry(last_token).or_insert(1)`; count is never incremented, so `*count == 1` always holds
    seen.insert(last_token);
    mask[last_token] = amt;
    context.push_front(last_token);
    if (context.size() > max_ctx) {
        size_t dropped = context.back();
        context.pop_back();
        auto it = seen.find(dropped);
        if (it != seen.end()) {  // count 1 -> 0 -> remove, un-penalise (even if it re-occurs inside the window)
            seen.erase(it);
            mask[dropped] = 1.0f;
        }
    }
    for (size_t i = 0; i < logits.size(); ++i) logits[i] = logits[i] / mask[i];
}

// ---------------------------------------------------------------- model
void LM::init(const ModelArgs& args, const TokenCfg& tc) {
    a = args; t = tc;
    layers.assign(a.n_layer, Block());
    fast_layers.assign(a.n_fast_layer, Block());
    precompute_freqs(a, cos_t, sin_t);
}

// Synthetic init (SURVEY.md §8d "Synthetic weights", with non-trivial norm weights so that a wrong
// norm tensor cannot hide): matrices/embeddings N(0, 0.02^2) (initializer_range, dual_ar.rs:93),
// norm weights 1 + N(0, 0.1^2).  Tensor names are the reference loader's (dual_ar.rs:125-156,219-223,415-419,466-511).
// mode 0: f32 values; 1: every tensor rounded to bf16 (a bf16 checkpoint); 2: the product's FS_FP8 storage -- Linear weights
// quantised per output row to e4m3fn (fsgen::quant_rows_fp8), embeddings and norm vectors rounded to bf16.
void LM::load_synthetic(uint64_t seed, int mode) {
    const bool bf16 = mode != 0, fp8 = mode == 2;
    auto lin = [&](std::vector<float>& dst, size_t rows, size_t cols, const std::string& name) {
        dst.resize(rows * cols);
        fsgen::fill(dst.data(), rows * cols, name, seed, 0.f, 0.02, bf16 && !fp8);
        if (fp8) fsgen::quant_rows_fp8(dst.data(), rows, cols);
    };
    auto emb = [&](std::vector<float>& dst, size_t n, const std::string& name) {
        dst.resize(n);
        fsgen::fill(dst.data(), n, name, seed, 0.f, 0.02, bf16);
    };
    auto nrm = [&](std::vector<float>& dst, size_t n, const std::string& name) {
        dst.resize(n);
        fsgen::fill(dst.data(), n, name, seed, 1.f, 0.1, bf16);
    };
    const size_t D = a.dim, I = a.intermediate_size;
    const size_t QKV = (size_t)(a.n_head + 2 * a.n_local_heads) * a.head_dim;
    emb(embeddings, (size_t)a.vocab_size * D, "embeddings.weight");
    emb(codebook_embeddings, (size_t)a.codebook_size * a.num_codebooks * D, "codebook_embeddings.weight");
    auto blocks = [&](std::vector<Block>& ls, const std::string& pre) {
        for (size_t l = 0; l < ls.size(); ++l) {
            std::string p = pre + std::to_string(l) + ".";
            lin(ls[l].wqkv, QKV, D, p + "attention.wqkv.weight");
            lin(ls[l].wo, D, D, p + "attention.wo.weight");
            lin(ls[l].w1, I, D, p + "feed_forward.w1.weight");
            lin(ls[l].w2, D, I, p + "feed_forward.w2.weight");
            lin(ls[l].w3, I, D, p + "feed_forward.w3.weight");
            nrm(ls[l].ffn_norm, D, p + "ffn_norm.weight");
            nrm(ls[l].attention_norm, D, p + "attention_norm.weight");
        }
    };
    blocks(layers, "layers.");
    nrm(norm, D, "norm.weight");
    if (a.tie_word_embeddings) {  // dual_ar.rs:482-486
        if (fp8) lin(output, (size_t)a.vocab_size, D, "embeddings.weight");  // head = fp8 of the same checkpoint tensor
        else output = embeddings;
    } else lin(output, (size_t)a.vocab_size, D, "output.weight");
    emb(fast_embeddings, (size_t)a.codebook_size * D, "fast_embeddings.weight");
    blocks(fast_layers, "fast_layers.");
    nrm(fast_norm, D, "fast_norm.weight");
    lin(fast_output, (size_t)a.codebook_size, D, "fast_output.weight");
}

// dual_ar.rs:532-567
void LM::embed(const uint32_t* toks, int B, int L, float* x) {
    const int C = a.num_codebooks, D = a.dim;
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < L; ++l) {
            const uint32_t sem = toks[((size_t)b * (C + 1) + 0) * L + l];
            if (sem >= (uint32_t)a.vocab_size) throw std::runtime_error("semantic token out of vocab");
            bool keep = t.has_semantic_end ? (sem <= t.semantic_end_id && sem >= t.semantic_start_id)
                                           : (sem == t.semantic_start_id);
            const float m = keep ? 1.f : 0.f;
            float* xo = x + ((size_t)b * L + l) * D;
            // cat([semantic, codebook_0..C-1]).sum(1): sequential accumulation from zero
            for (int d = 0; d < D; ++d) xo[d] = 0.f + embeddings[(size_t)sem * D + d];
            for (int c = 0; c < C; ++c) {
                const uint32_t code = toks[((size_t)b * (C + 1) + 1 + c) * L + l];
                if (code >= (uint32_t)a.codebook_size) throw std::runtime_error("codebook token out of range");
                const float* e = &codebook_embeddings[((size_t)c * a.codebook_size + code) * D];
                for (int d = 0; d < D; ++d) xo[d] += e[d] * m;
            }
        }
}

// TransformerBlock::forward (dual_ar.rs:429-440) with Attention::forward (:281-384) and FeedForward (:160-165).
// x: (B, L, D) in place.  RoPE rows [input_pos, input_pos+L).  Mask applied only when L > 1 (:360).
void LM::block_forward(Block& blk, float* x, int B, int L, int input_pos, int /*unused*/) {
    const int D = a.dim, H = a.n_head, Hk = a.n_local_heads, Dh = a.head_dim, I = a.intermediate_size;
    const int QKV = (H + 2 * Hk) * Dh, half = Dh / 2, M = B * L;
    std::vector<float> xn((size_t)M * D), qkv((size_t)M * QKV);
    rms_norm(x, M, D, blk.attention_norm.data(), a.norm_eps, xn.data());
    linear(xn.data(), M, blk.wqkv.data(), QKV, D, qkv.data());

    if (input_pos + L > a.max_seq_len) throw std::runtime_error("input_pos + seqlen exceeds max_seq_len (dual_ar.rs:623)");
    // q: (B,H,L,Dh), new k/v: (B,Hk,L,Dh); interleaved RoPE (rope_i) on q and k (:239-249)
    std::vector<float> q((size_t)B * H * L * Dh), kn((size_t)B * Hk * L * Dh), vn((size_t)B * Hk * L * Dh);
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < L; ++l) {
            const float* row = &qkv[((size_t)b * L + l) * QKV];
            const float* cs = &cos_t[(size_t)(input_pos + l) * half];
            const float* sn = &sin_t[(size_t)(input_pos + l) * half];
            for (int h = 0; h < H; ++h)
                for (int j = 0; j < half; ++j) {
                    float x0 = row[h * Dh + 2 * j], x1 = row[h * Dh + 2 * j + 1];
                    float* o = &q[(((size_t)b * H + h) * L + l) * Dh];
                    o[2 * j] = x0 * cs[j] - x1 * sn[j];
                    o[2 * j + 1] = x0 * sn[j] + x1 * cs[j];
                }
            for (int h = 0; h < Hk; ++h) {
                for (int j = 0; j < half; ++j) {
                    float x0 = row[H * Dh + h * Dh + 2 * j], x1 = row[H * Dh + h * Dh + 2 * j + 1];
                    float* o = &kn[(((size_t)b * Hk + h) * L + l) * Dh];
                    o[2 * j] = x0 * cs[j] - x1 * sn[j];
                    o[2 * j + 1] = x0 * sn[j] + x1 * cs[j];
                }
                for (int d = 0; d < Dh; ++d)
                    vn[(((size_t)b * Hk + h) * L + l) * Dh + d] = row[(H + Hk) * Dh + h * Dh + d];
            }
        }
    if (kv_round_bf16) {
        for (auto& f : kn) f = fsgen::round_bf16(f);
        for (auto& f : vn) f = fsgen::round_bf16(f);
    }
    if (!blk.force_k.empty()) {  // test hook: see Block::force_k
        if (B != 1 || L != 1 || blk.force_k.size() != kn.size()) throw std::runtime_error("orc_lm_force_kv: needs a batch-1 single-token step");
        float worst = 0.f;
        for (size_t i = 0; i < kn.size(); ++i) {
            // bf16 ulp (8 significant bits) of the forced value, floored at 2^-17: below |x| ~ 1e-3 the f32 summation-order noise of the
            // projection (~1e-6 absolute) is worth several ulps of the tiny value without being a rounding-boundary flip of anything that matters
            const float uk = std::ldexp(1.f, std::max(std::ilogb(std::max(std::fabs(blk.force_k[i]), 1e-30f)) - 7, -17));
            const float uv = std::ldexp(1.f, std::max(std::ilogb(std::max(std::fabs(blk.force_v[i]), 1e-30f)) - 7, -17));
            worst = std::max(worst, std::max(std::fabs(kn[i] - blk.force_k[i]) / uk, std::fabs(vn[i] - blk.force_v[i]) / uv));
        }
        blk.force_diff = worst;
        kn = blk.force_k; vn = blk.force_v;
        blk.force_k.clear(); blk.force_v.clear();
    }
    // Tensor::cat(&[prev, new], 2): full re-copy every call (:316-324)
    const int Tp = blk.kv_len, T = Tp + L;
    if (Tp > 0 && blk.kv_b != B) throw std::runtime_error("KV cache batch mismatch");
    {
        std::vector<float> k2((size_t)B * Hk * T * Dh), v2((size_t)B * Hk * T * Dh);
        for (int bh = 0; bh < B * Hk; ++bh) {
            if (Tp) {
                std::memcpy(&k2[(size_t)bh * T * Dh], &blk.k[(size_t)bh * Tp * Dh], sizeof(float) * Tp * Dh);
                std::memcpy(&v2[(size_t)bh * T * Dh], &blk.v[(size_t)bh * Tp * Dh], sizeof(float) * Tp * Dh);
            }
            std::memcpy(&k2[((size_t)bh * T + Tp) * Dh], &kn[(size_t)bh * L * Dh], sizeof(float) * L * Dh);
            std::memcpy(&v2[((size_t)bh * T + Tp) * Dh], &vn[(size_t)bh * L * Dh], sizeof(float) * L * Dh);
        }
        blk.k.swap(k2); blk.v.swap(v2); blk.kv_len = T; blk.kv_b = B;
    }
    // mask (only when L > 1): get_mask_abs(L, T) (:585-588)
    std::vector<uint8_t> mask;
    if (L > 1) { mask.resize((size_t)L * T); get_mask_abs(L, T, a.max_seq_len, mask.data()); }
    const float scale = 1.f / std::sqrt((float)Dh);
    const int n_rep = H / Hk;
    std::vector<float> y((size_t)M * D);  // (B, L, H*Dh)
#pragma omp parallel for collapse(2) schedule(static) if ((size_t)B * H * L * T * Dh > (1u << 18))
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const int hk = h / n_rep;  // repeat_kv / expand+reshape (:328-357): q head h reads kv head h / n_rep
            const float* K = &blk.k[((size_t)b * Hk + hk) * T * Dh];
            const float* V = &blk.v[((size_t)b * Hk + hk) * T * Dh];
            std::vector<float> w(T);
            for (int l = 0; l < L; ++l) {
                const float* ql = &q[(((size_t)b * H + h) * L + l) * Dh];
                float mx = -std::numeric_limits<float>::infinity();
                for (int tt = 0; tt < T; ++tt) {
                    float acc = 0.f;
                    for (int d = 0; d < Dh; ++d) acc += ql[d] * (K[(size_t)tt * Dh + d] * scale);  // q . (k^T * scale) (:260)
                    if (L > 1 && mask[(size_t)l * T + tt]) acc = -std::numeric_limits<float>::infinity();
                    w[tt] = acc;
                    mx = std::max(mx, acc);
                }
                float sum = 0.f;
                for (int tt = 0; tt < T; ++tt) { w[tt] = std::exp(w[tt] - mx); sum += w[tt]; }
                for (int tt = 0; tt < T; ++tt) w[tt] /= sum;
                float* yo = &y[((size_t)b * L + l) * D + (size_t)h * Dh];
                for (int d = 0; d < Dh; ++d) yo[d] = 0.f;
                for (int tt = 0; tt < T; ++tt)
                    for (int d = 0; d < Dh; ++d) yo[d] += w[tt] * V[(size_t)tt * Dh + d];
            }
        }
    std::vector<float> att((size_t)M * D);
    linear(y.data(), M, blk.wo.data(), D, D, att.data());
    for (size_t i = 0; i < (size_t)M * D; ++i) x[i] = x[i] + att[i];  // residual + attention (:437)
    // FFN: w2(silu(w1 x) * w3 x) (:160-165)
    rms_norm(x, M, D, blk.ffn_norm.data(), a.norm_eps, xn.data());
    std::vector<float> h1((size_t)M * I), h3((size_t)M * I);
    linear(xn.data(), M, blk.w1.data(), I, D, h1.data());
    linear(xn.data(), M, blk.w3.data(), I, D, h3.data());
    for (size_t i = 0; i < (size_t)M * I; ++i) h1[i] = silu(h1[i]) * h3[i];
    linear(h1.data(), M, blk.w2.data(), D, I, att.data());
    for (size_t i = 0; i < (size_t)M * D; ++i) x[i] = x[i] + att[i];
}

// dual_ar.rs:574-635
void LM::forward_generate(const uint32_t* toks, int B, int L, int input_pos, float* logits, float* hidden,
                          bool full_vocab_head) {
    const int D = a.dim;
    std::vector<float> x((size_t)B * L * D);
    embed(toks, B, L, x.data());
    for (auto& blk : layers) block_forward(blk, x.data(), B, L, input_pos, 0);
    std::vector<float> last((size_t)B * D), nrm((size_t)B * D);
    for (int b = 0; b < B; ++b)
        std::memcpy(&last[(size_t)b * D], &x[((size_t)b * L + (L - 1)) * D], sizeof(float) * D);  // narrow(1, L-1, 1)
    rms_norm(last.data(), B, D, norm.data(), a.norm_eps, nrm.data());
    if (logits) {
        if (full_vocab_head) {
            linear(nrm.data(), B, output.data(), a.vocab_size, D, logits);
        } else {  // test-speed option: only rows [im_end, V) are ever consumed downstream (utils.rs:15)
            const int lo = (int)std::min(t.im_end_id, t.has_semantic_end ? t.semantic_start_id : t.im_end_id);  // (generic layout: utils.rs:17-30)
            std::vector<float> part((size_t)B * (a.vocab_size - lo));
            linear(nrm.data(), B, &output[(size_t)lo * D], a.vocab_size - lo, D, part.data());
            for (int b = 0; b < B; ++b) {
                for (int v = 0; v < lo; ++v) logits[(size_t)b * a.vocab_size + v] = 0.f;
                std::memcpy(&logits[(size_t)b * a.vocab_size + lo], &part[(size_t)b * (a.vocab_size - lo)],
                            sizeof(float) * (a.vocab_size - lo));
            }
        }
    }
    if (hidden) std::memcpy(hidden, last.data(), sizeof(float) * B * D);  // pre-norm hidden (:629-634)
}

// dual_ar.rs:638-673
void LM::forward_generate_fast(const float* xin, int B, int input_pos, float* logits) {
    const int D = a.dim;
    std::vector<float> x(xin, xin + (size_t)B * D), nrm((size_t)B * D);
    for (auto& blk : fast_layers) block_forward(blk, x.data(), B, 1, input_pos, 0);
    rms_norm(x.data(), B, D, fast_norm.data(), a.norm_eps, nrm.data());
    linear(nrm.data(), B, fast_output.data(), a.codebook_size, D, logits);
}

void LM::clear_fast() { for (auto& b : fast_layers) { b.k.clear(); b.v.clear(); b.kv_len = 0; b.kv_b = 0; } }
void LM::clear_slow() { for (auto& b : layers) { b.k.clear(); b.v.clear(); b.kv_len = 0; b.kv_b = 0; } }
void LM::clear_slow_until(int pos) {  // NOT inclusive (dual_ar.rs:391-404)
    const int Hk = a.n_local_heads, Dh = a.head_dim;
    for (auto& blk : layers) {
        if (blk.kv_len == 0) continue;
        const int T = blk.kv_len, nT = std::min(T, pos), BH = blk.kv_b * Hk;
        std::vector<float> k2((size_t)BH * nT * Dh), v2((size_t)BH * nT * Dh);
        for (int bh = 0; bh < BH; ++bh) {
            std::memcpy(&k2[(size_t)bh * nT * Dh], &blk.k[(size_t)bh * T * Dh], sizeof(float) * nT * Dh);
            std::memcpy(&v2[(size_t)bh * nT * Dh], &blk.v[(size_t)bh * T * Dh], sizeof(float) * nT * Dh);
        }
        blk.k.swap(k2); blk.v.swap(v2); blk.kv_len = nT;
    }
}

// ---------------------------------------------------------------- rand 0.8.5 StdRng (= rand_chacha 0.3.1 ChaCha12Rng)
struct ChaCha12Rng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[64];
    int idx = 64;
    static inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    ChaCha12Rng(const uint32_t* k, int) { std::memcpy(key, k, sizeof(key)); }  // from_seed: key words as given (known-answer tests)
    explicit ChaCha12Rng(uint64_t state) {  // rand_core SeedableRng::seed_from_u64 (PCG32 expansion)
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    void block(uint64_t ctr, uint32_t* out) {
        uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3],
                          key[4], key[5], key[6], key[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), 0, 0};
        uint32_t w[16];
        std::memcpy(w, s, sizeof(s));
#define QR(a, b, c, d) \
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12); \
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
        for (int r = 0; r < 6; ++r) {
            QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
            QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
        }
#undef QR
        for (int i = 0; i < 16; ++i) out[i] = w[i] + s[i];
    }
    void refill() {  // BlockRng over 4 consecutive blocks (64 words)
        for (int b = 0; b < 4; ++b) block(counter + b, buf + 16 * b);
        counter += 4;
        idx = 0;
    }
    uint32_t next_u32() {
        if (idx >= 64) refill();
        return buf[idx++];
    }
    uint64_t next_u64() {  // rand_core BlockRng::next_u64
        if (idx < 63) {
            uint64_t lo = buf[idx], hi = buf[idx + 1];
            idx += 2;
            return (hi << 32) | lo;
        } else if (idx >= 64) {
            refill();
            uint64_t lo = buf[0], hi = buf[1];
            idx = 2;
            return (hi << 32) | lo;
        } else {
            uint64_t lo = buf[63];
            refill();
            uint64_t hi = buf[0];
            idx = 1;
            return (hi << 32) | lo;
        }
    }
};

// rand 0.8.5 WeightedIndex<f32>::new + sample (UniformFloat<f32>)
static uint32_t weighted_index_sample(ChaCha12Rng& rng, const std::vector<float>& w) {
    if (w.empty()) return 0;
    std::vector<float> cum;
    cum.reserve(w.size());
    float total = w[0];
    if (!(total >= 0.f)) return 0;  // InvalidWeight -> unwrap_or(0) (sampling/mod.rs:116)
    for (size_t i = 1; i < w.size(); ++i) {
        if (!(w[i] >= 0.f)) return 0;
        cum.push_back(total);
        total += w[i];
    }
    if (total == 0.f) return 0;
    // UniformFloat::new(0, total)
    const float low = 0.f, high = total;
    uint32_t mr_bits = (0xFFFFFFFFu >> 9) | (127u << 23);
    float max_rand; std::memcpy(&max_rand, &mr_bits, 4); max_rand -= 1.0f;
    float scale = high - low;
    while (scale * max_rand + low >= high) {
        uint32_t b; std::memcpy(&b, &scale, 4); b -= 1; std::memcpy(&scale, &b, 4);
    }
    uint32_t bits = (rng.next_u32() >> 9) | (127u << 23);
    float v12; std::memcpy(&v12, &bits, 4);
    float chosen = (v12 - 1.0f) * scale + low;
    // first item whose cumulative weight is > chosen
    size_t lo = 0, hi = cum.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (cum[mid] <= chosen) lo = mid + 1; else hi = mid; }
    return (uint32_t)lo;
}

// ---- test hooks: the RNG chain piece by piece (pinned in tests/test_oracle_known_answers.py against published vectors and an
// independent pure-Python restatement)
void rng_chacha12_block(const uint32_t* key8, uint64_t counter, uint32_t* out16) {
    ChaCha12Rng r(key8, 0);
    r.block(counter, out16);
}
void rng_seed_key(uint64_t seed, uint32_t* key8) {
    ChaCha12Rng r(seed);
    std::memcpy(key8, r.key, sizeof(r.key));
}
void rng_stream(uint64_t seed, int n_u32_first, uint32_t* out32, int n_u64, uint64_t* out64) {
    ChaCha12Rng r(seed);
    for (int i = 0; i < n_u32_first; ++i) out32[i] = r.next_u32();
    for (int i = 0; i < n_u64; ++i) out64[i] = r.next_u64();
}
void rng_weighted_index(uint64_t seed, const float* w, int n, int draws, uint32_t* out) {
    ChaCha12Rng r(seed);
    std::vector<float> wv(w, w + n);
    for (int i = 0; i < draws; ++i) out[i] = weighted_index_sample(r, wv);
}

LogitsProcessor::LogitsProcessor(uint64_t seed, const Sampling& sa) : s(sa), rng(new ChaCha12Rng(seed)) {}
LogitsProcessor::~LogitsProcessor() { delete rng; }

// sampling/mod.rs:119-132 (and candle_transformers LogitsProcessor::sample_topp)
static uint32_t sample_topp(ChaCha12Rng& rng, std::vector<float>& probs, float top_p) {
    std::vector<size_t> idx(probs.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t i, size_t j) { return probs[i] > probs[j]; });
    float cumsum = 0.f;
    for (size_t i : idx) {
        if (cumsum >= top_p) probs[i] = 0.f;
        cumsum += probs[i];
    }
    return weighted_index_sample(rng, probs);
}

// LogitsProcessor::sample for Sampling::ArgMax / TopKThenTopP (single_batch.rs:38-46).
// Tie rule of the host ArgMax: `iter().enumerate().max_by(total_cmp)` => the LAST maximal index wins.
// top-k: the reference uses `select_nth_unstable_by`, whose output ORDER is unspecified; this restatement
// (and the HIP path) fix the order to ascending token index, which leaves the sampling distribution unchanged.
// top-k then top-p then WeightedIndex draw on a probability vector (shared by the single and the batched processor)
// top_p64: BatchedLogitsProcessor::sample_single_top_p_k keeps top_p as f64 and compares it with `sum_p as f64` (sampling/mod.rs:68);
// candle's LogitsProcessor (the single-sequence path) narrows it to f32 first.  NaN = the single-sequence rule.
static uint32_t sample_probs(ChaCha12Rng& rng, std::vector<float>& p, size_t top_k, float top_p, double top_p64 = std::nan("")) {
    const size_t n = p.size();
    if (top_k == 0 || top_k >= n) return sample_topp(rng, p, top_p);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t i, size_t j) { return p[i] > p[j]; });
    std::vector<size_t> keep(idx.begin(), idx.begin() + top_k);
    std::sort(keep.begin(), keep.end());
    std::vector<float> tk(top_k);
    float sum_p = 0.f;
    for (size_t i = 0; i < top_k; ++i) { tk[i] = p[keep[i]]; sum_p += tk[i]; }
    const bool all = std::isnan(top_p64) ? (top_p <= 0.f || top_p >= sum_p) : (top_p64 <= 0.0 || top_p64 >= (double)sum_p);
    uint32_t j = all ? weighted_index_sample(rng, tk) : sample_topp(rng, tk, top_p);
    return (uint32_t)keep[j];
}

static void softmax_temp(const float* logits, size_t n, float inv_t, std::vector<float>& p) {
    p.resize(n);
    float mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; ++i) { p[i] = logits[i] * inv_t; mx = std::max(mx, p[i]); }
    double sum = 0.0;  // f64 denominator: independent of reduction order (see LogitsProcessor::sample)
    for (size_t i = 0; i < n; ++i) { p[i] = std::exp(p[i] - mx); sum += (double)p[i]; }
    const float denom = (float)sum;
    for (size_t i = 0; i < n; ++i) p[i] /= denom;
}

// BatchedLogitsProcessor::sample (sampling/mod.rs:77-109): rows (B, n).  temp <= 1e-7 -> device argmax (FIRST max);
// else per-row child StdRng seeded from the master's next u64 (:93-95).
std::vector<uint32_t> batched_sample(ChaCha12Rng& master, const Sampling& s, const float* logits, size_t B, size_t n, size_t ld) {
    std::vector<uint32_t> out(B);
    if (s.temp <= 1e-7) {
        for (size_t b = 0; b < B; ++b) {
            const float* l = logits + b * ld;
            size_t best = 0;
            for (size_t i = 1; i < n; ++i) if (l[i] > l[best]) best = i;
            out[b] = (uint32_t)best;
        }
        return out;
    }
    std::vector<uint64_t> seeds(B);
    for (size_t b = 0; b < B; ++b) seeds[b] = master.next_u64();
    const float inv_t = (float)(1.0 / s.temp);
    for (size_t b = 0; b < B; ++b) {
        ChaCha12Rng child(seeds[b]);
        std::vector<float> p;
        softmax_temp(logits + b * ld, n, inv_t, p);
        out[b] = sample_probs(child, p, (size_t)s.top_k, (float)s.top_p, s.top_p);
    }
    return out;
}

void rng_batched_sample(uint64_t seed, const Sampling& s, const float* logits, int B, int n, int call_index, uint32_t* out) {
    ChaCha12Rng master(seed);
    for (long long i = 0; i < (long long)call_index * B; ++i) (void)master.next_u64();  // the earlier sample() calls of the request
    const std::vector<uint32_t> r = batched_sample(master, s, logits, (size_t)B, (size_t)n, (size_t)n);
    std::memcpy(out, r.data(), sizeof(uint32_t) * B);
}

uint32_t LogitsProcessor::sample(const float* logits, size_t n) {
    if (s.temp == 0.0) {
        size_t best = 0;
        for (size_t i = 1; i < n; ++i)
            if (!(logits[i] < logits[best])) best = i;  // >= : last max wins
        return (uint32_t)best;
    }
    const float inv_t = (float)(1.0 / s.temp);
    std::vector<float> p(n);
    float mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; ++i) { p[i] = logits[i] * inv_t; mx = std::max(mx, p[i]); }
    // softmax denominator accumulated in f64 so that it does not depend on the reduction order (candle's own CPU
    // reduction order is not specified either); the HIP sampler does the same
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) { p[i] = std::exp(p[i] - mx); sum += (double)p[i]; }
    const float denom = (float)sum;
    for (size_t i = 0; i < n; ++i) p[i] /= denom;
    const size_t top_k = (size_t)s.top_k;
    const float top_p = (float)s.top_p;
    if (top_k == 0 || top_k >= n) return sample_topp(*rng, p, top_p);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t i, size_t j) { return p[i] > p[j]; });
    std::vector<size_t> keep(idx.begin(), idx.begin() + top_k);
    std::sort(keep.begin(), keep.end());
    std::vector<float> tk(top_k);
    float sum_p = 0.f;
    for (size_t i = 0; i < top_k; ++i) { tk[i] = p[keep[i]]; sum_p += tk[i]; }
    uint32_t j = (top_p <= 0.f || top_p >= sum_p) ? weighted_index_sample(*rng, tk) : sample_topp(*rng, tk, top_p);
    return (uint32_t)keep[j];
}

// ---------------------------------------------------------------- generate_blocking (single_batch.rs)
std::vector<uint32_t> LM::generate(const uint32_t* prompt, int L, int max_new_tokens, const Sampling& s, uint64_t seed,
                                   bool ignore_eos, int* n_frames, std::vector<float>* hidden_out, double* prefill_s,
                                   double* decode_s, int max_frames, std::vector<float>* margins) {
    const int C = a.num_codebooks, D = a.dim, V = a.vocab_size;
    using clk = std::chrono::steady_clock;
    // SingleBatchGenerator::new (:31-70)
    LogitsProcessor lp(seed, s);
    std::vector<RepPen> rp(C);
    for (auto& r : rp) r.init(a.codebook_size, 16, s.repetition_penalty);
    size_t input_pos = (size_t)kv_len();
    const size_t max_pos = (size_t)max_new_tokens + (size_t)kv_len();  // budget counts prompt tokens (:61,77)
    std::vector<uint32_t> cur(prompt, prompt + (size_t)(C + 1) * L);
    int curL = L;
    bool have_prompt = true, have_prev = false;
    std::vector<uint32_t> prev_codes;
    std::vector<std::vector<uint32_t>> frames;  // each [C+1]
    std::vector<float> logits((size_t)V), hidden(D), fl(a.codebook_size), x(D);
    // test aid: smallest top-2 logit margin among the 9 sampling decisions of each iteration (greedy parity tests use
    // it to tell a kernel bug from a legitimate near-tie flip under reduced-precision storage)
    auto top2_margin = [](const float* v, size_t n) {
        float a = -std::numeric_limits<float>::infinity(), b = a;
        for (size_t i = 0; i < n; ++i) { if (v[i] > a) { b = a; a = v[i]; } else if (v[i] > b) b = v[i]; }
        return a - b;
    };
    auto t0 = clk::now();
    auto t_first = t0;
    int it = 0;
    while (true) {
        if (input_pos > max_pos) break;   // :77
        if (!have_prompt) break;          // :86
        if (max_frames >= 0 && it >= max_frames) break;  // bench-only bounded sample
        forward_generate(cur.data(), 1, curL, (int)input_pos, logits.data(), hidden.data(), true);
        // slow token: constrain_probs_to_audio + sample + rescale (Fish 1.5 contiguous case, utils.rs:13-16,45-46)
        uint32_t semantic;
        if (t.has_semantic_end) {
            // utils.rs:13-16: contiguous slice when <|im_end|> directly precedes the semantic range; :17-30: else cat(im_end logit, logits of
            // [semantic_start, V)) -- whatever follows the range (control tokens, <|im_end|> itself) stays a candidate
            const bool adjacent = t.im_end_id == t.semantic_start_id - 1;
            std::vector<float> sl;
            if (adjacent) sl.assign(logits.begin() + t.im_end_id, logits.end());
            else { sl.push_back(logits[t.im_end_id]); sl.insert(sl.end(), logits.begin() + t.semantic_start_id, logits.end()); }
            if (ignore_eos) sl[0] = -std::numeric_limits<float>::infinity();
            if (margins) margins->push_back(top2_margin(sl.data(), sl.size()));
            const uint32_t pick = lp.sample(sl.data(), sl.size());
            // rescale_semantic_tokens (:45-52)
            semantic = adjacent ? pick + t.im_end_id : (pick == 0 ? t.im_end_id : pick - 1 + t.semantic_start_id);
        } else {
            // Fish <= 1.4 (single_batch.rs:104-124): legacy_softmax_sample over {pad_id, im_end_id}, temperature ignored
            // (sampling/mod.rs:8-26).  The reference draws `rng.gen::<f32>()` from an unseeded thread_rng; the restatement
            // draws the same Standard<f32> ((next_u32 >> 8) * 2^-24) from the request's seeded StdRng instead.
            const float pad = logits[t.pad_id], eos = logits[t.im_end_id], m = std::max(pad, eos);
            const float e_pad = std::exp(pad - m), e_eos = std::exp(eos - m);
            const float p_pad = e_pad / (e_pad + e_eos);
            const float u = (float)(lp.rng->next_u32() >> 8) * (1.0f / 16777216.0f);
            semantic = (u < p_pad || ignore_eos) ? t.pad_id : t.im_end_id;
            if (margins) margins->push_back(std::fabs(u - p_pad));
        }
        std::vector<uint32_t> cb = {semantic};
        clear_fast();  // :146
        x.assign(hidden.begin(), hidden.end());
        for (int ci = 0; ci < C; ++ci) {
            if (semantic == t.im_end_id) { cb.push_back(0); continue; }  // :153-156
            forward_generate_fast(x.data(), 1, ci, fl.data());
            if (have_prev) rp[ci].apply(fl, prev_codes[ci + 1]);  // :162-168
            if (margins) margins->back() = std::min(margins->back(), top2_margin(fl.data(), fl.size()));
            uint32_t tok = lp.sample(fl.data(), fl.size());
            if (ci != C - 1) std::memcpy(x.data(), &fast_embeddings[(size_t)tok * D], sizeof(float) * D);  // :176-182
            cb.push_back(tok);
        }
        input_pos += have_prev ? 1 : (size_t)curL;  // :193-197
        have_prev = true;
        prev_codes = cb;
        if (semantic == t.im_end_id) have_prompt = false;
        else { cur = cb; curL = 1; }
        // generate_blocking_with_hidden: first frame unconditionally, later frames unless slow tok == im_end (:250,264-266)
        if (it == 0 || cb[0] != t.im_end_id) {
            frames.push_back(cb);
        }
        if (hidden_out && (it == 0 || true)) hidden_out->insert(hidden_out->end(), hidden.begin(), hidden.end());
        if (it == 0) t_first = clk::now();
        ++it;
    }
    auto t1 = clk::now();
    if (prefill_s) *prefill_s = std::chrono::duration<double>(t_first - t0).count();
    if (decode_s) *decode_s = std::chrono::duration<double>(t1 - t_first).count();
    const int n = (int)frames.size();
    if (n_frames) *n_frames = n;
    std::vector<uint32_t> out((size_t)C * n);  // drop row 0 (:280-284)
    for (int f = 0; f < n; ++f)
        for (int c = 0; c < C; ++c) out[(size_t)c * n + f] = frames[f][c + 1];
    return out;
}

// ---------------------------------------------------------------- generate_static_batch (static_batch.rs:17-390), audio_only
std::vector<std::vector<uint32_t>> LM::generate_batch(const std::vector<std::vector<uint32_t>>& prompts, const std::vector<int>& lens,
                                                      int max_new_tokens, const Sampling& s, uint64_t seed, bool ignore_eos,
                                                      std::vector<int>* n_frames, std::vector<float>* margins) {
    const int C = a.num_codebooks, C1 = C + 1, D = a.dim, V = a.vocab_size, B = (int)prompts.size();
    if (B == 0) throw std::runtime_error("Must have at least one prompt");
    if (!t.has_semantic_end) throw std::runtime_error("static batches: only the Fish 1.5 / DualAR token layouts are restated");
    int Lmax = 0;
    for (int l : lens) Lmax = std::max(Lmax, l);
    // pad_prompts (:68-111): left pad with [im_end; 0...]; the mask is built but never applied (dual_ar.rs:589-615)
    std::vector<uint32_t> cur((size_t)B * C1 * Lmax);
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < C1; ++r) {
            const int pad = Lmax - lens[b];
            for (int j = 0; j < pad; ++j) cur[((size_t)b * C1 + r) * Lmax + j] = r == 0 ? t.im_end_id : 0u;
            for (int j = 0; j < lens[b]; ++j) cur[((size_t)b * C1 + r) * Lmax + pad + j] = prompts[b][(size_t)r * lens[b] + j];
        }
    ChaCha12Rng master(seed);  // BatchedLogitsProcessor::new(seed) (:63 passes 42)
    std::vector<bool> dead(B, false);
    std::vector<std::vector<std::vector<uint32_t>>> frames(B);
    std::vector<float> logits((size_t)B * V), hidden((size_t)B * D), fl((size_t)B * a.codebook_size), x((size_t)B * D);
    size_t input_pos = 0;
    int curL = Lmax;
    bool have_prompt = true, first = true;
    const bool adjacent = t.im_end_id == t.semantic_start_id - 1;  // utils.rs:13 / :17-30 (see LM::generate)
    const size_t lo = adjacent ? t.im_end_id : t.semantic_start_id - 1, na = (size_t)V - lo;
    while (true) {
        if (input_pos == 0) clear_slow();                                 // :118-121
        if (!have_prompt || input_pos > (size_t)max_new_tokens) break;    // :122
        forward_generate(cur.data(), B, curL, (int)input_pos, logits.data(), hidden.data(), true);
        std::vector<float> sl((size_t)B * na);
        for (int b = 0; b < B; ++b) {
            std::memcpy(&sl[(size_t)b * na], &logits[(size_t)b * V + lo], sizeof(float) * na);
            if (!adjacent) sl[(size_t)b * na] = logits[(size_t)b * V + t.im_end_id];  // candidate 0 = <|im_end|>, then [semantic_start, V)
            if (ignore_eos) sl[(size_t)b * na] = -std::numeric_limits<float>::infinity();
        }
        // test aid (as in LM::generate): smallest top-2 margin among a row's 9 decisions of this iteration
        auto top2 = [](const float* v, size_t n) {
            float a = -std::numeric_limits<float>::infinity(), b2 = a;
            for (size_t i = 0; i < n; ++i) { if (v[i] > a) { b2 = a; a = v[i]; } else if (v[i] > b2) b2 = v[i]; }
            return a - b2;
        };
        size_t mbase = 0;
        if (margins) { mbase = margins->size(); for (int b = 0; b < B; ++b) margins->push_back(top2(&sl[(size_t)b * na], na)); }
        std::vector<uint32_t> slow = batched_sample(master, s, sl.data(), B, na, na);
        for (auto& v : slow) v = adjacent ? v + t.im_end_id : (v == 0 ? t.im_end_id : v - 1 + t.semantic_start_id);  // rescale_semantic_tokens
        for (int b = 0; b < B; ++b) dead[b] = dead[b] || slow[b] == t.im_end_id;  // :160-173
        x = hidden;
        clear_fast();
        std::vector<std::vector<uint32_t>> ids(B, std::vector<uint32_t>{});
        for (int b = 0; b < B; ++b) ids[b].push_back(slow[b]);
        for (int ci = 0; ci < C; ++ci) {
            forward_generate_fast(x.data(), B, ci, fl.data());
            // rep_pen.apply_mask: the mask is never updated for Fish models (:204-206) -> logits / 1.0
            if (margins) for (int b = 0; b < B; ++b) (*margins)[mbase + b] = std::min((*margins)[mbase + b], top2(&fl[(size_t)b * a.codebook_size], a.codebook_size));
            std::vector<uint32_t> tok = batched_sample(master, s, fl.data(), B, a.codebook_size, a.codebook_size);
            for (int b = 0; b < B; ++b) {
                std::memcpy(&x[(size_t)b * D], &fast_embeddings[(size_t)tok[b] * D], sizeof(float) * D);
                ids[b].push_back(tok[b]);
            }
        }
        std::vector<uint32_t> next((size_t)B * C1);
        bool all_dead = true;
        for (int b = 0; b < B; ++b) {
            const bool is_audio = ids[b][0] >= t.semantic_start_id;        // :229
            std::vector<uint32_t> vq = ids[b];
            if (!is_audio) for (int c = 1; c <= C; ++c) vq[c] = 0;
            if (first || !dead[b]) frames[b].push_back(vq);                // generate_static_batch :305-338
            for (int r = 0; r < C1; ++r) next[(size_t)b * C1 + r] = vq[r];
            all_dead = all_dead && dead[b];
        }
        have_prompt = !all_dead;                                          // :255-261
        cur = next; 
        input_pos += first ? (size_t)Lmax : 1;                            // :262-267
        curL = 1;
        first = false;
    }
    std::vector<std::vector<uint32_t>> out(B);
    if (n_frames) n_frames->assign(B, 0);
    for (int b = 0; b < B; ++b) {
        const int n = (int)frames[b].size();
        out[b].resize((size_t)C * n);
        for (int f = 0; f < n; ++f)
            for (int c = 0; c < C; ++c) out[b][(size_t)c * n + f] = frames[b][f][c + 1];  // drop row 0 (:371-374)
        if (n_frames) (*n_frames)[b] = n;
    }
    return out;
}

}  // namespace oracle
