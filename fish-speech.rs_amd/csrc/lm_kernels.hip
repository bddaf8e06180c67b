// Dual-AR transformer decode kernels for gfx950 (MI355X), batch-1 token path.
//
// Every kernel here is HBM/latency-bound weight streaming (a 1024x1024 bf16 matrix is 2 MB; one CU can keep
// ~32 KB in flight), so the design rules are (MI355X guide, "GEMV / M <= 16 decode weights"):
//   * weights go global -> VGPR directly as 16-B/lane non-temporal loads, all of a wave's loads issued before
//     the first use (no LDS round trip: nothing is shared between waves);
//   * one wave = 64 lanes owns whole rows; the activation slice a lane needs (K/64 floats) is loaded once into
//     registers and reused for every row of the wave;
//   * RMSNorm, RoPE, SwiGLU, residual adds, KV append and the attention combine are fused into the GEMV that
//     produces / consumes them, so a transformer block is 5 launches (4 for the fast decoder);
//   * reductions are wave64 shuffles; no atomics, fixed summation order => run-to-run deterministic tokens.
// Reference semantics implemented: fish_speech_core/lib/lm/dual_ar.rs:118-165 (FFN), :239-249 (rope_i),
// :252-279 (SDPA), :281-384 (Attention::forward), :429-440 (block), :532-567 (embed), :629-631 (head).
#include <hip/hip_runtime.h>

#include "fs_common.h"
#include "fs_synth.h"
#include "lm_kernels.h"

namespace fs {

// ------------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename WT>
struct WTr;
template <>
struct WTr<bf16_t> {
    static constexpr int EPL = 8;  // elements per 16-byte lane load
    using vec = u32x4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
    }
    __device__ static __forceinline__ float to_f32(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
    __device__ static __forceinline__ bf16_t from_f32(float f) {  // RNE
        uint32_t u = __float_as_uint(f);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (bf16_t)(u >> 16);
    }
};
template <>
struct WTr<float> {
    static constexpr int EPL = 4;
    using vec = f32x4;
    __device__ static __forceinline__ void unpack(const f32x4& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    __device__ static __forceinline__ float to_f32(float h) { return h; }
    __device__ static __forceinline__ float from_f32(float f) { return f; }
};

template <typename V>
__device__ __forceinline__ V ld_stream(const V* p) {  // streamed-once weights: non-temporal (guide: nt-weights)
    return __builtin_nontemporal_load(p);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// A wave's view of a length-K vector / weight row: chunk c covers elements [c*64*EPL, (c+1)*64*EPL), lane l owns
// EPL consecutive elements starting at c*64*EPL + l*EPL.
template <typename WT, int K>
struct Row {
    static constexpr int EPL = WTr<WT>::EPL;
    static constexpr int CH = 64 * EPL;
    static constexpr int NCH = (K + CH - 1) / CH;
    static constexpr int NX = NCH * EPL;
    using vec = typename WTr<WT>::vec;

    __device__ static __forceinline__ void load_x(const float* __restrict__ x, int lane, float (&xr)[NX]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int base = c * CH + lane * EPL;
            if (base < K) {
#pragma unroll
                for (int i = 0; i < EPL; i += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(x + base + i);
                    xr[c * EPL + i] = t.x; xr[c * EPL + i + 1] = t.y; xr[c * EPL + i + 2] = t.z; xr[c * EPL + i + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPL; ++i) xr[c * EPL + i] = 0.f;
            }
        }
    }
    __device__ static __forceinline__ void load_w(const WT* __restrict__ row, int lane, vec (&wv)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int base = c * CH + lane * EPL;
            if (base < K) wv[c] = ld_stream(reinterpret_cast<const vec*>(row + base));
            else wv[c] = vec(0);
        }
    }
    __device__ static __forceinline__ float dot(const vec (&wv)[NCH], const float (&xr)[NX]) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float f[EPL];
            WTr<WT>::unpack(wv[c], f);
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc = fmaf(f[i], xr[c * EPL + i], acc);
        }
        return acc;
    }
    // in-register RMSNorm of the wave's x slice: x / sqrt(mean(x^2) + eps) * w   (candle_nn::RmsNorm)
    __device__ static __forceinline__ void rmsnorm(float (&xr)[NX], const float* __restrict__ nw, float eps, int lane) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NX; ++i) ss = fmaf(xr[i], xr[i], ss);
        ss = wave_sum(ss);
        const float d = sqrtf(ss / (float)K + eps);
        float wr[NX];
        load_x(nw, lane, wr);
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = (xr[i] / d) * wr[i];
    }
};

template <typename WT>
__device__ __forceinline__ WT* kv_addr(void* pool, const int* __restrict__ page_table, int t, int g, int Hk, int Dh) {
    const int page = page_table[t / KV_PAGE];
    return reinterpret_cast<WT*>(pool) + ((size_t)(page * Hk + g) * KV_PAGE + (t % KV_PAGE)) * Dh;
}

// ------------------------------------------------------------------------------------------------ qkv + rope + kv append
// One wave per row PAIR (2p, 2p+1) of Wqkv: the interleaved-RoPE partner is in the same wave.
template <typename WT, int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_qkv(const float* __restrict__ x, const float* __restrict__ norm_w, float eps,
                                                    const WT* __restrict__ W, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, const SeqState* __restrict__ state,
                                                    int pos_static, int rope_static, float* __restrict__ q_out, KVView kv,
                                                    int H, int Hk, int Dh) {
    using R = Row<WT, K>;
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int n_pairs = (H + 2 * Hk) * Dh / 2;
    if (pair >= n_pairs) return;
    typename R::vec w0[R::NCH], w1[R::NCH];
    R::load_w(W + (size_t)(2 * pair) * K, lane, w0);       // issue the weight loads first
    R::load_w(W + (size_t)(2 * pair + 1) * K, lane, w1);
    float xr[R::NX];
    R::load_x(x, lane, xr);
    R::rmsnorm(xr, norm_w, eps, lane);
    float a = wave_sum(R::dot(w0, xr));
    float b = wave_sum(R::dot(w1, xr));
    if (lane != 0) return;
    const int pos = state ? state->pos : pos_static;
    const int rpos = state ? state->pos + state->rope_off : rope_static;
    const int r = 2 * pair, qdim = H * Dh, kdim = Hk * Dh, half = Dh / 2;
    if (r < qdim + kdim) {  // q or k: rope_i on the pair (2j, 2j+1) of its head (dual_ar.rs:246-247)
        const int j = (r % Dh) / 2;
        const float c = cos_t[(size_t)rpos * half + j], s = sin_t[(size_t)rpos * half + j];
        const float o0 = a * c - b * s, o1 = a * s + b * c;
        if (r < qdim) {
            q_out[r] = o0; q_out[r + 1] = o1;
        } else {
            const int rk = r - qdim, g = rk / Dh, dd = rk % Dh;
            WT* dst = kv_addr<WT>(kv.k, kv.page_table, pos, g, Hk, Dh) + dd;
            dst[0] = WTr<WT>::from_f32(o0); dst[1] = WTr<WT>::from_f32(o1);
        }
    } else {
        const int rv = r - qdim - kdim, g = rv / Dh, dd = rv % Dh;
        WT* dst = kv_addr<WT>(kv.v, kv.page_table, pos, g, Hk, Dh) + dd;
        dst[0] = WTr<WT>::from_f32(a); dst[1] = WTr<WT>::from_f32(b);
    }
}

// ------------------------------------------------------------------------------------------------ decode attention
// grid = Hk * nsplit blocks of 256 threads.  Block (g, s) attends the n_rep q heads of kv head g over its slice of
// the T cached tokens and writes an un-normalised partial {m, l, o[Dh]} per q head (flash-decoding split).
// LPT lanes cover one token's Dh elements with 16-B loads; a wave covers 64/LPT tokens per iteration.
template <typename WT, int DH, int NREP>
__global__ __launch_bounds__(256) void k_attn_decode(const float* __restrict__ q, KVView kv,
                                                     const SeqState* __restrict__ state, float* __restrict__ part,
                                                     int Hk, int nsplit) {
    constexpr int EPL = WTr<WT>::EPL;
    constexpr int LPT = DH / EPL;   // lanes per token
    constexpr int TPW = 64 / LPT;   // tokens per wave iteration
    using vec = typename WTr<WT>::vec;
    const int g = blockIdx.x / nsplit, s = blockIdx.x % nsplit;
    const int T = state->pos + 1;  // the current token's K/V were appended by k_qkv
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPT, tl = lane / LPT;
    int chunk = (T + nsplit - 1) / nsplit;
    chunk = (chunk + TPW * 4 - 1) / (TPW * 4) * (TPW * 4);
    const int t_lo = s * chunk, t_hi = min(T, t_lo + chunk);
    const float scale = 1.0f / sqrtf((float)DH);

    float qr[NREP][EPL], o[NREP][EPL], m[NREP], l[NREP];
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        const float* qp = q + (size_t)(g * NREP + r) * DH + sub * EPL;
#pragma unroll
        for (int i = 0; i < EPL; ++i) { qr[r][i] = qp[i]; o[r][i] = 0.f; }
        m[r] = -1e30f; l[r] = 0.f;
    }
    for (int t0 = t_lo + wave * TPW; t0 < t_hi; t0 += 4 * TPW) {
        const int t = t0 + tl;
        const bool valid = t < t_hi;
        const int tc = valid ? t : t_lo;
        const vec kvv = *reinterpret_cast<const vec*>(kv_addr<WT>(kv.k, kv.page_table, tc, g, Hk, DH) + sub * EPL);
        const vec vvv = *reinterpret_cast<const vec*>(kv_addr<WT>(kv.v, kv.page_table, tc, g, Hk, DH) + sub * EPL);
        float kf[EPL], vf[EPL];
        WTr<WT>::unpack(kvv, kf);
        WTr<WT>::unpack(vvv, vf);
#pragma unroll
        for (int i = 0; i < EPL; ++i) kf[i] *= scale;  // q . (k^T * scale)  (dual_ar.rs:260)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            float sc = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) sc = fmaf(qr[r][i], kf[i], sc);
#pragma unroll
            for (int msk = LPT / 2; msk >= 1; msk >>= 1) sc += __shfl_xor(sc, msk, 64);
            if (valid) {
                const float mn = fmaxf(m[r], sc);
                const float corr = __expf(m[r] - mn), p = __expf(sc - mn);
                l[r] = l[r] * corr + p;
#pragma unroll
                for (int i = 0; i < EPL; ++i) o[r][i] = o[r][i] * corr + p * vf[i];
                m[r] = mn;
            }
        }
    }
    // merge the TPW token groups of the wave (lanes with equal `sub`)
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
#pragma unroll
        for (int msk = LPT; msk < 64; msk <<= 1) {
            const float m2 = __shfl_xor(m[r], msk, 64), l2 = __shfl_xor(l[r], msk, 64);
            const float mn = fmaxf(m[r], m2);
            const float c1 = __expf(m[r] - mn), c2 = __expf(m2 - mn);
            l[r] = l[r] * c1 + l2 * c2;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
                const float o2 = __shfl_xor(o[r][i], msk, 64);
                o[r][i] = o[r][i] * c1 + o2 * c2;
            }
            m[r] = mn;
        }
    }
    // merge the 4 waves through LDS
    __shared__ float sm[4][NREP][DH + 2];
    if (tl == 0) {
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) sm[wave][r][sub * EPL + i] = o[r][i];
            if (sub == 0) { sm[wave][r][DH] = m[r]; sm[wave][r][DH + 1] = l[r]; }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NREP * DH; e += 256) {
        const int r = e / DH, dd = e % DH;
        float mn = -1e30f;
#pragma unroll
        for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm[w][r][DH]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float c = __expf(sm[w][r][DH] - mn);
            L += sm[w][r][DH + 1] * c;
            O += sm[w][r][dd] * c;
        }
        float* dst = part + ((size_t)(g * NREP + r) * nsplit + s) * (DH + 2);
        dst[dd] = O;
        if (dd == 0) { dst[DH] = mn; dst[DH + 1] = L; }
    }
}

// ------------------------------------------------------------------------------------------------ wo + residual
// Prologue (whole block, result in LDS): either combine the flash-decoding partials of k_attn_decode, or -- FUSED,
// used by the fast decoder whose KV length is <= 8 -- run the whole attention for all heads redundantly per block.
// Body: one wave per output row: x[r] += Wo[r,:] . attn   (dual_ar.rs:383,437)
template <typename WT, int K, int WAVES, bool FUSED>
__global__ __launch_bounds__(WAVES * 64) void k_wo(const float* __restrict__ part, int nsplit, const float* __restrict__ q,
                                                   KVView kv, int fused_T, const WT* __restrict__ W, float* __restrict__ x,
                                                   int H, int Hk, int Dh, int n_rows) {
    using R = Row<WT, K>;
    __shared__ __attribute__((aligned(16))) float attn[K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    typename R::vec wv[R::NCH];
    if (row < n_rows) R::load_w(W + (size_t)row * K, lane, wv);  // weights in flight during the prologue
    const int n_rep = H / Hk;
    for (int e = threadIdx.x; e < K && !FUSED; e += WAVES * 64) {
        const int h = e / Dh, dd = e % Dh;
        {
            const float* p = part + (size_t)h * nsplit * (Dh + 2);
            float mn = -1e30f;
            for (int s = 0; s < nsplit; ++s) mn = fmaxf(mn, p[s * (Dh + 2) + Dh]);
            float L = 0.f, O = 0.f;
            for (int s = 0; s < nsplit; ++s) {
                const float c = __expf(p[s * (Dh + 2) + Dh] - mn);
                L += p[s * (Dh + 2) + Dh + 1] * c;
                O += p[s * (Dh + 2) + dd] * c;
            }
            attn[e] = O / L;
        }
    }
    if (FUSED) {
        // step 1: scores[h][t] for all H x fused_T pairs (one thread each, 16-B K loads); step 2: softmax . V per (h, dd)
        __shared__ float sc[32 * 8];
        const float scale = 1.0f / sqrtf((float)Dh);
        constexpr int EPL = WTr<WT>::EPL;
        for (int e = threadIdx.x; e < H * fused_T; e += WAVES * 64) {
            const int h = e / fused_T, t = e % fused_T, g = h / n_rep;
            const WT* kp = kv_addr<WT>(kv.k, kv.page_table, t, g, Hk, Dh);
            float acc = 0.f;
            for (int i = 0; i < Dh; i += EPL) {
                float kf[EPL];
                WTr<WT>::unpack(*reinterpret_cast<const typename WTr<WT>::vec*>(kp + i), kf);
#pragma unroll
                for (int j = 0; j < EPL; ++j) acc = fmaf(q[h * Dh + i + j], kf[j] * scale, acc);
            }
            sc[h * 8 + t] = acc;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < K; e += WAVES * 64) {
            const int h = e / Dh, dd = e % Dh, g = h / n_rep;
            float mn = -1e30f;
            for (int t = 0; t < fused_T; ++t) mn = fmaxf(mn, sc[h * 8 + t]);
            float L = 0.f, O = 0.f;
            for (int t = 0; t < fused_T; ++t) {
                const float p = __expf(sc[h * 8 + t] - mn);
                L += p;
                O = fmaf(p, WTr<WT>::to_f32(kv_addr<WT>(kv.v, kv.page_table, t, g, Hk, Dh)[dd]), O);
            }
            attn[e] = O / L;
        }
    }
    __syncthreads();
    if (row >= n_rows) return;
    float xr[R::NX];
    R::load_x(attn, lane, xr);
    const float d = wave_sum(R::dot(wv, xr));
    if (lane == 0) x[row] = x[row] + d;
}

// ------------------------------------------------------------------------------------------------ SwiGLU up
// One wave per PAIRS (w1[r], w3[r]) pairs of the row-interleaved W13: act[r] = silu(w1[r].xn) * (w3[r].xn)
template <typename WT, int K, int WAVES, int PAIRS>
__global__ __launch_bounds__(WAVES * 64) void k_ffn_up(const float* __restrict__ x, const float* __restrict__ norm_w,
                                                       float eps, const WT* __restrict__ W13, float* __restrict__ act,
                                                       int inter) {
    using R = Row<WT, K>;
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) * PAIRS;
    if (r0 >= inter) return;
    typename R::vec wv[2 * PAIRS][R::NCH];
#pragma unroll
    for (int p = 0; p < 2 * PAIRS; ++p) R::load_w(W13 + (size_t)(2 * r0 + p) * K, lane, wv[p]);
    float xr[R::NX];
    R::load_x(x, lane, xr);
    R::rmsnorm(xr, norm_w, eps, lane);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        const float a = wave_sum(R::dot(wv[2 * p], xr));
        const float b = wave_sum(R::dot(wv[2 * p + 1], xr));
        if (lane == 0) act[r0 + p] = (a / (1.f + __expf(-a))) * b;  // candle silu = x / (1 + exp(-x))
    }
}

// ------------------------------------------------------------------------------------------------ down + residual
// One wave per row of W2 (K = inter): x[r] += W2[r,:] . act.  act (K floats) staged once per block in LDS.
template <typename WT, int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_ffn_down(const float* __restrict__ act, const WT* __restrict__ W2,
                                                         float* __restrict__ x, int n_rows) {
    using R = Row<WT, K>;
    __shared__ __attribute__((aligned(16))) float sa[K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    typename R::vec wv[R::NCH];
    if (row < n_rows) R::load_w(W2 + (size_t)row * K, lane, wv);
    for (int e = threadIdx.x * 4; e < K; e += WAVES * 64 * 4)
        *reinterpret_cast<float4*>(&sa[e]) = *reinterpret_cast<const float4*>(&act[e]);
    __syncthreads();
    if (row >= n_rows) return;
    float xr[R::NX];
    R::load_x(sa, lane, xr);
    const float d = wave_sum(R::dot(wv, xr));
    if (lane == 0) x[row] = x[row] + d;
}

// ------------------------------------------------------------------------------------------------ norm + head GEMV
template <typename WT, int K, int WAVES, int ROWS>
__global__ __launch_bounds__(WAVES * 64) void k_head(const float* __restrict__ x, const float* __restrict__ norm_w, float eps,
                                                     const WT* __restrict__ W, int n_rows, float* __restrict__ logits) {
    using R = Row<WT, K>;
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) * ROWS;
    if (r0 >= n_rows) return;
    typename R::vec wv[ROWS][R::NCH];
#pragma unroll
    for (int p = 0; p < ROWS; ++p)
        if (r0 + p < n_rows) R::load_w(W + (size_t)(r0 + p) * K, lane, wv[p]);
    float xr[R::NX];
    R::load_x(x, lane, xr);
    R::rmsnorm(xr, norm_w, eps, lane);
#pragma unroll
    for (int p = 0; p < ROWS; ++p) {
        if (r0 + p < n_rows) {
            const float d = wave_sum(R::dot(wv[p], xr));
            if (lane == 0) logits[r0 + p] = d;
        }
    }
}

// ------------------------------------------------------------------------------------------------ embedding
// dual_ar.rs:532-567: x = tok_emb[t0] + sum_c (sem_lo <= t0 <= sem_hi) * cb_emb[c*cb_size + t_{c+1}], summed in order.
template <typename WT>
__device__ __forceinline__ void embed_tokens(const WT* __restrict__ tok_emb, const WT* __restrict__ cb_emb, int dim, int n_cb,
                                             int cb_size, uint32_t sem_lo, uint32_t sem_hi, const uint32_t* toks, int stride,
                                             float* __restrict__ x, int tid, int nthreads) {
    const uint32_t sem = toks[0];
    const float m = (sem >= sem_lo && sem <= sem_hi) ? 1.f : 0.f;
    for (int d = tid; d < dim; d += nthreads) {
        float acc = 0.f + WTr<WT>::to_f32(tok_emb[(size_t)sem * dim + d]);
        for (int c = 0; c < n_cb; ++c) {
            const uint32_t code = toks[(size_t)(c + 1) * stride];
            acc += WTr<WT>::to_f32(cb_emb[((size_t)c * cb_size + code) * dim + d]) * m;
        }
        x[d] = acc;
    }
}

template <typename WT>
__global__ void k_embed(const WT* __restrict__ tok_emb, const WT* __restrict__ cb_emb, int dim, int n_cb, int cb_size,
                        const SampleCfg* __restrict__ cfg, const uint32_t* __restrict__ prompt, SeqState* __restrict__ state,
                        float* __restrict__ x) {
    const uint32_t sem_lo = cfg->sem_lo, sem_hi = cfg->sem_hi;
    if (prompt) embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, sem_lo, sem_hi, prompt + state->step, state->prompt_L, x,
                                 threadIdx.x, blockDim.x);
    else embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, sem_lo, sem_hi, state->cur, 1, x, threadIdx.x, blockDim.x);
}

template <typename WT>
__global__ void k_fast_embed(const WT* __restrict__ fast_emb, int dim, const uint32_t* __restrict__ ids, float* __restrict__ out) {
    const uint32_t id = ids[blockIdx.x];
    for (int d = threadIdx.x; d < dim; d += blockDim.x) out[(size_t)blockIdx.x * dim + d] = WTr<WT>::to_f32(fast_emb[(size_t)id * dim + d]);
}

__global__ void k_advance(SeqState* state) {
    state->pos += 1;
    state->step += 1;
}

// ------------------------------------------------------------------------------------------------ sampling
// rand 0.8.5 StdRng == ChaCha12 (rand_chacha 0.3.1): word `n` of the keystream, 64-bit block counter, stream id 0.
__device__ inline uint32_t chacha12_word(const uint32_t* key, unsigned long long n) {
    const unsigned long long ctr = n >> 4;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = s[i];
#define FS_ROTL(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
#define FS_QR(a, b, c, d)                                   \
    w[a] += w[b]; w[d] = FS_ROTL(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = FS_ROTL(w[b] ^ w[c], 12); \
    w[a] += w[b]; w[d] = FS_ROTL(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = FS_ROTL(w[b] ^ w[c], 7);
    for (int r = 0; r < 6; ++r) {
        FS_QR(0, 4, 8, 12) FS_QR(1, 5, 9, 13) FS_QR(2, 6, 10, 14) FS_QR(3, 7, 11, 15)
        FS_QR(0, 5, 10, 15) FS_QR(1, 6, 11, 12) FS_QR(2, 7, 8, 13) FS_QR(3, 4, 9, 14)
    }
#undef FS_QR
#undef FS_ROTL
    uint32_t out = 0;
    const int idx = (int)(n & 15);
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i == idx) out = w[i] + s[i];
    return out;
}

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_MAXN = 4096;  // candidates handled by the sampler (audio range 2037, codebook 1024)

// Block-wide selection of one index from `n` logits held in LDS (already penalised / masked).
//  temp == 0: host-ArgMax rule of candle's LogitsProcessor (max_by(total_cmp)): LAST maximal index wins.
//  temp  > 0: softmax(logits / temp) -> top-k (ties: lower index first) -> top-p -> WeightedIndex draw, evaluated in
//             ascending-index order with the StdRng stream (sampling/mod.rs:51-132).  Identical decision procedure to
//             oracle::LogitsProcessor::sample; the softmax denominator is accumulated in f64 on both sides so that
//             the result does not depend on reduction order.
__device__ int block_sample(float* lg /*LDS [n]*/, int n, const SampleCfg& c, RngState* rng, float* sp /*LDS [SAMPLE_MAXN]*/,
                            int* si /*LDS [SAMPLE_MAXN]*/, double* red /*LDS [SAMPLE_THREADS]*/) {
    const int tid = threadIdx.x;
    __shared__ int s_result;
    if (c.temp == 0.f) {
        float bv = -INFINITY;
        int bi = -1;
        for (int i = tid; i < n; i += SAMPLE_THREADS) {
            const float v = lg[i];
            if (bi < 0 || !(v < bv)) { bv = v; bi = i; }  // ascending i per thread: >= keeps the last
        }
        float* rv = reinterpret_cast<float*>(red);
        int* ri = reinterpret_cast<int*>(red) + SAMPLE_THREADS;
        rv[tid] = bv; ri[tid] = bi;
        __syncthreads();
        for (int s = SAMPLE_THREADS / 2; s >= 1; s >>= 1) {
            if (tid < s) {
                const float v2 = rv[tid + s];
                const int i2 = ri[tid + s];
                const float v1 = rv[tid];
                const int i1 = ri[tid];
                const bool take2 = (i1 < 0) || (i2 >= 0 && (v2 > v1 || (v2 == v1 && i2 > i1)));
                if (take2) { rv[tid] = v2; ri[tid] = i2; }
            }
            __syncthreads();
        }
        const int res = ri[0];
        __syncthreads();
        return res;
    }
    // softmax(logits * (1/temp))
    const float inv_t = (float)(1.0 / (double)c.temp);
    float mx = -INFINITY;
    for (int i = tid; i < n; i += SAMPLE_THREADS) { const float v = lg[i] * inv_t; lg[i] = v; mx = fmaxf(mx, v); }
    float* rv = reinterpret_cast<float*>(red);
    rv[tid] = mx;
    __syncthreads();
    for (int s = SAMPLE_THREADS / 2; s >= 1; s >>= 1) { if (tid < s) rv[tid] = fmaxf(rv[tid], rv[tid + s]); __syncthreads(); }
    mx = rv[0];
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < n; i += SAMPLE_THREADS) { const float e = expf(lg[i] - mx); lg[i] = e; part += (double)e; }
    red[tid] = part;
    __syncthreads();
    for (int s = SAMPLE_THREADS / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float denom = (float)red[0];
    __syncthreads();
    // sort keys: (prob desc, index asc) via bitonic sort over the next power of two
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += SAMPLE_THREADS) {
        if (i < n) { sp[i] = lg[i] / denom; si[i] = i; } else { sp[i] = -1.f; si[i] = 0x7FFFFFFF; }
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += SAMPLE_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const float p1 = sp[i], p2 = sp[ixj];
                    const int i1 = si[i], i2 = si[ixj];
                    const bool before = (p1 > p2) || (p1 == p2 && i1 < i2);  // element i sorts before element ixj
                    if (before != up) { sp[i] = p2; sp[ixj] = p1; si[i] = i2; si[ixj] = i1; }
                }
            }
            __syncthreads();
        }
    }
    // serial tail on one lane: few elements survive top-k / top-p in practice; order and f32 rounding identical to
    // the oracle restatement.
    if (tid == 0) {
        const bool use_k = c.top_k > 0 && c.top_k < n;
        const int kk = use_k ? c.top_k : n;
        // sum over the kept set in ascending index order == f32 sum over all n with zeros elsewhere
        // mark kept probabilities back into lg[] by index (0 elsewhere)
        for (int i = 0; i < n; ++i) lg[i] = 0.f;
        bool do_topp = true;
        if (use_k) {
            // sum_p in ascending index order: need the kept set sorted by index -> accumulate through lg[]
            for (int r = 0; r < kk; ++r) lg[si[r]] = sp[r];
            float sum_p = 0.f;
            for (int i = 0; i < n; ++i) if (lg[i] != 0.f) sum_p += lg[i];
            do_topp = !(c.top_p <= 0.f || c.top_p >= sum_p);
        } else {
            for (int r = 0; r < kk; ++r) lg[si[r]] = sp[r];
        }
        if (do_topp) {  // zero every prob once the running cumsum (descending order) reached top_p
            float cumsum = 0.f;
            for (int r = 0; r < kk; ++r) {
                if (cumsum >= c.top_p) lg[si[r]] = 0.f;
                cumsum += lg[si[r]];
            }
        }
        // WeightedIndex::new + sample over lg[0..n) (ascending index; within top-k: position among kept entries)
        float total = 0.f;
        for (int i = 0; i < n; ++i) if (lg[i] != 0.f) total += lg[i];
        int res = 0;
        if (total > 0.f) {
            const float max_rand = __uint_as_float((0xFFFFFFFFu >> 9) | (127u << 23)) - 1.0f;
            float scale = total;
            while (scale * max_rand + 0.f >= total) scale = __uint_as_float(__float_as_uint(scale) - 1u);
            const uint32_t w = chacha12_word(rng->key, rng->consumed);
            rng->consumed += 1;
            const float chosen = (__uint_as_float((w >> 9) | (127u << 23)) - 1.0f) * scale + 0.f;
            // first kept item whose cumulative weight (exclusive prefix) is > chosen; zero-weight items are skipped by
            // construction (cum does not move), matching partition_point over the full cumulative array only when the
            // chosen item has non-zero weight -- which WeightedIndex guarantees as chosen < total.
            float cum = 0.f;
            int last_nz = 0;
            res = -1;
            for (int i = 0; i < n; ++i) {
                if (lg[i] == 0.f) continue;
                last_nz = i;
                cum += lg[i];
                if (cum > chosen) { res = i; break; }
            }
            if (res < 0) res = last_nz;
        }
        s_result = res;
    }
    __syncthreads();
    const int res = s_result;
    __syncthreads();
    return res;
}

template <typename WT>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_slow(const float* __restrict__ logits, int n,
                                                                const SampleCfg* __restrict__ cp, RngState* rng, SeqState* __restrict__ state,
                                                                const float* __restrict__ x, float* __restrict__ xf, int dim) {
    __shared__ float lg[SAMPLE_MAXN];
    __shared__ float sp[SAMPLE_MAXN];
    __shared__ int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    const int tid = threadIdx.x;
    const SampleCfg c = *cp;
    for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = logits[i];
    for (int i = tid; i < dim; i += SAMPLE_THREADS) xf[i] = x[i];  // hidden_states -> fast decoder input (:149)
    __syncthreads();
    if (c.ignore_eos && tid == 0) lg[0] = -INFINITY;
    __syncthreads();
    const int idx = block_sample(lg, n, c, rng, sp, si, red);
    if (tid == 0) {
        uint32_t tok = (uint32_t)idx + c.im_end_id;  // rescale_semantic_tokens (utils.rs:45-46)
        if (state->done) tok = c.im_end_id;  // generator already terminated (single_batch.rs:86-88): stay terminated
        state->cur[0] = tok;
        if (tok == c.im_end_id && state->done == 0) state->done = 1;  // 1 = terminated by THIS frame, 2 = earlier
    }
}

template <typename WT>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_fast(const float* __restrict__ logits, int cb, int n_cb, int cb_size,
                                                                const SampleCfg* __restrict__ cp, RngState* rng, RepPenState rp,
                                                                SeqState* __restrict__ state, const WT* __restrict__ fast_emb,
                                                                float* __restrict__ xf, const WT* __restrict__ tok_emb,
                                                                const WT* __restrict__ cb_emb, float* __restrict__ x, int dim,
                                                                uint32_t* __restrict__ out_codes, int out_cap) {
    __shared__ float lg[SAMPLE_MAXN];
    __shared__ float sp[SAMPLE_MAXN];
    __shared__ int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    const int tid = threadIdx.x;
    const int n = cb_size;
    const SampleCfg c = *cp;
    const bool eos = state->cur[0] == c.im_end_id;  // single_batch.rs:153-156: push 0, skip the fast step
    float* mask = rp.mask + (size_t)cb * cb_size;
    if (!eos) {
        if (state->have_prev && tid == 0) {  // SingleBatchedRepPenProcessor::apply (rep_pen.rs:37-65)
            const int last = (int)state->prev[cb + 1];
            uint8_t* seen = rp.seen + (size_t)cb * cb_size;
            int* ring = rp.ring + cb * 17;
            int* meta = rp.ring_meta + cb * 2;  // head (index of front), len
            seen[last] = 1;
            mask[last] = c.rep_pen;
            int head = (meta[0] + 16) % 17, len = meta[1] + 1;  // push_front
            ring[head] = last;
            if (len > 16) {
                const int back = (head + len - 1) % 17;
                const int dropped = ring[back];
                len -= 1;
                if (seen[dropped]) { seen[dropped] = 0; mask[dropped] = 1.0f; }
            }
            meta[0] = head; meta[1] = len;
        }
        __syncthreads();
        const bool pen = state->have_prev != 0;
        for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = pen ? logits[i] / mask[i] : logits[i];
        __syncthreads();
    }
    int code = 0;
    if (!eos) code = block_sample(lg, n, c, rng, sp, si, red);
    if (tid == 0) state->cur[cb + 1] = (uint32_t)code;
    if (cb != n_cb - 1) {
        if (!eos)
            for (int d = tid; d < dim; d += SAMPLE_THREADS) xf[d] = WTr<WT>::to_f32(fast_emb[(size_t)code * dim + d]);
        return;
    }
    // ---- end of frame (single_batch.rs:185-210 + generate_blocking :250,264-266)
    __syncthreads();
    __shared__ uint32_t cur[16];
    if (tid <= n_cb) cur[tid] = (tid == n_cb) ? (uint32_t)code : state->cur[tid];
    __syncthreads();
    if (tid == 0 && state->done != 2) {
        const int frame = state->frame;
        if (state->done == 1) state->done = 2;  // replays after termination leave pos / outputs untouched
        if (frame == 0 || cur[0] != c.im_end_id) {
            const int o = state->n_out;
            if (o < out_cap)
                for (int cc = 0; cc < n_cb; ++cc) out_codes[(size_t)cc * out_cap + o] = cur[cc + 1];
            state->n_out = o + 1;
        }
        for (int i = 0; i <= n_cb; ++i) state->prev[i] = cur[i];
        state->have_prev = 1;
        state->pos += 1;
        state->frame = frame + 1;
    }
    // next slow input: embed([slow, c0..c7]) (dual_ar.rs:532-567)
    embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, c.sem_lo, c.sem_hi, cur, 1, x, tid, SAMPLE_THREADS);
}

__global__ void k_reppen_reset(RepPenState rp, int n_cb, int cb_size) {
    const int n = n_cb * cb_size;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { rp.mask[i] = 1.0f; rp.seen[i] = 0; }
    if (blockIdx.x == 0 && threadIdx.x < n_cb) { rp.ring_meta[threadIdx.x * 2] = 0; rp.ring_meta[threadIdx.x * 2 + 1] = 0; }
}

// ------------------------------------------------------------------------------------------------ weight fill / convert
template <typename WT>
__global__ void k_synth_fill(WT* __restrict__ dst, uint64_t key, long long n_rows, long long n_cols, int row_mul, int row_off,
                             float mean, float scale, int round_bf16) {
    const long long n = n_rows * n_cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = synth_elem(key, (uint64_t)i, mean, scale);
        if (round_bf16) v = bf16_bits_to_f32(WTr<bf16_t>::from_f32(v));
        const long long r = i / n_cols, cidx = i % n_cols;
        dst[(r * row_mul + row_off) * n_cols + cidx] = WTr<WT>::from_f32(v);
    }
}
template <typename WT>
__global__ void k_convert_rows(WT* __restrict__ dst, const float* __restrict__ src, long long n_rows, long long n_cols,
                               int row_mul, int row_off) {
    const long long n = n_rows * n_cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / n_cols, cidx = i % n_cols;
        dst[(r * row_mul + row_off) * n_cols + cidx] = WTr<WT>::from_f32(src[i]);
    }
}

// ================================================================================================ launchers
#define FS_LAUNCH_CHECK() FS_HIP(hipGetLastError())

template <typename F>
static void dispatch_k(int K, F&& f) {
    switch (K) {
        case 128: f(std::integral_constant<int, 128>()); break;
        case 256: f(std::integral_constant<int, 256>()); break;
        case 1024: f(std::integral_constant<int, 1024>()); break;
        case 4096: f(std::integral_constant<int, 4096>()); break;
        default: throw Error("unsupported GEMV width K=" + std::to_string(K) + " (supported: 128, 256, 1024, 4096)");
    }
}

template <typename WT>
void LmKernels<WT>::qkv(const ModelDims& d, const float* x, const LayerW& w, const float* cos_t, const float* sin_t,
                        const SeqState* state, int pos_static, int rope_static, float* q_out, KVView kv, hipStream_t st) {
    constexpr int WAVES = 2;
    const int n_pairs = (d.H + 2 * d.Hk) * d.Dh / 2;
    const int grid = (n_pairs + WAVES - 1) / WAVES;
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        hipLaunchKernelGGL((k_qkv<WT, K, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.attn_norm, d.eps,
                           (const WT*)w.wqkv, cos_t, sin_t, state, pos_static, rope_static, q_out, kv, d.H, d.Hk, d.Dh);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::attn_decode(const ModelDims& d, const float* q, KVView kv, const SeqState* state, float* part, int nsplit,
                                hipStream_t st) {
    const int grid = d.Hk * nsplit;
    if (d.Dh == 64 && d.n_rep == 8)
        hipLaunchKernelGGL((k_attn_decode<WT, 64, 8>), dim3(grid), dim3(256), 0, st, q, kv, state, part, d.Hk, nsplit);
    else if (d.Dh == 32 && d.n_rep == 2)
        hipLaunchKernelGGL((k_attn_decode<WT, 32, 2>), dim3(grid), dim3(256), 0, st, q, kv, state, part, d.Hk, nsplit);
    else if (d.Dh == 64 && d.n_rep == 2)
        hipLaunchKernelGGL((k_attn_decode<WT, 64, 2>), dim3(grid), dim3(256), 0, st, q, kv, state, part, d.Hk, nsplit);
    else
        throw Error("unsupported attention geometry (head_dim, n_rep) = (" + std::to_string(d.Dh) + ", " +
                    std::to_string(d.n_rep) + ")");
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::wo(const ModelDims& d, const float* part, int nsplit, const float* q, KVView kv, int fused_T,
                       const LayerW& w, float* x, hipStream_t st) {
    constexpr int WAVES = 4;
    const int grid = (d.dim + WAVES - 1) / WAVES;
    FS_REQUIRE(fused_T <= 8, "fused attention supports at most 8 cached tokens");
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        if (fused_T > 0)
            hipLaunchKernelGGL((k_wo<WT, K, WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, st, part, nsplit, q, kv, fused_T,
                               (const WT*)w.wo, x, d.H, d.Hk, d.Dh, d.dim);
        else
            hipLaunchKernelGGL((k_wo<WT, K, WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, st, part, nsplit, q, kv, 0,
                               (const WT*)w.wo, x, d.H, d.Hk, d.Dh, d.dim);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::ffn_up(const ModelDims& d, const float* x, const LayerW& w, float* act, hipStream_t st) {
    constexpr int WAVES = 4, PAIRS = 2;
    const int grid = (d.inter + WAVES * PAIRS - 1) / (WAVES * PAIRS);
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        hipLaunchKernelGGL((k_ffn_up<WT, K, WAVES, PAIRS>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.ffn_norm, d.eps,
                           (const WT*)w.w13, act, d.inter);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::ffn_down(const ModelDims& d, const float* act, const LayerW& w, float* x, hipStream_t st) {
    constexpr int WAVES = 4;
    const int grid = (d.dim + WAVES - 1) / WAVES;
    dispatch_k(d.inter, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        hipLaunchKernelGGL((k_ffn_down<WT, K, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, act, (const WT*)w.w2, x, d.dim);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::head(const ModelDims& d, const float* x, const float* norm_w, const void* W, int n_rows, float* logits,
                         hipStream_t st) {
    constexpr int WAVES = 4, ROWS = 2;
    const int grid = (n_rows + WAVES * ROWS - 1) / (WAVES * ROWS);
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        hipLaunchKernelGGL((k_head<WT, K, WAVES, ROWS>), dim3(grid), dim3(WAVES * 64), 0, st, x, norm_w, d.eps, (const WT*)W,
                           n_rows, logits);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::embed(const ModelDims& d, const void* tok_emb, const void* cb_emb, int n_cb, int cb_size,
                          const SampleCfg* cfg, const uint32_t* prompt, SeqState* state, float* x, hipStream_t st) {
    hipLaunchKernelGGL((k_embed<WT>), dim3(1), dim3(256), 0, st, (const WT*)tok_emb, (const WT*)cb_emb, d.dim, n_cb, cb_size,
                       cfg, prompt, state, x);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::fast_embed(const ModelDims& d, const void* fast_emb, const uint32_t* ids, int n, float* out,
                               hipStream_t st) {
    hipLaunchKernelGGL((k_fast_embed<WT>), dim3(n), dim3(256), 0, st, (const WT*)fast_emb, d.dim, ids, out);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void SampleKernels<WT>::sample_slow(const ModelDims& d, const float* logits, int n, const SampleCfg* c, RngState* rng,
                                    SeqState* state, const float* x, float* xf, hipStream_t st) {
    FS_REQUIRE(n <= SAMPLE_MAXN, "audio-range vocabulary larger than the sampler capacity");
    hipLaunchKernelGGL((k_sample_slow<WT>), dim3(1), dim3(SAMPLE_THREADS), 0, st, logits, n, c, rng, state, x, xf, d.dim);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void SampleKernels<WT>::sample_fast(const ModelDims& d, const float* logits, int cb, int n_cb, int cb_size, const SampleCfg* c,
                                    RngState* rng, RepPenState rp, SeqState* state, const void* fast_emb, float* xf,
                                    const void* tok_emb, const void* cb_emb, float* x, uint32_t* out_codes, int out_cap,
                                    hipStream_t st) {
    FS_REQUIRE(cb_size <= SAMPLE_MAXN, "codebook larger than the sampler capacity");
    hipLaunchKernelGGL((k_sample_fast<WT>), dim3(1), dim3(SAMPLE_THREADS), 0, st, logits, cb, n_cb, cb_size, c, rng, rp, state,
                       (const WT*)fast_emb, xf, (const WT*)tok_emb, (const WT*)cb_emb, x, d.dim, out_codes, out_cap);
    FS_LAUNCH_CHECK();
}

void launch_advance(SeqState* state, hipStream_t st) {
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, state);
    FS_LAUNCH_CHECK();
}
void launch_reppen_reset(RepPenState rp, int n_cb, int cb_size, hipStream_t st) {
    hipLaunchKernelGGL(k_reppen_reset, dim3(8), dim3(256), 0, st, rp, n_cb, cb_size);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void launch_synth_fill(WT* dst, uint64_t key, int64_t n_rows, int64_t n_cols, int row_mul, int row_off, float mean, float scale,
                       int round_bf16, hipStream_t st) {
    const long long n = n_rows * n_cols;
    const int grid = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL((k_synth_fill<WT>), dim3(grid), dim3(256), 0, st, dst, key, (long long)n_rows, (long long)n_cols, row_mul,
                       row_off, mean, scale, round_bf16);
    FS_LAUNCH_CHECK();
}
template <typename WT>
void launch_convert_rows(WT* dst, const float* src, int64_t n_rows, int64_t n_cols, int row_mul, int row_off, hipStream_t st) {
    const long long n = n_rows * n_cols;
    const int grid = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL((k_convert_rows<WT>), dim3(grid), dim3(256), 0, st, dst, src, (long long)n_rows, (long long)n_cols,
                       row_mul, row_off);
    FS_LAUNCH_CHECK();
}

template struct LmKernels<bf16_t>;
template struct LmKernels<float>;
template struct SampleKernels<bf16_t>;
template struct SampleKernels<float>;
template void launch_synth_fill<bf16_t>(bf16_t*, uint64_t, int64_t, int64_t, int, int, float, float, int, hipStream_t);
template void launch_synth_fill<float>(float*, uint64_t, int64_t, int64_t, int, int, float, float, int, hipStream_t);
template void launch_convert_rows<bf16_t>(bf16_t*, const float*, int64_t, int64_t, int, int, hipStream_t);
template void launch_convert_rows<float>(float*, const float*, int64_t, int64_t, int, int, hipStream_t);

}  // namespace fs
