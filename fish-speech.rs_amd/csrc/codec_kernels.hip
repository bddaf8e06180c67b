// Firefly-GAN-VQ vocoder kernels for gfx950 (FireflyCodec::decode, fish_speech_core/lib/codec/*.rs).
//
// Precision: f32 end to end (the reference server runs the codec in f32, server/lib/utils/load.rs:161-164, and the
// acceptance bound is PCM within 1e-4 RMS of the f32 oracle), so every convolution is an exact-f32 FMA chain.
// Layout: activations (C, T) row-major per batch item (the reference's (b, C, T)); weights are re-laid at load time to
// [Cin/groups][K][Cout] (output channel fastest) so that a block stages a contiguous [ICH][K][OT] weight tile.
// Every conv of the 1.4+/1.5 codec is CAUSAL (left zero pad (k-1)*dil, utils/mod.rs:53-62) and every transposed conv
// trims k - stride samples on the right (utils/mod.rs:110-122); SiLU pre-activations, bias, GELU, gamma/residual and the
// final tanh are fused into the producing / consuming convolution.
#include <hip/hip_runtime.h>
#include <set>

#include "codec_kernels.h"
#include "fs_common.h"
#include "fs_synth.h"

namespace fs {

__device__ __forceinline__ float dsilu(float x) { return x / (1.f + __expf(-x)); }
// candle Tensor::gelu == tanh approximation (convnext.rs:115)
__device__ __forceinline__ float dgelu(float x) {
    return 0.5f * x * (1.f + tanhf(0.7978845608028654f * x * (1.f + 0.044715f * x * x)));
}

// ------------------------------------------------------------------------------------------------ FSQ lookup + project_out
// quantizer.rs:135-146 + grouped_residual_fsq.rs:95-114,175-185 + fsq.rs:119-159: per group g, token t:
//   code[k] = ((idx / basis_k) % levels_k - hw_k) / hw_k, levels (8,5,5,5), basis (1,8,40,200), hw (4,2,2,2)
//   z[g*dg + o][t] = sum_k code[k] * Wout_g[o][k] + b_g[o]
// `rows`: the source row of (batch b', group slot g') is r = g'*B + b' of the (B*G, T) index matrix -- the reference's raw
// reshape (b, g, t) -> (g, b, t, 1) (quantizer.rs:138-143), which is the identity only for B == 1.
__global__ void k_fsq_project(const uint32_t* __restrict__ codes, int B, int G, int T, const float* __restrict__ pw /*[G][dg][4]*/,
                              const float* __restrict__ pb /*[G][dg]*/, int dg, float* __restrict__ z /*[B][G*dg][T]*/) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int r = g * B + b;
    const uint32_t idx = codes[(size_t)r * T + t];
    float code[4];
    const int levels[4] = {8, 5, 5, 5}, basis[4] = {1, 8, 40, 200};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float hw = (float)(levels[k] / 2);
        code[k] = ((float)((idx / basis[k]) % levels[k]) - hw) / hw;
    }
    for (int o = 0; o < dg; ++o) {
        const float* w = pw + ((size_t)g * dg + o) * 4;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += code[k] * w[k];
        z[((size_t)b * G * dg + g * dg + o) * T + t] = acc + pb[g * dg + o];
    }
}

// ------------------------------------------------------------------------------------------------ causal conv1d
// y[o][t] = epi( b[o] + sum_i sum_k W[o][i][k] * pre(x[i][t + k*dil - (K-1)*dil]) ),  stride 1.
// Block = 256 threads computes OT output channels x TT = 32*TPT time steps; thread (ty = tid/32, tx = tid%32) owns
// channels ty*CPT .. +CPT and times tx + 32*j (lanes of a wave read consecutive LDS words: conflict-free, weights are
// wave-broadcast reads).  Input channels are staged ICH at a time: x tile [ICH][TT + halo], weight tile [ICH][K][OT].
enum { CEPI_NONE = 0, CEPI_GELU = 1, CEPI_GAMMA_RES = 2, CEPI_RES = 3, CEPI_TANH = 4 };

template <int CPT, int TPT, int ICH>
__global__ __launch_bounds__(256) void k_conv1d(const float* __restrict__ x, int Cin, int T, const float* __restrict__ wt /*[Cin][K][Cout]*/,
                                                const float* __restrict__ bias, int Cout, int K, int dil, int pre_silu, int epi,
                                                const float* __restrict__ res, const float* __restrict__ gamma, float* __restrict__ y, int ps,
                                                const float* __restrict__ ctx /*streaming: [Cin][CODEC_CTX_F32] left context, or null*/) {
    constexpr int OT = 8 * CPT, TT = 32 * TPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = (K - 1) * dil;
    const int XS = TT + halo;
    float* xs = smem;                 // [ICH][XS]
    float* ws = smem + ICH * XS;      // [ICH][K][OT]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * TT, o0 = blockIdx.y * OT;
    const size_t boff_in = (size_t)blockIdx.z * Cin * T, boff_out = (size_t)blockIdx.z * Cout * T;
    float acc[CPT][TPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int j = 0; j < TPT; ++j) acc[c][j] = 0.f;
    for (int i0 = 0; i0 < Cin; i0 += ICH) {
        const int nic = min(ICH, Cin - i0);
        for (int e = threadIdx.x; e < nic * XS; e += 256) {
            const int i = e / XS, tl = e % XS;
            const int t = t0 + tl - halo;
            float v = (t >= 0 && t < T) ? x[boff_in + (size_t)(i0 + i) * T + t] : 0.f;
            if (ctx && t < 0 && t >= -CODEC_CTX_F32) v = ctx[(size_t)(i0 + i) * CODEC_CTX_F32 + CODEC_CTX_F32 + t];
            if (pre_silu) v = dsilu(v);
            xs[i * XS + tl] = v;
        }
        for (int e = threadIdx.x; e < nic * K * OT; e += 256) {
            const int o = e % OT, ik = e / OT;
            ws[e] = (o0 + o < Cout) ? wt[((size_t)(i0 + ik / K) * K + ik % K) * Cout + o0 + o] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < nic; ++i)
            for (int k = 0; k < K; ++k) {
                float xv[TPT], wv[CPT];
                const float* xp = xs + i * XS + tx + k * dil;
#pragma unroll
                for (int j = 0; j < TPT; ++j) xv[j] = xp[32 * j];
                const float* wp = ws + (i * K + k) * OT + ty * CPT;
#pragma unroll
                for (int c = 0; c < CPT; ++c) wv[c] = wp[c];
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int j = 0; j < TPT; ++j) acc[c][j] = fmaf(wv[c], xv[j], acc[c][j]);
            }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int o = o0 + ty * CPT + c;
        if (o >= Cout) continue;
        const float b = bias[o / ps];
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int t = t0 + tx + 32 * j;
            if (t >= T) continue;
            float v = acc[c][j] + b;
            // ps > 1: polyphase transposed conv -- GEMM row o = channel * ps + phase writes y[channel][t * ps + phase]
            const size_t oi = boff_out + (size_t)(o / ps) * T * ps + (size_t)t * ps + o % ps;
            if (epi == CEPI_GELU) v = dgelu(v);
            else if (epi == CEPI_GAMMA_RES) v = res[oi] + gamma[o] * v;
            else if (epi == CEPI_RES) v = res[oi] + v;
            else if (epi == CEPI_TANH) v = tanhf(v);
            y[oi] = v;
        }
    }
}

// ---- one output channel (HiFi-GAN conv_post: 16 -> 1, k = 13, tanh, over 2048 T samples).  k_conv1d's block shape (8 output channels
// x 256 samples) leaves 7 of 8 threads idle there (73 us for 34 MB of input).  Here a block stages silu(x) of all input channels for 512
// samples once and every thread produces four consecutive ones from 20 window values held in registers; same fmaf chain (channel-major, then taps), same silu / tanh as k_conv1d, so the
// result is bit-identical to it.  ctx: streaming left context [Cin][CODEC_CTX_F32] or null.
template <int CIN>
__global__ __launch_bounds__(128) void k_conv1d_one(const float* __restrict__ x, int T, const float* __restrict__ wt /*[CIN][K][1]*/,
                                                    const float* __restrict__ bias, int K, int pre_silu, int epi, float* __restrict__ y,
                                                    const float* __restrict__ ctx) {
    constexpr int TT = 512, HMAX = CODEC_CTX_F32;  // 128 threads x 4 consecutive samples
    __shared__ __attribute__((aligned(16))) float xs[CIN][TT + HMAX];
    __shared__ float ws[CIN * (HMAX + 1)];
    const int halo = K - 1, t0 = blockIdx.x * TT;
    const size_t boff = (size_t)blockIdx.z * CIN * T;
    // window position tl <-> sample t0 + tl - HMAX (the window always starts HMAX samples early, so that a thread's 4 + 16 values sit at
    // 16-byte aligned offsets; taps index it at HMAX - halo + k)
    // (row-wise, no index division; the CIN loads of a window position are in flight together)
    for (int tl = threadIdx.x; tl < TT + HMAX; tl += 128) {
        const int t = t0 + tl - HMAX, tc = min(max(t, 0), T - 1);
        const bool inside = t >= 0 && t < T, from_ctx = ctx && t < 0 && t >= -CODEC_CTX_F32;
        float v[CIN];
#pragma unroll
        for (int i = 0; i < CIN; ++i) v[i] = x[boff + (size_t)i * T + tc];
#pragma unroll
        for (int i = 0; i < CIN; ++i) {
            float u = inside ? v[i] : 0.f;
            if (from_ctx) u = ctx[(size_t)i * CODEC_CTX_F32 + CODEC_CTX_F32 + t];
            xs[i][tl] = pre_silu ? dsilu(u) : u;
        }
    }
    for (int e = threadIdx.x; e < CIN * K; e += 128) ws[e] = wt[e];
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int base = 4 * threadIdx.x;  // window offset of this thread's first sample minus HMAX
    for (int i = 0; i < CIN; ++i) {
        float v[4 + HMAX];
#pragma unroll
        for (int q = 0; q < (4 + HMAX) / 4; ++q) {
            const float4 f = *reinterpret_cast<const float4*>(&xs[i][base + 4 * q]);
            v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
        // sample j of this thread sits at v[HMAX + j]; tap k reads v[HMAX + j - halo + k]  (fully unrolled for K = 13: register indexing)
        if (K == 13) {
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                const float w = ws[i * 13 + k];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, v[HMAX + j - 12 + k], acc[j]);
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const float w = ws[i * K + k];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, xs[i][base + HMAX + j - halo + k], acc[j]);
            }
        }
    }
    const float b = bias[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + base + j;
        if (t >= T) continue;
        float v = acc[j] + b;
        if (epi == CEPI_TANH) v = tanhf(v);
        else if (epi == CEPI_GELU) v = dgelu(v);
        y[(size_t)blockIdx.z * T + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------ causal conv1d on the matrix cores
// Same contract as k_conv1d, for the wide layers (Cout >= 64): the convolution is the GEMM Y[o][t] = sum_{(i,k)} W[o][(i,k)] *
// X[(i,k)][t] with X[(i,k)][t] = pre(x[i][t + k*dil - halo]), evaluated with v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, so
// every product is the exact f32 product of the VALU kernel (only the summation order differs).  Block = 4 waves = 64 output
// channels x 128 time steps; wave w owns channels (w&1)*32.. and two 32-step time tiles (w>>1)*64 + {0, 32} (32 accumulator
// registers).  Input channels are staged 16 at a time into LDS exactly like k_conv1d (x window [16][128 + halo], weights
// [16*K][64]); one MFMA consumes TWO reduction items: lanes 0..31 feed channel 2p, lanes 32..63 channel 2p+1 of the same tap.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// OT = output channels per block: 64 (wave = 32 channels x 64 samples, two MFMA tiles) or 32 (thin late stages with 16..63
// channels: wave = 32 channels x 32 samples, one tile; rows >= Cout are zero-filled and never stored)
// TT = samples per block: 128, or 256 where the grid stays large (twice the MFMA work per staged weight tile and per barrier).
template <int ICH, int OT, int TT>
__global__ __launch_bounds__(256) void k_conv1d_mfma(const float* __restrict__ x, int Cin, int T, const float* __restrict__ wt /*[Cin][K][Cout]*/,
                                                     const float* __restrict__ bias, int Cout, int K, int dil, int pre_silu, int epi,
                                                     const float* __restrict__ res, const float* __restrict__ gamma, float* __restrict__ y, int ps) {
    constexpr int WT_ = OT == 64 ? TT / 2 : TT / 4;  // samples per wave
    constexpr int NT = WT_ / 32;                     // MFMA tiles (32 samples each) per wave
    constexpr int NPX = TT / 128;                     // window elements per thread per row (XS = TT + halo <= 256 * NPX, checked by the launcher)
    static_assert((OT == 64 || OT == 32) && (TT == 128 || TT == 256), "block shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = (K - 1) * dil;
    const int XS = TT + halo;
    float* xs = smem;                       // [ICH][XS]
    float* ws = smem + ((ICH * XS + 3) & ~3);  // [ICH*K][OT], 16-byte aligned
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int ob = OT == 64 ? (wave & 1) * 32 : 0, tb = OT == 64 ? (wave >> 1) * WT_ : wave * WT_;
    const int t0 = blockIdx.x * TT, o0 = blockIdx.y * OT;
    const size_t boff_in = (size_t)blockIdx.z * Cin * T, boff_out = (size_t)blockIdx.z * Cout * T;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    static_assert(ICH % 2 == 0, "channel pairs");
    for (int i0 = 0; i0 < Cin; i0 += ICH) {
        const int nic = min(ICH, Cin - i0);
        // staging without integer division (it dominated the first version of this kernel: 2 runtime div/mod per element):
        // x window: one row per input channel, lanes along time; weights: the tile's rows (i, k) are CONSECUTIVE rows of the
        // re-laid [Cin][K][Cout] tensor ((i0 + i) * K + k == i0 * K + row), read as float4
        // (all loads of a stage are issued before the first LDS store: 16 x-window rows + up to 13 weight float4 per thread)
        {
            float xv[NPX][ICH];  // window element tl + 256 * q of every row
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = (int)threadIdx.x + 256 * q, t = t0 + tl - halo;
#pragma unroll
                for (int i = 0; i < ICH; ++i)
                    xv[q][i] = (tl < XS && i < nic && t >= 0 && t < T) ? x[boff_in + (size_t)(i0 + i) * T + t] : 0.f;
            }
            const int rows = ICH * K, rows_valid = nic * K;
            if (o0 + OT <= Cout && (Cout & 3) == 0) {
                constexpr int Q4 = OT / 4;                          // float4 per weight row of the tile
                constexpr int NW4 = (16 * 13 * Q4 + 255) / 256;     // ceil(ICH * K * Q4 / 256) for K <= 13
                float4 wv[NW4];
#pragma unroll
                for (int j = 0; j < NW4; ++j) {
                    const int e = j * 256 + (int)threadIdx.x, rr = e / Q4, q = e % Q4;
                    wv[j] = (rr < rows_valid) ? *reinterpret_cast<const float4*>(wt + ((size_t)i0 * K + rr) * Cout + o0 + q * 4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < NW4; ++j) {
                    const int e = j * 256 + (int)threadIdx.x, rr = e / Q4, q = e % Q4;
                    if (rr < rows) *reinterpret_cast<float4*>(ws + rr * OT + q * 4) = wv[j];
                }
            } else {
                for (int e = threadIdx.x; e < rows * OT; e += 256) {
                    const int rr = e / OT, o = e % OT;
                    ws[e] = (rr < rows_valid && o0 + o < Cout) ? wt[((size_t)i0 * K + rr) * Cout + o0 + o] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = (int)threadIdx.x + 256 * q;
                if (tl < XS) {
#pragma unroll
                    for (int i = 0; i < ICH; ++i) xs[i * XS + tl] = pre_silu ? dsilu(xv[q][i]) : xv[q][i];
                }
            }
        }
        __syncthreads();
        // reduction items in (k outer, channel-pair inner) order: lanes 0..31 take channel 2p, lanes 32..63 channel 2p+1 --
        // every address below is a running sum, the 8 pairs of one tap are issued back to back
        const float* wl = ws + h * K * OT + ob + c;
        const float* xl = xs + h * XS + tb + c;
        for (int k = 0; k < K; ++k) {
            const float* wk = wl + k * OT;
            const float* xk = xl + k * dil;
#pragma unroll
            for (int p = 0; p < ICH / 2; ++p) {
                const float a = wk[p * 2 * K * OT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xk[p * 2 * XS + 32 * j], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[row][col]: register r of lane (h, c) holds row (r/4)*8 + h*4 + r%4, column c
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + ob + (r >> 2) * 8 + h * 4 + (r & 3);
        if (o >= Cout) continue;
        const float b = bias[o / ps];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + tb + j * 32 + c;
            if (t >= T) continue;
            float v = acc[j][r] + b;
            const size_t oi = boff_out + (size_t)(o / ps) * T * ps + (size_t)t * ps + o % ps;  // (see k_conv1d)
            if (epi == CEPI_GELU) v = dgelu(v);
            else if (epi == CEPI_GAMMA_RES) v = res[oi] + gamma[o] * v;
            else if (epi == CEPI_RES) v = res[oi] + v;
            else if (epi == CEPI_TANH) v = tanhf(v);
            y[oi] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ ConvNeXt: dwconv k7 + LayerNorm
// convnext.rs:110-115: depthwise causal conv (k = 7) then LayerNorm over channels (eps 1e-6, biased variance).
// One block per time step (C <= 1024 threads-strided); output stays (C, T) for the pointwise convs (k = 1) that follow.
__global__ __launch_bounds__(256) void k_dwconv_ln(const float* __restrict__ x, int C, int T, const float* __restrict__ dw /*[C][7]*/,
                                                   const float* __restrict__ db, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, float* __restrict__ y, const float* __restrict__ ctx) {
    __shared__ float red[256];
    __shared__ float vals[1024];
    const int t = blockIdx.x;
    const size_t boff = (size_t)blockIdx.y * C * T;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int tt = t + k - 6;
            if (tt >= 0) a = fmaf(dw[c * 7 + k], x[boff + (size_t)c * T + tt], a);
            else if (ctx) a = fmaf(dw[c * 7 + k], ctx[(size_t)c * CODEC_CTX_F32 + CODEC_CTX_F32 + tt], a);  // streaming: the previous chunk's tail
        }
        a += db[c];
        vals[c] = a;
        s += a;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float mean = red[0] / (float)C;
    __syncthreads();
    float v = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float d = vals[c] - mean; v = fmaf(d, d, v); }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float inv = 1.0f / sqrtf(red[0] / (float)C + 1e-6f);
    for (int c = threadIdx.x; c < C; c += 256) y[boff + (size_t)c * T + t] = (vals[c] - mean) * inv * lnw[c] + lnb[c];
}

// ParallelBlock mean (hifi_gan.rs:114-117): stack(...).mean(0) == (a + b + c) * (1/3)
__global__ void k_mean3(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ y, size_t n) {
    const float third = (float)(1.0 / 3.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = ((a[i] + b[i]) + c[i]) * third;
}

// weight re-layout [A][B][K] -> [.. see callers ..]: dst[(i*K + k)*Cout + o] = src[src_index(o, i, k)]
__global__ void k_relayout_conv(const float* __restrict__ src, float* __restrict__ dst, int Cout, int CinG, int K, int transposed) {
    const size_t n = (size_t)Cout * CinG * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(e % Cout);
        const int k = (int)((e / Cout) % K);
        const int i = (int)(e / ((size_t)Cout * K));
        const size_t si = transposed ? ((size_t)i * Cout + o) * K + k   // ConvTranspose1d weight [Cin][Cout][K]
                                     : ((size_t)o * CinG + i) * K + k;  // Conv1d weight [Cout][Cin/g][K]
        dst[e] = src[si];
    }
}

// ConvTranspose1d weight [Cin][Cout][K], K = s * Kc -> polyphase causal-conv layout [Cin][Kc][Cout * s]:
//   dst[(i * Kc + kc) * (Cout * s) + o * s + ph] = src[(i * Cout + o) * K + ph + (Kc - 1 - kc) * s]
// (tap kc of the causal conv multiplies x[t + kc - (Kc - 1)], i.e. x[t - j] with j = Kc - 1 - kc)
__global__ void k_relayout_tconv(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int K, int s) {
    const int Kc = K / s;
    const size_t n = (size_t)Cout * Cin * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(e % ((size_t)Cout * s)), o = col / s, ph = col % s;
        const int kc = (int)((e / ((size_t)Cout * s)) % Kc);
        const int i = (int)(e / ((size_t)Cout * s * Kc));
        dst[e] = src[((size_t)i * Cout + o) * K + ph + (Kc - 1 - kc) * s];
    }
}

__global__ void k_synth_f32(float* __restrict__ dst, uint64_t key, size_t n, float mean, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = synth_elem(key, (uint64_t)i, mean, scale);
}

// ================================================================================================ launchers
#define FS_LAUNCH_CHECK() FS_HIP(hipGetLastError())

void codec_fsq_project(const uint32_t* codes, int B, int G, int T, const float* pw, const float* pb, int dg, float* z, hipStream_t st) {
    hipLaunchKernelGGL(k_fsq_project, dim3((T + 63) / 64, G, B), dim3(64), 0, st, codes, B, G, T, pw, pb, dg, z);
    FS_LAUNCH_CHECK();
}

static void conv1d_launch(const float* x, int B, int Cin, int T, const ConvW& w, int dil, bool pre_silu, int epi, const float* res,
                          const float* gamma, float* y, int ps, hipStream_t st, const float* ctx = nullptr) {
    const int K = w.k, Cout = w.cout;
    const int halo = (K - 1) * dil;
    FS_REQUIRE(!ctx || (B == 1 && halo <= CODEC_CTX_F32 && Cout < 16), "f32 streaming context: one item, halo <= 16, the VALU conv kernel");
    auto launch = [&](auto cpt, auto tpt, auto ich) {
        constexpr int CPT = decltype(cpt)::value, TPT = decltype(tpt)::value, ICH = decltype(ich)::value;
        constexpr int OT = 8 * CPT, TT = 32 * TPT;
        const size_t smem = sizeof(float) * ((size_t)ICH * (TT + halo) + (size_t)ICH * K * OT);
        FS_REQUIRE(smem <= 64 * 1024, "conv tile does not fit LDS");
        hipLaunchKernelGGL((k_conv1d<CPT, TPT, ICH>), dim3((T + TT - 1) / TT, (Cout + OT - 1) / OT, B), dim3(256), smem, st, x, Cin, T,
                           w.wt, w.b, Cout, K, dil, pre_silu ? 1 : 0, epi, res, gamma, y, ps, ctx);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>; using I16 = std::integral_constant<int, 16>;
    const bool mfma_ok = Cin >= 16 && Cout >= 16 && K <= 13 && halo <= 256;
    if (Cout == 1 && Cin == 16 && dil == 1 && ps == 1 && K - 1 <= CODEC_CTX_F32 && (epi == CODEC_EPI_NONE || epi == CODEC_EPI_TANH || epi == CODEC_EPI_GELU)) {
        hipLaunchKernelGGL((k_conv1d_one<16>), dim3((T + 511) / 512, 1, B), dim3(128), 0, st, x, T, w.wt, w.b, K, pre_silu ? 1 : 0, epi, y, ctx);
        FS_LAUNCH_CHECK();
        return;
    }
    if (w.wp && codec_conv1d_bf3_ok(Cin, Cout, K, dil)) {  // "bf16x3" precision mode: split operands on the bf16 matrix cores
        codec_conv1d_bf3(x, nullptr, B, Cin, T, w.wp, w.f16, w.b, Cout, K, dil, pre_silu, epi, res, gamma, y, nullptr, false, ps, st);
        return;
    }
    if (mfma_ok) {  // matrix cores: 64 (or, for the thin late stages, 32) channels x 128 or 256 samples per block
        constexpr int ICH = 16;
        // 64-channel blocks only where they still give every CU a block; else 32-channel blocks (twice as many).  Measured on the
        // 256-channel stage: 256 blocks of 64 channels 120 us vs 512 of 32 channels 126 us; 64 blocks 116 us vs 128 blocks 96 us
        const bool tall = Cout >= 64 && (long long)((T + 127) / 128) * ((Cout + 63) / 64) * B >= 256;
        const int OT = tall ? 64 : 32;
        // 256-sample blocks (twice the MFMA work per staged weight tile / barrier) where >= 512 blocks remain
        const bool wide = (long long)((T + 255) / 256) * ((Cout + OT - 1) / OT) * B >= 512 && 256 + halo <= 512;
        const int TT = wide ? 256 : 128;
        FS_REQUIRE(TT + halo <= (TT == 256 ? 512 : 256), "conv window does not fit the MFMA kernel's staging");
        const size_t smem = sizeof(float) * ((((size_t)ICH * (TT + halo) + 3) & ~(size_t)3) + (size_t)ICH * K * OT);
        FS_REQUIRE(smem <= 128 * 1024, "conv tile does not fit LDS");
        auto go = [&](auto kern) {
            if (smem > 64 * 1024) {  // above the default dynamic-LDS limit: raise it once per kernel (not inside a graph capture)
                static thread_local std::set<const void*> raised;
                if (raised.insert((const void*)kern).second)
                    FS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            }
            hipLaunchKernelGGL(kern, dim3((T + TT - 1) / TT, (Cout + OT - 1) / OT, B), dim3(256), smem, st, x, Cin, T, w.wt, w.b, Cout, K, dil,
                               pre_silu ? 1 : 0, epi, res, gamma, y, ps);
        };
        if (OT == 64 && TT == 256) go(k_conv1d_mfma<ICH, 64, 256>);
        else if (OT == 64) go(k_conv1d_mfma<ICH, 64, 128>);
        else if (TT == 256) go(k_conv1d_mfma<ICH, 32, 256>);
        else go(k_conv1d_mfma<ICH, 32, 128>);
    } else if (Cout >= 64) launch(I8(), I4(), I8());   // 64 ch x 128 t
    else if (Cout >= 32) launch(I4(), I4(), I8());     // 32 ch x 128 t
    else if (Cout >= 16) launch(I2(), I8(), I16());    // 16 ch x 256 t
    else launch(I1(), I8(), I16());                    // <= 8 ch x 256 t (conv_post: 1 channel)
    FS_LAUNCH_CHECK();
}

void codec_conv1d(const float* x, int B, int Cin, int T, const ConvW& w, int dil, bool pre_silu, int epi, const float* res,
                  const float* gamma, float* y, hipStream_t st, const float* ctx) {
    conv1d_launch(x, B, Cin, T, w, dil, pre_silu, epi, res, gamma, y, 1, st, ctx);
}

// Transposed conv (stride s, K = s * Kc taps, right trim K - s: utils/mod.rs:110-122) as ONE causal conv with Kc taps and
// Cout * s GEMM rows (row = channel * s + phase): y[o][t*s + ph] = b[o] + sum_i sum_j x[i][t - j] * W[i][o][ph + j*s].
// `w` must hold the polyphase re-layout made by codec_relayout_tconv ([Cin][Kc][Cout * s]) with w.cout = Cout, w.k = K.
void codec_tconv1d(const float* x, int B, int Cin, int Tin, const ConvW& w, int stride, bool pre_silu, float* y, hipStream_t st) {
    FS_REQUIRE(w.k % stride == 0, "transposed conv kernel size must be a multiple of its stride");
    ConvW p = w;
    p.cout = w.cout * stride;
    p.k = w.k / stride;
    conv1d_launch(x, B, Cin, Tin, p, 1, pre_silu, CODEC_EPI_NONE, nullptr, nullptr, y, stride, st);
}

void codec_conv1d_planes(const float* x, const uint16_t* xp, int B, int Cin, int T, const ConvW& w, int dil, bool pre_silu, int epi,
                         const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, hipStream_t st,
                         const uint16_t* ctx_in, uint16_t* ctx_out, const float* mean_a, const float* mean_b) {
    FS_REQUIRE(w.wp, "the plane data flow needs packed bf16x3 weights");
    codec_conv1d_bf3(x, xp, B, Cin, T, w.wp, w.f16, w.b, w.cout, w.k, dil, pre_silu, epi, res, gamma, y, yp, post_silu, 1, st, ctx_in, ctx_out, mean_a,
                     mean_b);
}

void codec_tconv1d_planes(const uint16_t* xp, int B, int Cin, int Tin, const ConvW& w, int stride, float* y, hipStream_t st) {
    FS_REQUIRE(w.wp && w.k % stride == 0, "the plane data flow needs packed bf16x3 weights of a polyphase transposed conv");
    codec_conv1d_bf3(nullptr, xp, B, Cin, Tin, w.wp, w.f16, w.b, w.cout * stride, w.k / stride, 1, false, CODEC_EPI_NONE, nullptr, nullptr, y, nullptr,
                     false, stride, st);
}

__global__ void k_save_tail_f32(const float* __restrict__ x, int T, float* __restrict__ ctx) {
    ctx[(size_t)blockIdx.x * CODEC_CTX_F32 + threadIdx.x] = x[(size_t)blockIdx.x * T + T - CODEC_CTX_F32 + threadIdx.x];
}
void codec_save_tail_f32(const float* x, int C, int T, float* ctx_out, hipStream_t st) {
    FS_REQUIRE(T >= CODEC_CTX_F32, "chunk shorter than the f32 streaming context");
    hipLaunchKernelGGL(k_save_tail_f32, dim3(C), dim3(CODEC_CTX_F32), 0, st, x, T, ctx_out);
    FS_LAUNCH_CHECK();
}

void codec_dwconv_ln(const float* x, int B, int C, int T, const float* dw, const float* db, const float* lnw, const float* lnb, float* y,
                     hipStream_t st, const float* ctx) {
    FS_REQUIRE(C <= 1024, "ConvNeXt width above 1024 channels");
    FS_REQUIRE(!ctx || B == 1, "f32 streaming context: one item");
    hipLaunchKernelGGL(k_dwconv_ln, dim3(T, B), dim3(256), 0, st, x, C, T, dw, db, lnw, lnb, y, ctx);
    FS_LAUNCH_CHECK();
}

void codec_mean3(const float* a, const float* b, const float* c, float* y, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_mean3, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, a, b, c, y, n);
    FS_LAUNCH_CHECK();
}

void codec_relayout(const float* src, float* dst, int Cout, int CinG, int K, bool transposed, hipStream_t st) {
    const size_t n = (size_t)Cout * CinG * K;
    hipLaunchKernelGGL(k_relayout_conv, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, src, dst, Cout, CinG, K,
                       transposed ? 1 : 0);
    FS_LAUNCH_CHECK();
}

void codec_relayout_tconv(const float* src, float* dst, int Cout, int Cin, int K, int stride, hipStream_t st) {
    FS_REQUIRE(K % stride == 0, "transposed conv kernel size must be a multiple of its stride");
    const size_t n = (size_t)Cout * Cin * K;
    hipLaunchKernelGGL(k_relayout_tconv, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, src, dst, Cout, Cin, K, stride);
    FS_LAUNCH_CHECK();
}

void codec_synth_fill(float* dst, uint64_t key, size_t n, float mean, float scale, hipStream_t st) {
    hipLaunchKernelGGL(k_synth_f32, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, dst, key, n, mean, scale);
    FS_LAUNCH_CHECK();
}

}  // namespace fs

// ================================================================================================ encoder side kernels
// (FireflyCodec::encode, firefly.rs:37-40: log-mel front-end -> ConvNeXt encoder -> downsample -> grouped FSQ)
namespace fs {

// ---- STFT magnitude (audio/spectrogram.rs:29-88 + stft.rs:52-90).  One block per frame: frame f = padded[f*hop, f*hop + N),
// padded = reflect pad that REPEATS the edge sample ((N - hop)/2 on both sides), tail beyond the padded signal zero-filled;
// periodic Hann window and a radix-2 FFT in f64 (the reference runs rustfft in f64), |.| rounded to f32, + 1e-6.
// Output channel-first: lin[k][f], k < N/2+1.
template <int N>
__global__ __launch_bounds__(256) void k_stft_mag(const float* __restrict__ pcm, int n, int hop, int n_frames, float* __restrict__ lin) {
    __shared__ double re[N], im[N];
    __shared__ double twr[N / 2], twi[N / 2];
    const int f = blockIdx.x, pad = (N - hop) / 2;
    const long long Lp = (long long)n + 2 * pad;
    for (int j = threadIdx.x; j < N / 2; j += 256) {
        const double a = -2.0 * 3.14159265358979323846 * (double)j / (double)N;
        twr[j] = cos(a); twi[j] = sin(a);
    }
    for (int j = threadIdx.x; j < N; j += 256) {
        const long long p = (long long)f * hop + j;
        double v = 0.0;
        if (p < Lp) {
            const long long q = p - pad;
            const long long src = q < 0 ? (-q - 1) : (q >= n ? (2LL * n - 1 - q) : q);
            v = (double)pcm[src];
        }
        const double w = 0.5 * (1.0 - cos((2.0 * 3.14159265358979323846 * (double)j) / (double)N));
        // bit-reversed store
        const int r = (int)(__brev((unsigned)j) >> (32 - __builtin_ctz(N)));
        re[r] = v * w; im[r] = 0.0;
    }
    __syncthreads();
    for (int len = 2; len <= N; len <<= 1) {
        const int half = len >> 1, step = N / len;
        for (int b = threadIdx.x; b < N / 2; b += 256) {
            const int k = b % half, i = (b / half) * len + k;
            const double wr = twr[k * step], wi = twi[k * step];
            const double xr = re[i + half] * wr - im[i + half] * wi, xi = re[i + half] * wi + im[i + half] * wr;
            re[i + half] = re[i] - xr; im[i + half] = im[i] - xi;
            re[i] += xr; im[i] += xi;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k <= N / 2; k += 256)
        lin[(size_t)k * n_frames + f] = (float)sqrt(re[k] * re[k] + im[k] * im[k]) + 1e-6f;
}

// mel[m][f] = log(clamp(sum_k lin[k][f] * fb[k][m], 1e-5, 100))  (spectrogram.rs:136-151); ascending-k f32 chain
__global__ void k_mel_log(const float* __restrict__ lin, const float* __restrict__ fb, int nf, int n_mels, int F, float* __restrict__ mel) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (f >= F) return;
    float acc = 0.f;
    for (int k = 0; k < nf; ++k) acc = __fadd_rn(acc, __fmul_rn(lin[(size_t)k * F + f], fb[(size_t)k * n_mels + m]));
    mel[(size_t)m * F + f] = logf(fminf(fmaxf(acc, 1e-5f), 100.0f));
}

// LayerNormChannelsFirst (convnext.rs:144-154), eps 1e-6: one thread per time step, two passes over the channels
__global__ void k_layernorm_cf(const float* __restrict__ x, int C, int T, const float* __restrict__ w, const float* __restrict__ b,
                               float* __restrict__ y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float mean = 0.f;
    for (int c = 0; c < C; ++c) mean += x[(size_t)c * T + t];
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) { const float d = x[(size_t)c * T + t] - mean; var += d * d; }
    var /= (float)C;
    const float sd = sqrtf(var + 1e-6f);
    for (int c = 0; c < C; ++c) y[(size_t)c * T + t] = (x[(size_t)c * T + t] - mean) / sd * w[c] + b[c];
}

// strided FishConvNet with k == stride (quantizer.downsample, quantizer.rs:44-57: left pad k - stride = 0) as a 1x1 conv over
// the space-to-depth view: y[i*s + k][t] = x[i][t*s + k]
__global__ void k_space_to_depth(const float* __restrict__ x, int C, int T, int s, int Tout, float* __restrict__ y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, ck = blockIdx.y;
    if (t >= Tout) return;
    y[(size_t)ck * Tout + t] = x[(size_t)(ck / s) * T + (size_t)t * s + ck % s];
}

// grouped residual FSQ with one quantizer per group (grouped_residual_fsq.rs:75-93,154-173, fsq.rs:68-118), levels (8,5,5,5):
// z = project_in(x_g); r = bound(z); code = round(bound(r)) / half_width; index = sum_k (code_k * hw_k + hw_k) * basis_k
__device__ __forceinline__ float dfsq_bound(float z, int lv) {
    const float half_l = ((float)lv - 1.0f) * 1.001f / 2.0f;
    const float offset = (lv % 2 == 0) ? 0.5f : 0.0f;
    const float q = offset / half_l;
    const float shift = logf((1.0f + q) / (1.0f - q)) * 0.5f;
    return tanhf(z + shift) * half_l - offset;
}
__global__ void k_fsq_encode(const float* __restrict__ z, int C, int T, int G, const float* __restrict__ pin_w /*[G][4][dg]*/,
                             const float* __restrict__ pin_b /*[G][4]*/, uint32_t* __restrict__ codes /*[G][T]*/) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (t >= T) return;
    const int dg = C / G;
    const int levels[4] = {8, 5, 5, 5}, basis[4] = {1, 8, 40, 200};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < dg; ++c) {
        const float v = z[(size_t)(g * dg + c) * T + t];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __fadd_rn(acc[k], __fmul_rn(v, pin_w[((size_t)g * 4 + k) * dg + c]));
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = dfsq_bound(acc[k] + pin_b[g * 4 + k], levels[k]);
        const float hw = (float)(levels[k] / 2);
        const float code = roundf(dfsq_bound(r, levels[k])) / hw;
        sum += (code * hw + hw) * (float)basis[k];
    }
    codes[(size_t)g * T + t] = (uint32_t)(long long)sum;
}

void codec_stft_mag(const float* pcm, int n, int n_fft, int hop, int n_frames, float* lin, hipStream_t st) {
    FS_REQUIRE(n_fft == 2048, "the STFT kernel is built for n_fft = 2048 (LogMelSpectrogramConfig::default)");
    hipLaunchKernelGGL((k_stft_mag<2048>), dim3(n_frames), dim3(256), 0, st, pcm, n, hop, n_frames, lin);
    FS_HIP(hipGetLastError());
}
void codec_mel_log(const float* lin, const float* fb, int nf, int n_mels, int F, float* mel, hipStream_t st) {
    hipLaunchKernelGGL(k_mel_log, dim3((F + 63) / 64, n_mels), dim3(64), 0, st, lin, fb, nf, n_mels, F, mel);
    FS_HIP(hipGetLastError());
}
void codec_layernorm_cf(const float* x, int C, int T, const float* w, const float* b, float* y, hipStream_t st) {
    hipLaunchKernelGGL(k_layernorm_cf, dim3((T + 63) / 64), dim3(64), 0, st, x, C, T, w, b, y);
    FS_HIP(hipGetLastError());
}
void codec_space_to_depth(const float* x, int C, int T, int s, float* y, hipStream_t st) {
    const int Tout = T / s;
    hipLaunchKernelGGL(k_space_to_depth, dim3((Tout + 63) / 64, C * s), dim3(64), 0, st, x, C, T, s, Tout, y);
    FS_HIP(hipGetLastError());
}
void codec_fsq_encode(const float* z, int C, int T, int G, const float* pin_w, const float* pin_b, uint32_t* codes, hipStream_t st) {
    hipLaunchKernelGGL(k_fsq_encode, dim3((T + 63) / 64, G), dim3(64), 0, st, z, C, T, G, pin_w, pin_b, codes);
    FS_HIP(hipGetLastError());
}

}  // namespace fs
