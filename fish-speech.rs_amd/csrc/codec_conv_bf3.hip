// Causal conv1d of the Firefly vocoder on the bf16 matrix cores with split operands ("bf16x3"), gfx950.
//
// Same contract as k_conv1d / k_conv1d_mfma (codec_kernels.hip; hifi_gan.rs:74-85,208-216, convnext.rs:110-122): stride 1, left zero
// pad (K-1)*dil, SiLU pre-activation, bias / GELU / gamma+residual / residual / tanh epilogues, polyphase transposed convs.
//
// Arithmetic.  The convolution is the GEMM Y[o][t] = sum_{(i,k)} W[o][i][k] * X[i][t + k*dil - halo].  Every f32 operand v is split
// into two bf16 parts v = hi + lo + e, hi = bf16_rne(v), lo = bf16_rne(v - hi), |e| <= 2^-17 |v|, and a product is evaluated as
// W_hi*X_hi + W_hi*X_lo + W_lo*X_hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation: three matrix instructions of 16 reduction
// items each instead of eight v_mfma_f32_32x32x2_f32 of two items (the f32 matrix rate is 1/16 of the bf16 rate).  The dropped
// W_lo*X_lo term and the two split residuals are each <= 2^-16 relative per product, i.e. ~1e-5: the PCM stays within the 1e-4 RMS
// acceptance bound of the f32 oracle (measured in tests/test_codec_gpu.py), but it is NOT the exact-f32 product chain of
// k_conv1d_mfma, which is kept as the codec's "f32" precision mode and for the encoder.
// The summation order of one output element is fixed -- 16-channel blocks ascending, taps ascending, (hi*hi, hi*lo, lo*hi) -- and
// does not depend on the tile shape, so a code prefix still decodes to the bit-identical PCM prefix whatever T is.
//
// Layout.  Weights are packed once at load time into MFMA A-operand order:
//   wp[ib][k][plane][o (Cout padded to 64)][8 bf16],  plane = part * 2 + half, part 0 = hi / 1 = lo,
//   element e of (ib, half) = input channel ib*16 + half*8 + e  (zero beyond Cin / Cout)
// so that the 32 lanes of a half-wave read 32 consecutive 16-byte slots (conflict-free ds_read_b128) and a block's weight tile
// for one (k, plane) is one contiguous run in global memory.  The x window is staged per 16-channel block as
//   xs[plane][window position][8 bf16]  (same planes)
// by the thread that owns the window position: 16 coalesced f32 loads (one per channel), SiLU, split, four 16-byte LDS stores.
// A block is 4 waves = OT output channels x TT samples; OT = 64: wave = 32 channels x TT/2 samples, OT = 32: 32 channels x TT/4.
// The global loads of stage n+1 are in flight (registers) while stage n runs on the matrix cores.  Layers whose whole weight
// set fits the stage (Cin <= 16 * NIBS: the thin 16/32-channel late stages) keep it resident in LDS and loop over time tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>

#include "codec_kernels.h"
#include "fs_common.h"

namespace fs {

// Range-checked twin (f16 precision mode only): this file is compiled a second time with -DFS_C3_CHECK into namespace
// fs::c3chk.  There every f32 -> f16 operand conversion (weights at pack time, activations in the producers' epilogues) counts operands that
// SATURATE (|x| > 65504) or are FLUSHED to zero (0 < |x| < 2^-24) in a device counter.  The unchecked build's launchers forward to the twin
// while codec_range_check(true) is set (fs_codec_set_range_check): the default kernels pay nothing for the diagnostic.
#ifdef FS_C3_CHECK
namespace c3chk {
#else
namespace c3chk {
size_t codec_pack_bf3_elems(int Cin, int K, int Cout, bool f16);
void codec_pack_bf3(const float* relaid, uint16_t* dst, int Cin, int K, int Cout, bool f16, hipStream_t st);
void codec_conv1d_bf3(const float* x, const uint16_t* xp, int B, int Cin, int T, const uint16_t* wp, bool f16, const float* bias, int Cout, int K,
                      int dil, bool pre_silu, int epi, const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, int ps,
                      hipStream_t st, const uint16_t* ctx_in, uint16_t* ctx_out, const float* mean_a, const float* mean_b);
void codec_respair_f16(const uint16_t* xp, int B, int C, int T, const uint16_t* w1p, const float* b1, const uint16_t* w2p, const float* b2, int K, int dil,
                       const float* res, float* y, uint16_t* yp, hipStream_t st, const uint16_t* mid_ctx_in, uint16_t* mid_ctx_out,
                       const uint16_t* ctx_in, uint16_t* ctx_out, const float* mean_a, const float* mean_b);
void codec_act_split(const float* x, int B, int C, int T, bool silu, uint16_t* planes, bool f16, hipStream_t st, const uint16_t* ctx_in, uint16_t* ctx_out);
void codec_mean3_planes(const float* a, const float* b, const float* c, int B, int C, int T, bool silu, uint16_t* planes, bool f16, hipStream_t st,
                        const uint16_t* ctx_in, uint16_t* ctx_out);
void codec_range_reset(hipStream_t st);
void codec_range_read(unsigned long long* out2, hipStream_t st);
}  // namespace c3chk
// thread_local: the twin is selected at LAUNCH time on the calling thread, so a concurrent decode of another codec handle on another thread
// keeps the plain kernels and cannot pollute this call's counts; two checked calls at once are serialised by the engine
// (codec_engine.hip range_guard_mutex: the counters are one pair per device)
static thread_local bool g_c3_checked = false;
void codec_range_check(bool on) { g_c3_checked = on; }
void codec_range_reset(hipStream_t st) { c3chk::codec_range_reset(st); }
void codec_range_read(unsigned long long* out2, hipStream_t st) { c3chk::codec_range_read(out2, st); }
#endif

#ifdef FS_C3_CHECK
__device__ unsigned long long g_c3_range[2];  // [0] operands beyond +-65504 (saturated), [1] non-zero operands below 2^-24 (flushed to zero)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float c3_silu(float x) { return x / (1.f + __expf(-x)); }
// plane outputs only (the value is split to 2 x bf16 right after): v_rcp_f32 (1 ulp) instead of the IEEE division sequence
__device__ __forceinline__ float c3_silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float c3_gelu(float x) { return 0.5f * x * (1.f + tanhf(0.7978845608028654f * x * (1.f + 0.044715f * x * x))); }
__device__ __forceinline__ uint32_t c3_bf16(float f) {  // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// round to nearest even, saturating.  |f| > 65504 does not occur with N(0, 1 / fan_in) synthetic convs; whether it does with a weight-normed
// checkpoint is what the range-checked twin (FS_C3_CHECK, see the top of the file) counts.
__device__ __forceinline__ uint32_t c3_f16(float f) {
#ifdef FS_C3_CHECK
    const float a = fabsf(f);
    if (a > 65504.f) atomicAdd(&g_c3_range[0], 1ull);
    else if (a != 0.f && a < 5.9604645e-8f) atomicAdd(&g_c3_range[1], 1ull);
#endif
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
}
// F16 = false: v = hi + lo in bf16 ("bf16x3"); F16 = true: one f16 value ("f16": `lo` is not used by any caller)
template <bool F16>
__device__ __forceinline__ void c3_split(float v, uint32_t& hi, uint32_t& lo) {
    if constexpr (F16) {
        hi = c3_f16(v);
        lo = 0;
    } else {
        hi = c3_bf16(v);
        lo = c3_bf16(v - __uint_as_float(hi << 16));
    }
}
// one 16-item reduction step of an accumulator tile: (hi*hi, hi*lo, lo*hi) in bf16, or the single f16 product
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 c3_mma(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x16 acc) {
    if constexpr (F16) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    }
}

// dst[((ib*K + k)*4 + p) * Cp + o][e] from the re-laid f32 weight src[(i*K + k)*Cout + o]
template <bool F16>
__global__ void k_pack_bf3(const float* __restrict__ src, uint16_t* __restrict__ dst, int Cin, int K, int Cout, int Cp) {
    constexpr int NPL = F16 ? 2 : 4;
    const int nib = (Cin + 15) / 16;
    const size_t n = (size_t)nib * K * NPL * Cp * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        size_t r = idx >> 3;
        const int o = (int)(r % Cp); r /= Cp;
        const int p = (int)(r % NPL); r /= NPL;
        const int k = (int)(r % K);
        const int ib = (int)(r / K);
        const int i = ib * 16 + (p & 1) * 8 + e;
        uint32_t hi = 0, lo = 0;
        if (i < Cin && o < Cout) c3_split<F16>(src[((size_t)i * K + k) * Cout + o], hi, lo);
        dst[idx] = (uint16_t)((p >> 1) ? lo : hi);
    }
}

// Activation planes: the split form of a (C, T) activation that the next convolution consumes,
//   planes[part (0 = hi, 1 = lo)][C/8][CODEC_PLANE_PAD + T][8 bf16]      (per batch item; C % 16 == 0)
// i.e. exactly the 16-byte MFMA B-operand slots of the x window, so staging is a straight copy (LDS-DMA: no registers, no VALU) or
// no staging at all (thin layers load their B operands straight from the planes).  The producing kernel writes them from its
// epilogue (SiLU of the consumer already applied): one activation is activated and split once, not once per (output-channel
// block, 16-channel stage) of the consumer.  Every (part, group) row starts with CODEC_PLANE_PAD zero slots -- the causal left
// padding -- written by the producer's first time tile, so a consumer never tests t >= 0; slots at t >= T may hold anything
// (causality: they only feed outputs at t >= T, which are not stored; the engine over-allocates the tail).
constexpr int PP = CODEC_PLANE_PAD;

#ifdef FS_C3_PROF  // tools/ubench_conv.hip only: cycles spent per phase of k_conv1d_bf3p, summed over the blocks' wave 0
__device__ unsigned long long g_c3prof[8];
#define C3_TICK(slot)                                                                 \
    do {                                                                              \
        const unsigned long long t_ = __builtin_readcyclecounter();                   \
        c3_acc[slot] += t_ - c3_t0;                                                   \
        c3_t0 = t_;                                                                   \
    } while (0)
#else
#define C3_TICK(slot) do {} while (0)
#endif

// Streaming (fs_codec_stream_*): the PP slots in front of a row are the causal LEFT CONTEXT of the tensor -- zeros for the first chunk of a
// stream (and for one-shot decoding), else the last PP slots of the same tensor in the previous chunk, kept in a per-tensor context
//   ctx[part][C/8][PP][8]
// `ci` = the context to copy in (null: zeros); producers also copy the last PP slots they write into `co` (null: not streaming).
// ParallelBlock mean folded into a residual conv (hifi_gan.rs:114-117): when `mean_a` is set, the residual epilogue's value c = res + conv
// (the third ResBlock's output) becomes ((mean_a + mean_b) + c) / 3 -- the same f32 operations in the same order as k_mean3_planes / k_mean3 --
// before it is stored / split, so the third block's f32 output and the mean kernel's three reads never touch memory.
struct PlaneCtx { const uint16_t* ci; uint16_t* co; const float* mean_a = nullptr; const float* mean_b = nullptr; };
template <bool F16>
__device__ __forceinline__ void c3_zero_pad(uint16_t* pb, int CG, int T, int g_first, int n_groups, int tid, int nthreads, const uint16_t* ci = nullptr) {
    constexpr int NP = F16 ? 1 : 2;
    const u32x4 z{0u, 0u, 0u, 0u};
    for (int e = tid; e < n_groups * NP * PP; e += nthreads) {
        const int slot = e % PP, gp = e / PP, g = g_first + gp / NP, part = gp % NP;
        if (g < CG)
            *reinterpret_cast<u32x4*>(pb + ((size_t)(part * CG + g) * (PP + T) + slot) * 8) =
                ci ? *reinterpret_cast<const u32x4*>(ci + ((size_t)(part * CG + g) * PP + slot) * 8) : z;
    }
}

// f32 (C, T) -> planes, optional SiLU.  One thread per (8-channel group, t).
template <bool F16>
__global__ void k_act_split(const float* __restrict__ x, int C, int T, int silu, uint16_t* __restrict__ planes, PlaneCtx pc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y, CG = C >> 3;
    const float* xb = x + (size_t)blockIdx.z * C * T;
    uint16_t* pb = planes + (size_t)blockIdx.z * (F16 ? 1 : 2) * CG * (PP + T) * 8;
    if (blockIdx.x == 0) c3_zero_pad<F16>(pb, CG, T, g, 1, threadIdx.x, blockDim.x, pc.ci);
    if (t >= T) return;
    u32x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = xb[(size_t)(g * 8 + 2 * e) * T + t], b = xb[(size_t)(g * 8 + 2 * e + 1) * T + t];
        if (silu) { a = c3_silu(a); b = c3_silu(b); }
        uint32_t ah, al, bh, bl;
        c3_split<F16>(a, ah, al); c3_split<F16>(b, bh, bl);
        vh[e] = ah | (bh << 16); vl[e] = al | (bl << 16);
    }
    *reinterpret_cast<u32x4*>(pb + ((size_t)g * (PP + T) + PP + t) * 8) = vh;
    if constexpr (!F16) *reinterpret_cast<u32x4*>(pb + ((size_t)(CG + g) * (PP + T) + PP + t) * 8) = vl;
    if (pc.co && t >= T - PP) {
        *reinterpret_cast<u32x4*>(pc.co + ((size_t)g * PP + (t - (T - PP))) * 8) = vh;
        if constexpr (!F16) *reinterpret_cast<u32x4*>(pc.co + ((size_t)(CG + g) * PP + (t - (T - PP))) * 8) = vl;
    }
}

// ParallelBlock mean (hifi_gan.rs:114-117) straight into planes: split(silu?(((a + b) + c) / 3))
template <bool F16>
__global__ void k_mean3_planes(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int C, int T, int silu,
                               uint16_t* __restrict__ planes, PlaneCtx pc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y, CG = C >> 3;
    const size_t boff = (size_t)blockIdx.z * C * T;
    uint16_t* pb = planes + (size_t)blockIdx.z * (F16 ? 1 : 2) * CG * (PP + T) * 8;
    if (blockIdx.x == 0) c3_zero_pad<F16>(pb, CG, T, g, 1, threadIdx.x, blockDim.x, pc.ci);
    if (t >= T) return;
    const float third = (float)(1.0 / 3.0);
    u32x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i0 = boff + (size_t)(g * 8 + 2 * e) * T + t, i1 = i0 + T;
        float u = ((a[i0] + b[i0]) + c[i0]) * third, v = ((a[i1] + b[i1]) + c[i1]) * third;
        if (silu) { u = c3_silu(u); v = c3_silu(v); }
        uint32_t uh, ul, wh, wl;
        c3_split<F16>(u, uh, ul); c3_split<F16>(v, wh, wl);
        vh[e] = uh | (wh << 16); vl[e] = ul | (wl << 16);
    }
    *reinterpret_cast<u32x4*>(pb + ((size_t)g * (PP + T) + PP + t) * 8) = vh;
    if constexpr (!F16) *reinterpret_cast<u32x4*>(pb + ((size_t)(CG + g) * (PP + T) + PP + t) * 8) = vl;
    if (pc.co && t >= T - PP) {
        *reinterpret_cast<u32x4*>(pc.co + ((size_t)g * PP + (t - (T - PP))) * 8) = vh;
        if constexpr (!F16) *reinterpret_cast<u32x4*>(pc.co + ((size_t)(CG + g) * PP + (t - (T - PP))) * 8) = vl;
    }
}

// Epilogue shared by the conv kernels.  D[row][col] of v_mfma_f32_32x32x*: register r of lane (h, c) holds row (r/4)*8 + h*4 + r%4,
// column c.  ob = first GEMM row of the wave's 32-row tile, tbase = first sample of its NT 32-sample tiles; y / res already carry
// the batch offset, ypb is the batch item's plane base (or null).
// Epilogue of one wave's NT 32x32 accumulator tiles, for ONE compile-time kind E (CODEC_EPI_*) and PS1 = "plain conv" (ps == 1).
// Branch-free up to the stores: the 16 bias values and the 16 * NT residual values of a lane are requested up front through clamped
// (always valid) addresses, so one memory round trip covers the tile -- with a validity branch around every output the loads were
// issued one at a time and the epilogue took half of a block's life; with a runtime kind switch every output carried the GELU /
// tanh / polyphase code (7 k instructions).
template <bool F16, int NT, int E, bool PS1>
__device__ __forceinline__ void c3_epilogue_k(const f32x16 (&acc)[NT], int ob, int tbase, int h, int c, int Cout, int T, int ps,
                                              const float* __restrict__ bias, const float* __restrict__ res, const float* __restrict__ gamma,
                                              float* __restrict__ y, uint16_t* __restrict__ ypb, int post_silu, uint16_t* __restrict__ yco = nullptr,
                                              const float* __restrict__ m0 = nullptr, const float* __restrict__ m1 = nullptr) {
    constexpr bool RES = E == CODEC_EPI_RES || E == CODEC_EPI_GAMMA_RES;
    const bool mean3 = E == CODEC_EPI_RES && m0 != nullptr;  // (uniform: a kernel argument)
    const float third = (float)(1.0 / 3.0);
    const int CGo = Cout >> 3;
    float bv[16], gv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = min(ob + (r >> 2) * 8 + h * 4 + (r & 3), Cout - 1);
        bv[r] = bias[PS1 ? o : o / ps];
        if (E == CODEC_EPI_GAMMA_RES) gv[r] = gamma[o];
    }
    // one 32-sample tile at a time (32-bit element offsets: every activation of a decode call is < 2^32 elements): the 16 residual values
    // of a tile are requested together, so a tile is one memory round trip, and only one tile's addresses / residuals are live
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = tbase + j * 32 + c, tc = min(t, T - 1);
        uint32_t oi[16];
        float rv[16], ma[16], mb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = min(ob + (r >> 2) * 8 + h * 4 + (r & 3), Cout - 1), oc = PS1 ? o : o / ps;
            oi[r] = PS1 ? (uint32_t)o * (uint32_t)T + (uint32_t)tc : ((uint32_t)oc * (uint32_t)T + (uint32_t)tc) * (uint32_t)ps + (uint32_t)(o % ps);  // (polyphase rows: see k_conv1d)
            if (RES) rv[r] = res[oi[r]];
        }
        if (E == CODEC_EPI_RES && mean3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ma[r] = m0[oi[r]]; mb[r] = m1[oi[r]]; }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float v4[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = q4 * 4 + rr, o = ob + q4 * 8 + h * 4 + rr;
                float v = acc[j][r] + bv[r];
                if (E == CODEC_EPI_GELU) v = c3_gelu(v);
                else if (E == CODEC_EPI_GAMMA_RES) v = rv[r] + gv[r] * v;
                else if (E == CODEC_EPI_RES) { v = rv[r] + v; if (mean3) v = ((ma[r] + mb[r]) + v) * third; }
                else if (E == CODEC_EPI_TANH) v = tanhf(v);
                if (y && o < Cout && t < T) y[oi[r]] = v;
                v4[rr] = v;
            }
            const int ob8 = ob + q4 * 8;  // first channel of this lane pair's 8-channel group
            if (ypb && ob8 < Cout && t < T) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)  // (the folded mean keeps k_mean3_planes' IEEE-division SiLU: bit-identical planes)
                    c3_split<F16>(post_silu ? (mean3 ? c3_silu(v4[rr]) : c3_silu_fast(v4[rr])) : v4[rr], hi[rr], lo[rr]);
                uint16_t* d = ypb + ((size_t)(ob8 >> 3) * (PP + T) + PP + t) * 8 + h * 4;
                *reinterpret_cast<uint2*>(d) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                if constexpr (!F16)
                    *reinterpret_cast<uint2*>(d + (size_t)CGo * (PP + T) * 8) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
                if (yco && t >= T - PP) {  // streaming: the last PP slots are the next chunk's left context
                    uint16_t* dc = yco + ((size_t)(ob8 >> 3) * PP + (t - (T - PP))) * 8 + h * 4;
                    *reinterpret_cast<uint2*>(dc) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                    if constexpr (!F16) *reinterpret_cast<uint2*>(dc + (size_t)CGo * PP * 8) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
                }
            }
        }
    }
}

// D[row][col] of v_mfma_f32_32x32x*: register r of lane (h, c) holds row (r/4)*8 + h*4 + r%4, column c.  ob = first GEMM row of the wave's
// 32-row tile, tbase = first sample of its NT 32-sample tiles; y / res already carry the batch offset, ypb is the batch item's plane base
// (or null).  EPI >= 0: compile-time kind (the plane kernels are instantiated per kind); EPI < 0: runtime `epi` (one specialised loop each).
template <bool F16, int NT, int EPI = -1, bool PS1 = false>
__device__ __forceinline__ void c3_epilogue(const f32x16 (&acc)[NT], int ob, int tbase, int h, int c, int Cout, int T, int ps,
                                            const float* __restrict__ bias, int epi, const float* __restrict__ res,
                                            const float* __restrict__ gamma, float* __restrict__ y, uint16_t* __restrict__ ypb,
                                            int post_silu, uint16_t* __restrict__ yco = nullptr, const float* __restrict__ m0 = nullptr,
                                            const float* __restrict__ m1 = nullptr) {
    if constexpr (EPI >= 0) {
        c3_epilogue_k<F16, NT, EPI, PS1>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco, m0, m1);
    } else {
        if (epi == CODEC_EPI_GELU) c3_epilogue_k<F16, NT, CODEC_EPI_GELU, false>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco);
        else if (epi == CODEC_EPI_GAMMA_RES) c3_epilogue_k<F16, NT, CODEC_EPI_GAMMA_RES, false>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco);
        else if (epi == CODEC_EPI_RES) c3_epilogue_k<F16, NT, CODEC_EPI_RES, false>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco, m0, m1);
        else if (epi == CODEC_EPI_TANH) c3_epilogue_k<F16, NT, CODEC_EPI_TANH, false>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco);
        else c3_epilogue_k<F16, NT, CODEC_EPI_NONE, false>(acc, ob, tbase, h, c, Cout, T, ps, bias, res, gamma, y, ypb, post_silu, yco);
    }
}

// ---- f32 input (the first convs of the decode path, whose producers are not plane writers): registers -> SiLU / split -> LDS.
// OT: output channels per block (64 | 32); TT: samples per block; NIBS: 16-channel blocks per stage; KMAX: largest tap count the
// register prefetch is sized for; NPX: window positions per thread (XS = TT + halo <= 256 * NPX).
// Outputs: f32 `y` (may be null when only planes are wanted) and / or planes `yp` (ps == 1 only) = split(post_silu ? silu(v) : v)
template <bool F16, int OT, int TT, int NIBS, int KMAX, int NPX>
__global__ __launch_bounds__(256) void k_conv1d_bf3(const float* __restrict__ x, int Cin, int T, const uint16_t* __restrict__ wp, int Cp,
                                                    const float* __restrict__ bias, int Cout, int K, int dil, int pre_silu, int epi,
                                                    const float* __restrict__ res, const float* __restrict__ gamma, float* __restrict__ y,
                                                    uint16_t* __restrict__ yp, int post_silu, int ps, int ntiles, PlaneCtx pc) {
    constexpr int WT_ = OT == 64 ? TT / 2 : TT / 4;  // samples per wave
    constexpr int NT = WT_ / 32;                     // 32-sample MFMA tiles per wave
    static_assert(NT >= 1 && (OT == 64 || OT == 32), "block shape");
    constexpr int NPL = F16 ? 2 : 4;                         // operand planes per 16-channel block: parts x channel halves
    constexpr int NWC = (NIBS * KMAX * NPL * OT + 255) / 256;  // 16-byte weight chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int halo = (K - 1) * dil, XS = TT + halo;
    const int nib = (Cin + 15) >> 4, nst = (nib + NIBS - 1) / NIBS;
    const bool resident = nst == 1;
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);      // [NIBS][NPL][XS]
    u32x4* ws = xs + NIBS * NPL * XS;                    // [NIBS][K][NPL][OT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int ob = OT == 64 ? (wave & 1) * 32 : 0, tb = OT == 64 ? (wave >> 1) * WT_ : wave * WT_;
    const int o0 = blockIdx.y * OT;
    const size_t boff_in = (size_t)blockIdx.z * Cin * T, boff_out = (size_t)blockIdx.z * Cout * T;
    const int wchunks = K * NPL * OT;  // per 16-channel block

    float xr[NPX][NIBS][16];
    u32x4 wr[NWC];
    auto load_x = [&](int st, int t0) {
#pragma unroll
        for (int b = 0; b < NIBS; ++b) {
            const int i0 = (st * NIBS + b) * 16;
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = tid + 256 * q, t = t0 + tl - halo;
                const bool tv = tl < XS && t >= 0 && t < T;
#pragma unroll
                for (int i = 0; i < 16; ++i) xr[q][b][i] = (tv && i0 + i < Cin) ? x[boff_in + (size_t)(i0 + i) * T + t] : 0.f;
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int b = 0; b < NIBS; ++b)
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = tid + 256 * q;
                if (tl < XS) {
                    uint32_t hi[16], lo[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) c3_split<F16>(pre_silu ? c3_silu(xr[q][b][i]) : xr[q][b][i], hi[i], lo[i]);
                    u32x4* d = xs + (b * NPL) * XS + tl;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        u32x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vh[e] = hi[hf * 8 + 2 * e] | (hi[hf * 8 + 2 * e + 1] << 16);
                            vl[e] = lo[hf * 8 + 2 * e] | (lo[hf * 8 + 2 * e + 1] << 16);
                        }
                        d[hf * XS] = vh;
                        if constexpr (!F16) d[(2 + hf) * XS] = vl;
                    }
                }
            }
    };
    // weight chunks of stage st: chunk e -> (block b, row kp = k*4 + plane, column o): contiguous OT-chunk runs in global memory
    auto load_w = [&](int st) {
#pragma unroll
        for (int j = 0; j < NWC; ++j) {
            const int e = j * 256 + tid, b = NIBS == 1 ? 0 : e / wchunks, r = e - b * wchunks, kp = r / OT, o = r % OT;
            const int ib = st * NIBS + b;
            if (e < NIBS * wchunks && ib < nib) wr[j] = *reinterpret_cast<const u32x4*>(wp + (((size_t)ib * K * NPL + kp) * Cp + o0 + o) * 8);
            else wr[j] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < NWC; ++j) {
            const int e = j * 256 + tid;
            if (e < NIBS * wchunks) ws[e] = wr[j];
        }
    };

    if (resident) {
        load_w(0);
        store_w();
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int t0 = tile * TT;
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        load_x(0, t0);
        if (!resident) load_w(0);
        for (int st = 0; st < nst; ++st) {
            __syncthreads();  // previous stage's (or tile's) LDS reads are done
            store_x();
            if (!resident) store_w();
            __syncthreads();
            if (st + 1 < nst) {
                load_x(st + 1, t0);
                load_w(st + 1);
            }
            const int nb = min(NIBS, nib - st * NIBS);
            for (int b = 0; b < nb; ++b) {
                const u32x4* wl = ws + (b * K * NPL + h) * OT + ob + c;
                const u32x4* xl = xs + (b * NPL + h) * XS + tb + c;
                for (int k = 0; k < K; ++k) {
                    const u32x4 ah = wl[k * NPL * OT], al = F16 ? ah : wl[(k * NPL + 2) * OT];
                    const u32x4* xk = xl + k * dil;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const u32x4 bh = xk[32 * j], bl = F16 ? bh : xk[2 * XS + 32 * j];
                        acc[j] = c3_mma<F16>(ah, al, bh, bl, acc[j]);
                    }
                }
            }
        }
        uint16_t* ypb = yp ? yp + (size_t)blockIdx.z * (F16 ? 1 : 2) * (Cout >> 3) * (PP + T) * 8 : nullptr;
        if (ypb && t0 == 0) c3_zero_pad<F16>(ypb, Cout >> 3, T, o0 >> 3, OT / 8, tid, 256, pc.ci);
        c3_epilogue<F16, NT>(acc, o0 + ob, t0 + tb, h, c, Cout, T, ps, bias, epi, res ? res + boff_out : nullptr, gamma, y ? y + boff_out : nullptr,
                        ypb, post_silu, ypb ? pc.co : nullptr);
    }
}

// ---- plane input, wide layers (Cin >= 64): one 16-channel block per stage, x window and weight tile staged by LDS-DMA
// (global_load_lds_dwordx4: 64 consecutive 16-byte slots per wave instruction, destination = wave-uniform base + lane * 16, source =
// uniform base + lane * 16: no per-lane address arithmetic, no staging registers).  Two or three blocks share a CU (LDS 42..65 KB
// each), so one block's DMA wait overlaps another's MFMA phase.  Requires Cin % 16 == 0.
template <bool F16, int OT, int TT, int KMAX, int EPI, bool PS1, int NIBS = 1>
__global__ __launch_bounds__(256, F16 ? 3 : 2) void k_conv1d_bf3p(const uint16_t* __restrict__ xp, int Cin, int T, const uint16_t* __restrict__ wp,
                                                        int Cp, const float* __restrict__ bias, int Cout, int K, int dil, int epi,
                                                        const float* __restrict__ res, const float* __restrict__ gamma,
                                                        float* __restrict__ y, uint16_t* __restrict__ yp, int post_silu, int ps, PlaneCtx pc) {
    constexpr int WT_ = OT == 64 ? TT / 2 : TT / 4, NT = WT_ / 32;
    static_assert(NT >= 1 && (OT == 64 || OT == 32), "block shape");
    constexpr int NPL = F16 ? 2 : 4, NPART = F16 ? 1 : 2;       // operand planes per 16-channel block (parts x channel halves)
    constexpr int NWP = (NIBS * KMAX * NPL * OT + 255) / 256;  // weight DMA pieces (256 chunks each) per stage, upper bound
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int halo = (K - 1) * dil, XSP = (TT + halo + 63) & ~63;
    const int nst = (Cin >> 4) / NIBS, CGi = Cin >> 3;  // stages of NIBS 16-channel blocks (NIBS > 1: pointwise convs, K = 1)
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);  // [NIBS][NPL][XSP]
    u32x4* ws = xs + NIBS * NPL * XSP;               // [NIBS][K][NPL][OT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int ob = OT == 64 ? (wave & 1) * 32 : 0, tb = OT == 64 ? (wave >> 1) * WT_ : wave * WT_;
    const int o0 = blockIdx.y * OT, t0 = blockIdx.x * TT;
    const size_t boff_out = (size_t)blockIdx.z * Cout * T;
    const size_t row = (size_t)(PP + T);  // slots per (part, group) row
    const u32x4* xpb = reinterpret_cast<const u32x4*>(xp) + (size_t)blockIdx.z * NPART * CGi * row + (PP + t0 - halo) + lane;
    // weight chunk (piece j, wave, lane) = chunk index e = j*256 + wave*64 + lane -> row kp = e / OT, column e % OT
    const int wl_off = OT == 64 ? lane : (lane >> 5) * Cp + (lane & 31);
    const u32x4* wpl = reinterpret_cast<const u32x4*>(wp) + o0 + wl_off;
    const int wtotal = NIBS * K * NPL * OT;  // chunks per stage: a multiple of 64, so a piece is whole per wave
    auto dma = [&](const u32x4* src, u32x4* dst_wave_base) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
    };
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#ifdef FS_C3_PROF
    unsigned long long c3_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long c3_t0 = __builtin_readcyclecounter();
#endif
    for (int st = 0; st < nst; ++st) {
        __syncthreads();  // previous stage's LDS reads are done
        C3_TICK(0);
        // x window: per 16-channel block 4 planes (hi / lo x channel half) of XSP slots
#pragma unroll
        for (int b = 0; b < NIBS; ++b)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const u32x4* src = xpb + ((size_t)((p >> 1) * CGi + 2 * (st * NIBS + b) + (p & 1))) * row;
                for (int q = wave * 64; q < XSP; q += 256) dma(src + q, xs + (b * NPL + p) * XSP + q);
            }
        // weight tiles [NIBS][K][4][OT] of these channel blocks: consecutive chunk rows of the packed tensor, Cp apart in global memory
        {
            const u32x4* src = wpl + (size_t)st * NIBS * K * NPL * Cp;
            constexpr int RPP = 256 / OT;  // chunk rows per piece
#pragma unroll
            for (int j = 0; j < NWP; ++j)
                if (j * 256 + wave * 64 < wtotal) dma(src + (size_t)(j * RPP + (OT == 64 ? wave : wave * 2)) * Cp, ws + j * 256 + wave * 64);
        }
        C3_TICK(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C3_TICK(2);
        __syncthreads();
        C3_TICK(3);
#pragma unroll
        for (int b = 0; b < NIBS; ++b) {
            const u32x4* wl = ws + (b * K * NPL + h) * OT + ob + c;
            const u32x4* xl = xs + (b * NPL + h) * XSP + tb + c;
            for (int k = 0; k < K; ++k) {
                const u32x4* xk = xl + k * dil;
                if constexpr (F16) {
                    const f16x8 a = __builtin_bit_cast(f16x8, wl[k * NPL * OT]);
                    f16x8 bv[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = __builtin_bit_cast(f16x8, xk[32 * j]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bv[j], acc[j], 0, 0, 0);
                } else {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wl[k * 4 * OT]), al = __builtin_bit_cast(bf16x8, wl[(k * 4 + 2) * OT]);
                    // (per accumulator the order stays hi*hi, hi*lo, lo*hi; the NT tiles are interleaved so that dependent MFMAs are NT apart)
                    bf16x8 bh[NT], bl[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) { bh[j] = __builtin_bit_cast(bf16x8, xk[32 * j]); bl[j] = __builtin_bit_cast(bf16x8, xk[2 * XSP + 32 * j]); }
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[j], 0, 0, 0);
                }
            }
        }
        C3_TICK(4);
    }
    uint16_t* ypb = yp ? yp + (size_t)blockIdx.z * NPART * (Cout >> 3) * row * 8 : nullptr;
    if (ypb && t0 == 0) c3_zero_pad<F16>(ypb, Cout >> 3, T, o0 >> 3, OT / 8, tid, 256, pc.ci);
    c3_epilogue<F16, NT, EPI, PS1>(acc, o0 + ob, t0 + tb, h, c, Cout, T, ps, bias, epi, res ? res + boff_out : nullptr, gamma,
                              y ? y + boff_out : nullptr, ypb, post_silu, ypb ? pc.co : nullptr, pc.mean_a ? pc.mean_a + boff_out : nullptr,
                              pc.mean_a ? pc.mean_b + boff_out : nullptr);
    C3_TICK(5);
#ifdef FS_C3_PROF
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&g_c3prof[i], c3_acc[i]);
#endif
}

// ---- plane input, thin layers (Cin = 16 * NIB <= 32: the 16 / 32-channel late stages, 40 % of the decode's samples x convs).
// These are bound by activation traffic, not by the matrix cores, so nothing is staged and nothing synchronises: the block's weight
// set (<= 45 KB) is loaded into LDS once, then every wave walks over its own 32-channel x (32 * NT)-sample tiles and loads its MFMA
// B operands -- 16-byte plane slots, 512 contiguous bytes per half-wave -- straight from global memory (the K taps of a tile re-read
// the same window from L1/L2).  No barrier after the weight load, so a CU keeps as many independent waves in flight as registers allow.
template <bool F16, int K, int NIB, int NT, int EPI, bool PS1>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void k_conv1d_bf3t(const uint16_t* __restrict__ xp, int T, const uint16_t* __restrict__ wp, int Cp,
                                                        const float* __restrict__ bias, int Cout, int dil, int epi,
                                                        const float* __restrict__ res, const float* __restrict__ gamma,
                                                        float* __restrict__ y, uint16_t* __restrict__ yp, int post_silu, int ps, int nwt, PlaneCtx pc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ws = reinterpret_cast<u32x4*>(smem_raw);  // [NIB][K][NPL][32]
    constexpr int NPL = F16 ? 2 : 4, NPART = F16 ? 1 : 2;
    constexpr int CGi = NIB * 2, WCH = NIB * K * NPL * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int o0 = blockIdx.y * 32, halo = (K - 1) * dil;
    const size_t row = (size_t)(PP + T);
    for (int e = tid; e < WCH; e += 256) {
        const int kp = e >> 5, o = e & 31;  // kp = (ib * K + k) * NPL + plane
        ws[e] = *reinterpret_cast<const u32x4*>(wp + ((size_t)kp * Cp + o0 + o) * 8);
    }
    __syncthreads();
    const u32x4* xpb = reinterpret_cast<const u32x4*>(xp) + (size_t)blockIdx.z * NPART * CGi * row + PP - halo + c;
    const size_t boff_out = (size_t)blockIdx.z * Cout * T;
    uint16_t* ypb = yp ? yp + (size_t)blockIdx.z * NPART * (Cout >> 3) * row * 8 : nullptr;
    if (ypb && blockIdx.x == 0) c3_zero_pad<F16>(ypb, Cout >> 3, T, o0 >> 3, 4, tid, 256, pc.ci);
    for (int wt = blockIdx.x * 4 + wave; wt < nwt; wt += gridDim.x * 4) {
        const int t0 = wt * (32 * NT);
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // flattened (channel block, tap) loop with the next step's B operands in flight while this step runs on the matrix cores
        const u32x4* xw = xpb + (size_t)h * row + t0;  // hi plane of this lane's channel half; lo plane = + CGi rows
        auto bsrc = [&](int i) { return xw + (size_t)(2 * (i / K)) * row + (i % K) * dil; };
        u32x4 bh[NT], bl[NT], nh[NT], nl[NT];
        {
            const u32x4* b0 = bsrc(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bh[j] = b0[32 * j]; bl[j] = F16 ? bh[j] : b0[(size_t)CGi * row + 32 * j]; }
        }
#pragma unroll 2
        for (int i = 0; i < NIB * K; ++i) {
            if (i + 1 < NIB * K) {
                const u32x4* b1 = bsrc(i + 1);
#pragma unroll
                for (int j = 0; j < NT; ++j) { nh[j] = b1[32 * j]; nl[j] = F16 ? nh[j] : b1[(size_t)CGi * row + 32 * j]; }
            }
            const u32x4* wl = ws + (i * NPL + h) * 32 + c;
            const u32x4 ah = wl[0], al = F16 ? ah : wl[64];
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = c3_mma<F16>(ah, al, bh[j], bl[j], acc[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bh[j] = nh[j]; bl[j] = nl[j]; }
        }
        int Tl = T;  // opaque per tile: keeps the epilogue's ~100 row addresses from being hoisted out of the tile loop (200+ VGPRs)
        asm volatile("" : "+s"(Tl));
        c3_epilogue<F16, NT, EPI, PS1>(acc, o0, t0, h, c, Cout, Tl, ps, bias, epi, res ? res + boff_out : nullptr, gamma, y ? y + boff_out : nullptr,
                                  ypb, post_silu, ypb ? pc.co : nullptr, pc.mean_a ? pc.mean_a + boff_out : nullptr,
                                  pc.mean_a ? pc.mean_b + boff_out : nullptr);
    }
}


// ---- one ResBlock pair of the thin late stages in ONE kernel (f16 mode): x' = x + conv2(silu(conv1(silu(x)))), both convs dilated by `dil`
// (hifi_gan.rs:74-85).  The intermediate never touches memory: a wave computes conv1 on its NTO output tiles PLUS the NH tiles to their left
// (the causal halo of conv2, recomputed by the left neighbour's wave as well), activates / rounds it exactly as the plane-writing epilogue would
// (bias, c3_silu_fast, f16), parks it in its own LDS scratch in B-operand order, and runs conv2 from there; the epilogue is the residual
// epilogue of the separate kernel (f32 residual stream in / out, next conv's planes, streaming context, folded ParallelBlock mean).  Same
// products in the same order as the two separate kernels: bit-identical output.  Per pair the planes of the intermediate (one write + one
// read of C x T x 2 bytes) and one launch disappear: 8 -> 6 plane-passes of HBM traffic on convs that are bandwidth-bound.
// Positions in front of the signal (t < 0) take the intermediate's streaming context (or zeros); the wave that owns t in [T - PP, T) saves it.
template <int K, int NIB, int NTO, int NH, int NW>
__global__ __launch_bounds__(NW * 64) void k_respair_f16t(const uint16_t* __restrict__ xp, int T, const uint16_t* __restrict__ w1p, const uint16_t* __restrict__ w2p,
                                                   int Cp, const float* __restrict__ b1, const float* __restrict__ b2, int dil, const float* __restrict__ res,
                                                   float* __restrict__ y, uint16_t* __restrict__ yp, int nwt, PlaneCtx pc, const uint16_t* __restrict__ mci,
                                                   uint16_t* __restrict__ mco) {
    constexpr int C = 16 * NIB, CG = 2 * NIB, NT1 = NTO + NH, HS = NT1 * 32, WCH = NIB * K * 2 * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ws1 = reinterpret_cast<u32x4*>(smem_raw);  // [NIB][K][2][32]
    u32x4* ws2 = ws1 + WCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4* hs = ws2 + WCH + (size_t)wave * CG * HS;   // this wave's intermediate: [CG][HS] slots of 8 channels
    const int h = lane >> 5, c = lane & 31;
    const int halo = (K - 1) * dil;
    const size_t row = (size_t)(PP + T);
    for (int e = tid; e < WCH; e += NW * 64) {
        const int kp = e >> 5, o = e & 31;
        ws1[e] = *reinterpret_cast<const u32x4*>(w1p + ((size_t)kp * Cp + o) * 8);
        ws2[e] = *reinterpret_cast<const u32x4*>(w2p + ((size_t)kp * Cp + o) * 8);
    }
    __syncthreads();
    const u32x4* xpb = reinterpret_cast<const u32x4*>(xp) + (size_t)blockIdx.z * CG * row;
    const size_t boff = (size_t)blockIdx.z * C * T;
    uint16_t* ypb = yp ? yp + (size_t)blockIdx.z * CG * row * 8 : nullptr;
    if (ypb && blockIdx.x == 0) c3_zero_pad<true>(ypb, CG, T, 0, CG, tid, NW * 64, pc.ci);
    float bv1[4 * CG > 16 ? 16 : 4 * CG];
#pragma unroll
    for (int r = 0; r < 4 * CG; ++r) bv1[r] = b1[(r >> 2) * 8 + h * 4 + (r & 3)];
    for (int wt = blockIdx.x * NW + wave; wt < nwt; wt += gridDim.x * NW) {
        const int t0 = wt * (32 * NTO), s0 = t0 - 32 * NH;
        // ---- conv1 on [s0, t0 + 32 NTO)
        {
            f32x16 acc[NT1];
#pragma unroll
            for (int j = 0; j < NT1; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // slot index of (tile j, step i) in a plane row, clamped at the row's first slot (only the first tiles of a signal reach in front of
            // it, and what they compute there is replaced by the context below)
            const int base = PP + s0 + c - halo;
            auto bsrc = [&](int i, int j) {
                const int idx = max(base + 32 * j + (i % K) * dil, 0);
                return xpb[(size_t)(2 * (i / K) + h) * row + idx];
            };
            u32x4 bh[NT1], nb[NT1];
#pragma unroll
            for (int j = 0; j < NT1; ++j) bh[j] = bsrc(0, j);
#pragma unroll 2
            for (int i = 0; i < NIB * K; ++i) {
                if (i + 1 < NIB * K) {
#pragma unroll
                    for (int j = 0; j < NT1; ++j) nb[j] = bsrc(i + 1, j);
                }
                const u32x4 a = ws1[(i * 2 + h) * 32 + c];
#pragma unroll
                for (int j = 0; j < NT1; ++j) acc[j] = c3_mma<true>(a, a, bh[j], bh[j], acc[j]);
#pragma unroll
                for (int j = 0; j < NT1; ++j) bh[j] = nb[j];
            }
            // ---- bias, SiLU, f16 -> this wave's scratch (B-operand slots), exactly what the plane-writing epilogue stores
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
                const int s = s0 + 32 * j + c;
#pragma unroll
                for (int q4 = 0; q4 < CG; ++q4) {
                    uint32_t hi[4], lo_;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) c3_split<true>(c3_silu_fast(acc[j][q4 * 4 + rr] + bv1[q4 * 4 + rr]), hi[rr], lo_);
                    uint2 v = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                    if (s < 0) {  // in front of the signal: the previous chunk's last PP slots, or zeros
                        v = make_uint2(0u, 0u);
                        if (mci && s >= -PP) v = *reinterpret_cast<const uint2*>(mci + ((size_t)q4 * PP + (PP + s)) * 8 + h * 4);
                    }
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(hs + (size_t)q4 * HS + 32 * j + c) + h * 4) = v;
                    if (mco && j >= NH && s >= T - PP && s < T)
                        *reinterpret_cast<uint2*>(mco + ((size_t)q4 * PP + (s - (T - PP))) * 8 + h * 4) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order; nothing to synchronise with)
        __builtin_amdgcn_wave_barrier();
        // ---- conv2 from the scratch
        f32x16 acc2[NTO];
#pragma unroll
        for (int j = 0; j < NTO; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
        const u32x4* hb = hs + (32 * NH - halo) + c;
#pragma unroll 2
        for (int i = 0; i < NIB * K; ++i) {
            const u32x4 a = ws2[(i * 2 + h) * 32 + c];
            const u32x4* hsrc = hb + (size_t)(2 * (i / K) + h) * HS + (i % K) * dil;
            u32x4 bb[NTO];
#pragma unroll
            for (int j = 0; j < NTO; ++j) bb[j] = hsrc[32 * j];
#pragma unroll
            for (int j = 0; j < NTO; ++j) acc2[j] = c3_mma<true>(a, a, bb[j], bb[j], acc2[j]);
        }
        __builtin_amdgcn_wave_barrier();  // the next tile's conv1 overwrites the scratch
        int Tl = T;
        asm volatile("" : "+s"(Tl));
        c3_epilogue<true, NTO, CODEC_EPI_RES, true>(acc2, 0, t0, h, c, C, Tl, 1, b2, CODEC_EPI_RES, res + boff, nullptr, y ? y + boff : nullptr, ypb, 1,
                                                  ypb ? pc.co : nullptr, pc.mean_a ? pc.mean_a + boff : nullptr, pc.mean_a ? pc.mean_b + boff : nullptr);
    }
}

}  // namespace

size_t codec_pack_bf3_elems(int Cin, int K, int Cout, bool f16) {
    const int Cp = (Cout + 63) / 64 * 64;
    return (size_t)((Cin + 15) / 16) * K * (f16 ? 2 : 4) * Cp * 8;
}

void codec_pack_bf3(const float* relaid, uint16_t* dst, int Cin, int K, int Cout, bool f16, hipStream_t st) {
#ifndef FS_C3_CHECK
    if (g_c3_checked) return c3chk::codec_pack_bf3(relaid, dst, Cin, K, Cout, f16, st);
#endif
    const int Cp = (Cout + 63) / 64 * 64;
    const size_t n = codec_pack_bf3_elems(Cin, K, Cout, f16);
    const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 4096));
    if (f16) hipLaunchKernelGGL(k_pack_bf3<true>, grid, dim3(256), 0, st, relaid, dst, Cin, K, Cout, Cp);
    else hipLaunchKernelGGL(k_pack_bf3<false>, grid, dim3(256), 0, st, relaid, dst, Cin, K, Cout, Cp);
    FS_HIP(hipGetLastError());
}

bool codec_conv1d_bf3_ok(int Cin, int Cout, int K, int dil) { return Cin >= 16 && Cout >= 16 && K <= 13 && (K - 1) * dil <= 256; }

// x (B, Cin, T) f32 or xp (activation planes) -> y (f32, may be null) and / or yp (planes, ps == 1); `Cout` = GEMM rows (channels * ps
// for a polyphase transposed conv)
template <bool F16>
static void conv1d_bf3_impl(const float* x, const uint16_t* xp, int B, int Cin, int T, const uint16_t* wp, const float* bias, int Cout, int K,
                            int dil, bool pre_silu, int epi, const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, int ps,
                            hipStream_t st, const uint16_t* ctx_in, uint16_t* ctx_out, const float* mean_a, const float* mean_b) {
    constexpr int NPL = F16 ? 2 : 4;
    const PlaneCtx pc{ctx_in, ctx_out, mean_a, mean_b};
    FS_REQUIRE((mean_a != nullptr) == (mean_b != nullptr) && (!mean_a || (xp && epi == CODEC_EPI_RES && ps == 1)),
               "the folded ParallelBlock mean needs both partners and a plane-input residual conv");
    FS_REQUIRE((!ctx_in && !ctx_out) || (yp && B == 1 && T >= PP), "streaming contexts need a plane output, one item and >= 64 samples per chunk");
    FS_REQUIRE(codec_conv1d_bf3_ok(Cin, Cout, K, dil), "conv shape outside the bf16x3 kernel's range");
    FS_REQUIRE((x != nullptr) != (xp != nullptr), "exactly one of the f32 input and the plane input");
    FS_REQUIRE(y || yp, "no output");
    FS_REQUIRE(!yp || (ps == 1 && Cout % 8 == 0), "plane output needs a plain conv with a multiple of 8 channels");
    const int Cp = (Cout + 63) / 64 * 64, halo = (K - 1) * dil, nib = (Cin + 15) / 16;
    auto raise_lds = [&](const void* kern, size_t smem) {
        FS_REQUIRE(smem <= 160 * 1024, "conv tile does not fit LDS");
        if (smem > 64 * 1024) {  // above the default dynamic-LDS limit: raise it once per kernel
            static thread_local std::set<const void*> raised;
            if (raised.insert(kern).second) FS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    };
    if (xp) {
        FS_REQUIRE(Cin % 16 == 0 && halo <= CODEC_PLANE_PAD, "plane input needs Cin % 16 == 0 and a halo within the plane padding");
        const bool resid = epi == CODEC_EPI_RES, ps1 = ps == 1;
        if (K == 1 && nib % 8 == 0) {
            // pointwise convs (ConvNeXt MLPs, the k = s = 2 upsampling convs in polyphase form): 8 channel blocks per stage
            FS_REQUIRE(epi == CODEC_EPI_NONE || ((epi == CODEC_EPI_GELU || epi == CODEC_EPI_GAMMA_RES) && ps1), "pointwise plane conv: epilogue kind");
            constexpr int OT = 32, TT = 128, NIBS = 8;
            const size_t smem = 16 * ((size_t)NIBS * NPL * TT + (size_t)NIBS * NPL * OT);
            auto go = [&](auto kern) {
                raise_lds((const void*)kern, smem);
                hipLaunchKernelGGL(kern, dim3((T + TT - 1) / TT, (Cout + OT - 1) / OT, B), dim3(256), smem, st, xp, Cin, T, wp, Cp, bias, Cout, K, dil,
                                   epi, res, gamma, y, yp, post_silu ? 1 : 0, ps, pc);
            };
            if (epi == CODEC_EPI_GELU) go(k_conv1d_bf3p<F16, OT, TT, 1, CODEC_EPI_GELU, true, NIBS>);
            else if (epi == CODEC_EPI_GAMMA_RES) go(k_conv1d_bf3p<F16, OT, TT, 1, CODEC_EPI_GAMMA_RES, true, NIBS>);
            else if (ps1) go(k_conv1d_bf3p<F16, OT, TT, 1, CODEC_EPI_NONE, true, NIBS>);
            else go(k_conv1d_bf3p<F16, OT, TT, 1, CODEC_EPI_NONE, false, NIBS>);
            FS_HIP(hipGetLastError());
            return;
        }
        FS_REQUIRE(epi == CODEC_EPI_NONE || (epi == CODEC_EPI_RES && ps == 1), "plane-input convs: plain or residual epilogue only");
        if (nib <= 2 && (K == 2 || K == 3 || K == 7 || K == 11)) {
            // thin late stages: B operands straight from the planes, weights resident in LDS, no barriers; one 32-sample tile per wave step
            // (3 blocks per CU; two tiles per step measured 7 % slower)
            constexpr int NT = 1;
            const int nwt = (T + 32 * NT - 1) / (32 * NT), ytiles = (Cout + 31) / 32;
            const int gx = std::max(1, std::min((nwt + 3) / 4, 768 / std::max(1, ytiles * B)));  // 3 blocks per CU, each walking over tiles
            const size_t smem = (size_t)nib * K * NPL * 32 * 16;
            auto go = [&](auto kern) {
                raise_lds((const void*)kern, smem);
                hipLaunchKernelGGL(kern, dim3(gx, ytiles, B), dim3(256), smem, st, xp, T, wp, Cp, bias, Cout, dil, epi, res, gamma, y, yp,
                                   post_silu ? 1 : 0, ps, nwt, pc);
            };
#define FS_THIN2(KK, NIB)                                                                        \
    do {                                                                                         \
        if (resid) go(k_conv1d_bf3t<F16, KK, NIB, NT, CODEC_EPI_RES, true>);                     \
        else if (ps1) go(k_conv1d_bf3t<F16, KK, NIB, NT, CODEC_EPI_NONE, true>);                 \
        else go(k_conv1d_bf3t<F16, KK, NIB, NT, CODEC_EPI_NONE, false>);                         \
    } while (0)
#define FS_THIN(KK)                            \
    do {                                       \
        if (nib == 1) FS_THIN2(KK, 1);         \
        else FS_THIN2(KK, 2);                  \
    } while (0)
            if (K == 2) FS_THIN(2);
            else if (K == 3) FS_THIN(3);
            else if (K == 7) FS_THIN(7);
            else FS_THIN(11);
#undef FS_THIN
#undef FS_THIN2
        } else {
            auto go = [&](auto kern, int OT, int TT, int NIBS = 1) {
                const int XSP = (TT + halo + 63) & ~63;
                const size_t smem = 16 * ((size_t)NIBS * NPL * XSP + (size_t)NIBS * K * NPL * OT);
                raise_lds((const void*)kern, smem);
                hipLaunchKernelGGL(kern, dim3((T + TT - 1) / TT, (Cout + OT - 1) / OT, B), dim3(256), smem, st, xp, Cin, T, wp, Cp, bias, Cout,
                                   K, dil, epi, res, gamma, y, yp, post_silu ? 1 : 0, ps, pc);
            };
            FS_REQUIRE(K <= 13, "tap count above the plane kernel's staging bound");
            // measured per tile shape (profiles/r02_vocoder_calls.txt): 32-channel blocks win everywhere (3 blocks per CU; 64-channel blocks
            // and double-buffered stages -- 1..2 blocks per CU -- were 5..30 % slower); 256-sample blocks where >= 1024 blocks remain.
            // Re-measured for the f16 mode in round 3 (half the LDS per block): 64-channel blocks +0..10 %, forced 128 / 256 samples +1 %
            static const char* cfg = getenv("FISHRT_BF3P_TT");  // tuning knob: force 128 / 256-sample blocks
            const long long b256 = (long long)((T + 255) / 256) * ((Cout + 31) / 32) * B;
            const int TT = cfg ? atoi(cfg) : (b256 >= 1024 ? 256 : 128);
#define FS_WIDE(TTv)                                                                             \
    do {                                                                                         \
        if (resid) go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_RES, true>, 32, TTv);            \
        else if (ps1) go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_NONE, true>, 32, TTv);        \
        else go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_NONE, false>, 32, TTv);                \
    } while (0)
            // channel blocks per stage (same summation order: blocks ascending, taps ascending): more matrix work between two barriers
            // f16 mode (half the LDS per channel block): 2 blocks per stage, still 3 thread blocks per CU -- 2.555 -> 2.470 ms per 256 frames,
            // bit-identical PCM; 4 blocks per stage (1..2 thread blocks per CU): 2.71 ms
            static const char* cfgn = getenv("FISHRT_BF3P_NIBS");
            // (four blocks per stage where the grid has at most two thread blocks per CU anyway -- the 256-channel stage at 256 frames: the LDS
            // they cost takes no occupancy away there)
            static const char* cfg4 = getenv("FISHRT_BF3P_NIBS4_BLOCKS");
            const long long blocks_tt = (long long)((T + TT - 1) / TT) * ((Cout + 31) / 32) * B;
            const int want = cfgn ? atoi(cfgn) : ((blocks_tt <= (cfg4 ? atoll(cfg4) : 512) && nib % 4 == 0) ? 4 : 2);
            const int nibs = (F16 && want > 0 && nib % want == 0) ? want : 1;
#define FS_WIDEN(TTv, NB)                                                                                \
    do {                                                                                                 \
        if (resid) go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_RES, true, NB>, 32, TTv, NB);            \
        else if (ps1) go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_NONE, true, NB>, 32, TTv, NB);        \
        else go(k_conv1d_bf3p<F16, 32, TTv, 13, CODEC_EPI_NONE, false, NB>, 32, TTv, NB);                \
    } while (0)
            if constexpr (F16) {
                if (nibs == 2) { if (TT == 256) FS_WIDEN(256, 2); else FS_WIDEN(128, 2); FS_HIP(hipGetLastError()); return; }
                if (nibs == 4) { if (TT == 256) FS_WIDEN(256, 4); else FS_WIDEN(128, 4); FS_HIP(hipGetLastError()); return; }
            }
#undef FS_WIDEN
            if (TT == 256) FS_WIDE(256);
            else FS_WIDE(128);
#undef FS_WIDE
        }
        FS_HIP(hipGetLastError());
        return;
    }
    auto go = [&](auto kern, int OT, int TT, int NIBS, bool loop_tiles) {
        const int XS = TT + halo;
        const size_t smem = 16 * ((size_t)NIBS * NPL * XS + (size_t)NIBS * K * NPL * OT);
        raise_lds((const void*)kern, smem);
        const int ntiles = (T + TT - 1) / TT, ytiles = (Cout + OT - 1) / OT;
        int gx = ntiles;
        if (loop_tiles) gx = std::max(1, std::min(ntiles, 1024 / std::max(1, ytiles * B)));  // weights stay in LDS over a loop of time tiles
        hipLaunchKernelGGL(kern, dim3(gx, ytiles, B), dim3(256), smem, st, x, Cin, T, wp, Cp, bias, Cout, K, dil, pre_silu ? 1 : 0, epi, res,
                           gamma, y, yp, post_silu ? 1 : 0, ps, ntiles, pc);
    };
    if (K == 1 && Cin >= 128) {
        // pointwise convs of the ConvNeXt blocks (frame-rate T, 512..2048 channels): 8 channel blocks per barrier pair
        go(k_conv1d_bf3<F16, 32, 128, 8, 1, 1>, 32, 128, 8, false);
    } else if (nib <= 2) {
        // thin layers: the whole weight set is resident, the block walks over time tiles
        if (nib == 1) go(k_conv1d_bf3<F16, 32, 256, 1, 13, 2>, 32, 256, 1, true);
        else go(k_conv1d_bf3<F16, 32, 128, 2, 13, 2>, 32, 128, 2, true);
    } else {
        const bool tall = Cout >= 64 && (long long)((T + 127) / 128) * ((Cout + 63) / 64) * B >= 256;
        const int OT = tall ? 64 : 32;
        const bool wide = (long long)((T + 255) / 256) * ((Cout + OT - 1) / OT) * B >= 512;
        if (OT == 64 && wide) go(k_conv1d_bf3<F16, 64, 256, 1, 13, 2>, 64, 256, 1, false);
        else if (OT == 64) go(k_conv1d_bf3<F16, 64, 128, 1, 13, 2>, 64, 128, 1, false);
        else if (wide) go(k_conv1d_bf3<F16, 32, 256, 1, 13, 2>, 32, 256, 1, false);
        else go(k_conv1d_bf3<F16, 32, 128, 1, 13, 2>, 32, 128, 1, false);
    }
    FS_HIP(hipGetLastError());
}

void codec_conv1d_bf3(const float* x, const uint16_t* xp, int B, int Cin, int T, const uint16_t* wp, bool f16, const float* bias, int Cout, int K,
                      int dil, bool pre_silu, int epi, const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, int ps,
                      hipStream_t st, const uint16_t* ctx_in, uint16_t* ctx_out, const float* mean_a, const float* mean_b) {
#ifndef FS_C3_CHECK
    if (g_c3_checked) return c3chk::codec_conv1d_bf3(x, xp, B, Cin, T, wp, f16, bias, Cout, K, dil, pre_silu, epi, res, gamma, y, yp, post_silu, ps, st, ctx_in, ctx_out, mean_a, mean_b);
#endif
    if (f16) conv1d_bf3_impl<true>(x, xp, B, Cin, T, wp, bias, Cout, K, dil, pre_silu, epi, res, gamma, y, yp, post_silu, ps, st, ctx_in, ctx_out, mean_a, mean_b);
    else conv1d_bf3_impl<false>(x, xp, B, Cin, T, wp, bias, Cout, K, dil, pre_silu, epi, res, gamma, y, yp, post_silu, ps, st, ctx_in, ctx_out, mean_a, mean_b);
}

bool codec_respair_ok(int C, int K, int dil, bool f16) {
    return f16 && (C == 16 || C == 32) && (K == 3 || K == 7 || K == 11) && (K - 1) * dil <= CODEC_PLANE_PAD && !getenv("FISHRT_VOC_NO_PAIR_FUSION");
}

void codec_respair_f16(const uint16_t* xp, int B, int C, int T, const uint16_t* w1p, const float* b1, const uint16_t* w2p, const float* b2, int K, int dil,
                       const float* res, float* y, uint16_t* yp, hipStream_t st, const uint16_t* mid_ctx_in, uint16_t* mid_ctx_out, const uint16_t* ctx_in,
                       uint16_t* ctx_out, const float* mean_a, const float* mean_b) {
#ifndef FS_C3_CHECK
    if (g_c3_checked) return c3chk::codec_respair_f16(xp, B, C, T, w1p, b1, w2p, b2, K, dil, res, y, yp, st, mid_ctx_in, mid_ctx_out, ctx_in, ctx_out, mean_a, mean_b);
#endif
    FS_REQUIRE(codec_respair_ok(C, K, dil, true), "ResBlock pair outside the fused kernel's range");
    FS_REQUIRE(res && (y || yp), "the fused ResBlock pair needs the residual input and an output");
    FS_REQUIRE((mean_a != nullptr) == (mean_b != nullptr), "the folded ParallelBlock mean needs both partners");
    FS_REQUIRE((!mid_ctx_in && !mid_ctx_out && !ctx_in && !ctx_out) || (B == 1 && T >= PP), "streaming contexts need one item and >= 64 samples per chunk");
    const PlaneCtx pc{ctx_in, ctx_out, mean_a, mean_b};
    const int Cp = 64, halo = (K - 1) * dil, nib = C / 16;
    constexpr int NTO = 4;
    const int nwt = (T + 32 * NTO - 1) / (32 * NTO);
    auto go = [&](auto kern, int NW, int NH) {
        const size_t smem = 16 * ((size_t)2 * nib * K * 2 * 32 + (size_t)NW * 2 * nib * (NTO + NH) * 32);
        FS_REQUIRE(smem <= 160 * 1024, "fused ResBlock pair does not fit LDS");
        if (smem > 64 * 1024) {
            static thread_local std::set<const void*> raised;
            if (raised.insert((const void*)kern).second) FS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        const int per_cu = std::max(1, (int)((150 * 1024) / smem));
        const int gx = std::max(1, std::min((nwt + NW - 1) / NW, 256 * per_cu / std::max(1, B)));
        hipLaunchKernelGGL(kern, dim3(gx, 1, B), dim3(NW * 64), smem, st, xp, T, w1p, w2p, Cp, b1, b2, dil, res, y, yp, nwt, pc, mid_ctx_in, mid_ctx_out);
    };
#define FS_PAIR(KK, NIBv, NWv)                                                    \
    do {                                                                          \
        if (halo <= 32) go(k_respair_f16t<KK, NIBv, NTO, 1, NWv>, NWv, 1);        \
        else go(k_respair_f16t<KK, NIBv, NTO, 2, NWv>, NWv, 2);                   \
    } while (0)
    // waves per block (the two weight sets are shared by a block's waves; every wave owns 2 nib x 6 KB of scratch): tuning knobs
    static const int nw1 = getenv("FISHRT_PAIR_NW1") ? atoi(getenv("FISHRT_PAIR_NW1")) : 4, nw2 = getenv("FISHRT_PAIR_NW2") ? atoi(getenv("FISHRT_PAIR_NW2")) : 8;
#define FS_PAIRK(NIBv, NWv) do { if (K == 3) FS_PAIR(3, NIBv, NWv); else if (K == 7) FS_PAIR(7, NIBv, NWv); else FS_PAIR(11, NIBv, NWv); } while (0)
    if (nib == 1) { if (nw1 == 8) FS_PAIRK(1, 8); else FS_PAIRK(1, 4); }
    else { if (nw2 == 4) FS_PAIRK(2, 4); else FS_PAIRK(2, 8); }
#undef FS_PAIRK
#undef FS_PAIR
    FS_HIP(hipGetLastError());
}

void codec_act_split(const float* x, int B, int C, int T, bool silu, uint16_t* planes, bool f16, hipStream_t st, const uint16_t* ctx_in,
                     uint16_t* ctx_out) {
#ifndef FS_C3_CHECK
    if (g_c3_checked) return c3chk::codec_act_split(x, B, C, T, silu, planes, f16, st, ctx_in, ctx_out);
#endif
    FS_REQUIRE(C % 8 == 0, "activation planes need a multiple of 8 channels");
    FS_REQUIRE((!ctx_in && !ctx_out) || (B == 1 && T >= PP), "streaming contexts need one item and >= 64 samples per chunk");
    const dim3 grid((T + 255) / 256, C / 8, B);
    const PlaneCtx pc{ctx_in, ctx_out};
    if (f16) hipLaunchKernelGGL(k_act_split<true>, grid, dim3(256), 0, st, x, C, T, silu ? 1 : 0, planes, pc);
    else hipLaunchKernelGGL(k_act_split<false>, grid, dim3(256), 0, st, x, C, T, silu ? 1 : 0, planes, pc);
    FS_HIP(hipGetLastError());
}

void codec_mean3_planes(const float* a, const float* b, const float* c, int B, int C, int T, bool silu, uint16_t* planes, bool f16,
                        hipStream_t st, const uint16_t* ctx_in, uint16_t* ctx_out) {
#ifndef FS_C3_CHECK
    if (g_c3_checked) return c3chk::codec_mean3_planes(a, b, c, B, C, T, silu, planes, f16, st, ctx_in, ctx_out);
#endif
    FS_REQUIRE(C % 8 == 0, "activation planes need a multiple of 8 channels");
    FS_REQUIRE((!ctx_in && !ctx_out) || (B == 1 && T >= PP), "streaming contexts need one item and >= 64 samples per chunk");
    const dim3 grid((T + 255) / 256, C / 8, B);
    const PlaneCtx pc{ctx_in, ctx_out};
    if (f16) hipLaunchKernelGGL(k_mean3_planes<true>, grid, dim3(256), 0, st, a, b, c, C, T, silu ? 1 : 0, planes, pc);
    else hipLaunchKernelGGL(k_mean3_planes<false>, grid, dim3(256), 0, st, a, b, c, C, T, silu ? 1 : 0, planes, pc);
    FS_HIP(hipGetLastError());
}

#ifdef FS_C3_CHECK
void codec_range_reset(hipStream_t st) {
    void* p = nullptr;
    FS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_c3_range)));
    FS_HIP(hipMemsetAsync(p, 0, 16, st));
}
void codec_range_read(unsigned long long* out2, hipStream_t st) {
    void* p = nullptr;
    FS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_c3_range)));
    FS_HIP(hipMemcpyAsync(out2, p, 16, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
}
}  // namespace c3chk
#endif

}  // namespace fs
