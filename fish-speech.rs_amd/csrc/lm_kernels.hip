// Dual-AR transformer kernels for gfx950 (MI355X): the batch-1 token path (GEMV kernels, flash-decoding attention, on-device
// samplers) first, then the MFMA row path (skinny GEMMs over activation rows, prefill flash attention) used by the prefill
// and by the static-batch generator.
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
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

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

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <>
struct WTr<fp8_t> {
    static constexpr int EPL = 16;  // OCP e4m3fn weights: 16 per 16-byte lane load
    using vec = u32x4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
            f[4 * i] = lo.x; f[4 * i + 1] = lo.y; f[4 * i + 2] = hi.x; f[4 * i + 3] = hi.y;
        }
    }
    __device__ static __forceinline__ float to_f32(fp8_t h) { return e4m3_to_f32(h.v); }
    __device__ static __forceinline__ fp8_t from_f32(float f) { return fp8_t{f32_to_e4m3(f)}; }
};
// per-row dequantisation scale (fp8 only; other weight types carry none)
template <typename WT>
__device__ __forceinline__ float row_scale(const float* __restrict__ ws, int row) {
    if constexpr (std::is_same<WT, fp8_t>::value) return ws[row];
    else return 1.0f;
}

template <typename V>
__device__ __forceinline__ V ld_stream(const V* p) {  // streamed-once weights: non-temporal (guide: nt-weights)
    return __builtin_nontemporal_load(p);
}

// ---- cross-lane reductions: DPP inside a 16-lane row (no LDS crossbar), v_readlane across the four rows.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each 16
__device__ __forceinline__ float readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// sum over the whole wave; the result is wave-uniform (scalar registers)
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_MIRROR>(v);
    return (readlane(v, 15) + readlane(v, 31)) + (readlane(v, 47) + readlane(v, 63));
}
// sum / max over aligned groups of N consecutive lanes (N in {1,2,4,8,16}); every lane of the group gets the result
template <int N>
__device__ __forceinline__ float group_sum(float v) {
    if (N >= 2) v += dpp_mov<DPP_XOR1>(v);
    if (N >= 4) v += dpp_mov<DPP_XOR2>(v);
    if (N >= 8) v += dpp_mov<DPP_HALF_MIRROR>(v);
    if (N >= 16) v += dpp_mov<DPP_MIRROR>(v);
    return v;
}

// Everything above this point is ISSUED before anything below it: keeps the machine scheduler from sinking the weight-stream
// loads under the wait for the small L2-resident vectors (it otherwise serialises x-load -> RMSNorm -> weight request, which
// costs a full L2 round trip + the norm per kernel node before the first HBM byte is even asked for).
#define FS_ISSUE_FENCE() __builtin_amdgcn_sched_barrier(0)

// A wave's view of a length-K vector / weight row: chunk c covers elements [c*64*EPL, (c+1)*64*EPL), lane l owns
// EPL consecutive elements starting at c*64*EPL + l*EPL.
template <typename WT, int K, bool NT = true>
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
            if (base < K) wv[c] = NT ? ld_stream(reinterpret_cast<const vec*>(row + base)) : *reinterpret_cast<const vec*>(row + base);
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
    // in-register RMSNorm of the wave's x slice: x / sqrt(mean(x^2) + eps) * w   (candle_nn::RmsNorm); `nr` = the
    // lane's slice of the norm weight, loaded by the caller together with x so that no load sits behind the reduction
    __device__ static __forceinline__ void rmsnorm(float (&xr)[NX], const float (&nr)[NX], float eps) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NX; ++i) ss = fmaf(xr[i], xr[i], ss);
        const float d = sqrtf(wave_sum(ss) / (float)K + eps);
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = (xr[i] / d) * nr[i];
    }
};

template <typename WT>
__device__ __forceinline__ WT* kv_addr(void* pool, const int* __restrict__ page_table, int t, int g, int Hk, int Dh) {
    const int page = page_table[t / KV_PAGE];
    return reinterpret_cast<WT*>(pool) + ((size_t)(page * Hk + g) * KV_PAGE + (t % KV_PAGE)) * Dh;
}

// ------------------------------------------------------------------------------------------------ qkv + rope + kv append
// One wave per ROW of Wqkv (measured: a 2 MB GEMV node costs 2.4 us at one row per wave, 3.1 us at two -- the wave count, not
// the byte count, sets the latency of these small nodes); the two rows of an interleaved-RoPE pair (2p, 2p+1) sit in adjacent
// waves of one block and meet through 8 bytes of LDS.  Position, page and cos/sin are requested while the weights are in
// flight, so the epilogue has no dependent load left.
template <typename WT, int K, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_qkv(const float* __restrict__ x, const float* __restrict__ norm_w, float eps,
                                                    const WT* __restrict__ W, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, const SeqState* __restrict__ state,
                                                    int pos_static, int rope_static, float* __restrict__ q_out, KVView kv,
                                                    int H, int Hk, int Dh, const float* __restrict__ wscale) {
    using R = Row<WT, K, NT>;
    using KT = KVT<WT>;
    static_assert(WAVES % 2 == 0, "row pairs live in one block");
    __shared__ float dots[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_rows = (H + 2 * Hk) * Dh;
    const int row = min(blockIdx.x * WAVES + wave, n_rows - 1);  // clamped (the grid covers whole pairs; n_rows is even)
    // small L2-resident vectors FIRST (vmcnt retires in order: a load issued after the weight stream would wait for it),
    // then the weight stream, then the position-dependent scalar chain (pos -> page / cos / sin, only needed by the
    // epilogue); the RMSNorm math and that chain overlap the weights' HBM flight
    float xr[R::NX], nr[R::NX];
    R::load_x(x, lane, xr);
    R::load_x(norm_w, lane, nr);
    const float sc = row_scale<WT>(wscale, row);  // fp8: requested in front of the weight stream (an L2 round trip at the tail otherwise)
    typename R::vec wv[R::NCH];
    R::load_w(W + (size_t)row * K, lane, wv);
    FS_ISSUE_FENCE();
    const int r0 = row & ~1, qdim = H * Dh, kdim = Hk * Dh, half = Dh / 2;
    const int pos = state ? state->pos : pos_static;
    const int rpos = state ? pos + state->rope_off : rope_static;
    float c = 1.f, s = 0.f;
    KT* dst = nullptr;
    if (r0 < qdim + kdim) {
        const int j = (r0 % Dh) / 2;
        c = cos_t[(size_t)rpos * half + j];
        s = sin_t[(size_t)rpos * half + j];
    }
    if (r0 >= qdim) {
        const int rk = (r0 - qdim) % kdim, g = rk / Dh, dd = rk % Dh;
        dst = kv_addr<KT>(r0 < qdim + kdim ? kv.k : kv.v, kv.page_table, pos, g, Hk, Dh) + dd;
    }
    R::rmsnorm(xr, nr, eps);
    const float d = wave_sum(R::dot(wv, xr)) * sc;
    if (lane == 0) dots[wave] = d;
    __syncthreads();
    if (lane != 0 || (wave & 1) || blockIdx.x * WAVES + wave >= n_rows) return;
    const float a = dots[wave], b = dots[wave + 1];
    if (r0 < qdim + kdim) {  // rope_i on the pair (2j, 2j+1) of its head (dual_ar.rs:246-247)
        const float o0 = a * c - b * s, o1 = a * s + b * c;
        if (r0 < qdim) { q_out[r0] = o0; q_out[r0 + 1] = o1; }
        else { dst[0] = WTr<KT>::from_f32(o0); dst[1] = WTr<KT>::from_f32(o1); }
    } else {
        dst[0] = WTr<KT>::from_f32(a); dst[1] = WTr<KT>::from_f32(b);
    }
}

// ------------------------------------------------------------------------------------------------ decode attention
// Flash-decoding over the paged cache.  grid = Hk * n_chunks_max blocks of 4 waves; block (g, c) owns the n_rep query
// heads of kv head g and the ATTN_CHUNK tokens [c*CH, (c+1)*CH); blocks past the current length exit at once, so the
// captured graph serves every sequence length.  Each wave stages its TW tokens of K and V (one coalesced 16-B/lane
// load per KiB, all in flight together) into its private LDS tile, then lane GROUPS of LPT lanes (LPT*16 B = one head
// row) each own one (query head, token subset): scores by DPP group-sums, a two-pass softmax in registers (no rescale
// chain), P.V accumulated on the group's own dims -- no cross-group reduction of the output.  The block merges its
// waves in LDS and writes one un-normalised partial {o[Dh], m, l} per (head, chunk); k_wo combines the chunks.
// chunk = NW waves x TW tokens (16 waves x 8 tokens was measured too: 4.49 vs 4.36 us, the wider block merge eats the shorter loop)
template <typename WT, int DH> struct AttnGeom { static constexpr int TW = 16, NW = 8; };                          // bf16: 8 waves x 16 tokens = 128-token chunks
template <int DH> struct AttnGeom<float, DH> { static constexpr int TW = 16, NW = 4; };                           // f32 :  64-token chunks

// position of activation row m: pos_step 1 = consecutive tokens of one sequence (prefill), 0 = every row at state->pos (lock-step static
// batch), -1 = row m is its own sequence with its own state (session slots, fs_lm_session_*)
__device__ __forceinline__ int row_pos(const SeqState* __restrict__ state, int m, int pos_step) {
    return pos_step < 0 ? state[m].pos : state->pos + m * pos_step;
}

template <typename WT, int DH, int NREP>
__global__ __launch_bounds__((AttnGeom<WT, DH>::NW * 64)) void k_attn_decode(const float* __restrict__ q_all, KVView kv,
                                                     const SeqState* __restrict__ state, float* __restrict__ part_all,
                                                     int Hk, int n_chunks_max, int nc_launch, int pos_step, int pt_stride, int hsplit) {
    // hsplit > 1: the query heads of a kv group are spread over hsplit blocks of NREP heads each (the score / P.V work of a wave is
    // VALU-bound: 16 tokens x 8 heads ~ 900 instructions; the K/V tiles are then read hsplit times, from L2)
    const int GH = NREP * hsplit;
    // blockIdx.y = activation row m (0 for the batch-1 decode step): prefill -> token pos + m of one sequence (pos_step 1,
    // pt_stride 0); batched decode -> sequence m at pos (pos_step 0, pt_stride = page-table stride)
    kv.page_table += (size_t)blockIdx.y * pt_stride;
    const float* q = q_all + (size_t)blockIdx.y * Hk * GH * DH;
    float* part = part_all + (size_t)blockIdx.y * Hk * GH * n_chunks_max * (DH + 2);
    constexpr int EPL = WTr<WT>::EPL;
    constexpr int LPT = DH / EPL;            // lanes per head row
    constexpr int G = 64 / LPT;              // lane groups per wave
    constexpr int NRP = NREP < G ? NREP : G; // heads served per pass
    constexpr int NTS = G / NRP;             // token subsets per head
    constexpr int NHP = NREP / NRP;          // head passes
    constexpr int TW = AttnGeom<WT, DH>::TW;     // tokens per wave
    constexpr int NW = AttnGeom<WT, DH>::NW;     // waves per block
    constexpr int CH = NW * TW;              // tokens per block
    constexpr int TPG = TW / NTS;            // tokens per group
    constexpr int NLD = TW * LPT / 64;       // 16-B loads per lane per tile
    static_assert(TW % NTS == 0 && NLD >= 1, "attention geometry");
    using vec = typename WTr<WT>::vec;
    // block id = head part * (Hk * nc_launch) + (kv head * nc_launch + chunk): with Hk * nc_launch a multiple of 8 the blocks that share
    // a K/V tile get ids congruent mod 8, i.e. the same XCD and L2 (workgroups go round-robin over the 8 XCDs)
    const int tile = blockIdx.x % (Hk * nc_launch), hb = (int)(blockIdx.x / (Hk * nc_launch)) * NREP;
    const int g = tile / nc_launch, c = tile % nc_launch;  // nc_launch <= n_chunks_max chunks are launched
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) WT sk[NW][TW * DH];
    __shared__ __attribute__((aligned(16))) WT sv[NW][TW * DH];
    __shared__ float sp[NW][NREP][NTS][DH + 2];
    const int t_base = c * CH + wave * TW;
    // stage K/V tiles: lane l of load i covers token (i*64 + l) / LPT, 16-B slice l % LPT (1 KiB contiguous per load)
    // all TW tokens of a wave live in ONE page (TW divides KV_PAGE, t_base is TW-aligned): a single wave-uniform
    // (scalar) page-table read, then 1 KiB-contiguous tile loads.  Neither depends on the current length: the page-table
    // slot exists for every launched chunk (unassigned slots hold a valid page id) and rows past the length are stale but
    // in bounds (the pools are zero-initialised, so always finite) -- they are masked below.  The length itself is read
    // in parallel instead of in front of the chain.
    static_assert(KV_PAGE % TW == 0, "a wave's tokens must not straddle a KV page");
    const int t_base_u = __builtin_amdgcn_readfirstlane(t_base);
    const int page = kv.page_table[t_base_u / KV_PAGE];
    const WT* kpage = reinterpret_cast<const WT*>(kv.k) + (size_t)(page * Hk + g) * KV_PAGE * DH;
    const WT* vpage = reinterpret_cast<const WT*>(kv.v) + (size_t)(page * Hk + g) * KV_PAGE * DH;
    vec kreg[NLD], vreg[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int tl = (i * 64 + lane) / LPT, sl = lane % LPT;
        const int t = t_base + tl;
        kreg[i] = *reinterpret_cast<const vec*>(kpage + (size_t)(t % KV_PAGE) * DH + sl * EPL);
        vreg[i] = *reinterpret_cast<const vec*>(vpage + (size_t)(t % KV_PAGE) * DH + sl * EPL);
    }
    const int gi = lane / LPT, sub = lane % LPT;
    const int rl = gi % NRP, ts = gi / NRP;
    float qr[NHP][EPL];
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {
        const float* qp = q + (size_t)(g * GH + hb + hp * NRP + rl) * DH + sub * EPL;
#pragma unroll
        for (int i = 0; i < EPL; ++i) qr[hp][i] = qp[i];
    }
    // q . (k^T * scale)  (dual_ar.rs:260).  For head_dim 64 the scale is 2^-3: scaling by a power of two commutes with every
    // f32 rounding, so folding it into q once is bit-identical to scaling each k element (8 multiplies per token saved).
    constexpr bool POW2 = (DH == 64 || DH == 16 || DH == 256);
    if (POW2) {
        const float s0 = 1.0f / sqrtf((float)DH);
#pragma unroll
        for (int hp = 0; hp < NHP; ++hp)
#pragma unroll
            for (int i = 0; i < EPL; ++i) qr[hp][i] *= s0;
    }
    FS_ISSUE_FENCE();
    const int T = row_pos(state, (int)blockIdx.y, pos_step) + 1;  // the row's own K/V were appended by the qkv stage
    if (c * CH >= T) return;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        *reinterpret_cast<vec*>(&sk[wave][(size_t)(i * 64 + lane) * EPL]) = kreg[i];
        *reinterpret_cast<vec*>(&sv[wave][(size_t)(i * 64 + lane) * EPL]) = vreg[i];
    }
    // (each wave reads only the tile it wrote: no block barrier needed, the compiler orders the LDS accesses)
    const float scale = 1.0f / sqrtf((float)DH);
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {
        float sc[TPG];
        float m = -1e30f;
#pragma unroll
        for (int j = 0; j < TPG; ++j) {
            const int tl = ts + j * NTS;
            float kf[EPL];
            WTr<WT>::unpack(*reinterpret_cast<const vec*>(&sk[wave][(size_t)tl * DH + sub * EPL]), kf);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) a = POW2 ? fmaf(qr[hp][i], kf[i], a) : fmaf(qr[hp][i], kf[i] * scale, a);
            a = group_sum<LPT>(a);
            sc[j] = (t_base + tl < T) ? a : -1e30f;
            m = fmaxf(m, sc[j]);
        }
        float l = 0.f, o[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) o[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TPG; ++j) {
            const int tl = ts + j * NTS;
            const float p = (t_base + tl < T) ? __expf(sc[j] - m) : 0.f;
            l += p;
            float vf[EPL];
            WTr<WT>::unpack(*reinterpret_cast<const vec*>(&sv[wave][(size_t)tl * DH + sub * EPL]), vf);
#pragma unroll
            for (int i = 0; i < EPL; ++i) o[i] = fmaf(p, vf[i], o[i]);
        }
        float* dst = sp[wave][hp * NRP + rl][ts];
#pragma unroll
        for (int i = 0; i < EPL; ++i) dst[sub * EPL + i] = o[i];
        if (sub == 0) { dst[DH] = m; dst[DH + 1] = l; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NREP * DH; e += NW * 64) {
        const int r = e / DH, dd = e % DH;
        float mn = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int k = 0; k < NTS; ++k) mn = fmaxf(mn, sp[w][r][k][DH]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int k = 0; k < NTS; ++k) {
                const float cf = __expf(sp[w][r][k][DH] - mn);
                L += sp[w][r][k][DH + 1] * cf;
                O += sp[w][r][k][dd] * cf;
            }
        float* dst = part + ((size_t)(g * GH + hb + r) * n_chunks_max + c) * (DH + 2);
        dst[dd] = O;
        if (dd == 0) { dst[DH] = mn; dst[DH + 1] = L; }
    }
}

// ------------------------------------------------------------------------------------------------ wo + residual
// Prologue (whole block, result in LDS): either combine the per-chunk partials of k_attn_decode, or -- FUSED, used by
// the fast decoder whose KV length is <= 8 -- run the whole attention for all heads redundantly per block.
// Body: one wave per output row: x[r] += Wo[r,:] . attn   (dual_ar.rs:383,437)
template <typename WT, int K, int WAVES, bool FUSED, int DH, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_wo(const float* __restrict__ part, int n_chunks_max, int chunk,
                                                   const SeqState* __restrict__ state, const float* __restrict__ q, KVView kv,
                                                   int fused_T, const WT* __restrict__ W, float* __restrict__ x, int H, int Hk,
                                                   int n_rows, const float* __restrict__ wscale, int nc_launch) {
    using R = Row<WT, K, NT>;
    using KT = KVT<WT>;
    constexpr int NTH = WAVES * 64;
    constexpr int EPL = WTr<KT>::EPL;   // the prologue reads the KV cache (KT), the body streams W (WT)
    using vec = typename WTr<KT>::vec;
    static_assert(K % 4 == 0 && DH % 16 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) float attn[K];
    __shared__ float wl[FUSED ? 32 * 8 : 32 * 128];
    __shared__ float ml[FUSED ? 2 : 2 * 32 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    const int n_rep = H / Hk;
    typename R::vec wv[R::NCH];
    const float xres = (row < n_rows && lane == 0) ? x[row] : 0.f;
    const float sc = row_scale<WT>(wscale, min(row, n_rows - 1));  // fp8: in front of the weight stream
    // Few, wide prologue loads (the texture-address unit retires one wave-instruction per ~16 cycles whatever its width),
    // all issued before the weight stream (vmcnt retires in issue order).
    if (!FUSED && nc_launch <= 8) {
        // short sequences (<= 8 chunks): every thread merges the chunks of its own head in registers -- its {m, l} pairs and its
        // four output dims of every chunk are 16 independent loads issued before the weight stream; no LDS hop, no barrier
        // until the result vector is complete (same arithmetic, in the same order, as the general path below)
        const int e4 = threadIdx.x * 4;
        const bool has_o = e4 < K;
        const int ho = (has_o ? e4 : 0) / DH, ddo = e4 % DH;
        const float* pb = part + (size_t)ho * n_chunks_max * (DH + 2);
        float2 mlv[8];
        float4 ov[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cc = min(c, nc_launch - 1);
            mlv[c] = *reinterpret_cast<const float2*>(pb + (size_t)cc * (DH + 2) + DH);
            ov[c] = *reinterpret_cast<const float4*>(pb + (size_t)cc * (DH + 2) + ddo);
        }
        R::load_w(W + (size_t)min(row, n_rows - 1) * K, lane, wv);
        FS_ISSUE_FENCE();
        const int T = state->pos + 1;
        const int nc = (T + chunk - 1) / chunk;  // chunks k_attn_decode produced (<= nc_launch)
        float mn = -1e30f;
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < nc) mn = fmaxf(mn, mlv[c].x);
        float L = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < nc) L += mlv[c].y * __expf(mlv[c].x - mn);
        const float inv = 1.f / L;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < nc) {
                const float wj = __expf(mlv[c].x - mn) * inv;
                acc.x = fmaf(wj, ov[c].x, acc.x); acc.y = fmaf(wj, ov[c].y, acc.y);
                acc.z = fmaf(wj, ov[c].z, acc.z); acc.w = fmaf(wj, ov[c].w, acc.w);
            }
        if (has_o) *reinterpret_cast<float4*>(&attn[e4]) = acc;
    } else if (!FUSED) {
        // every prologue load is addressed from launch-time constants only (nc_launch = chunks this graph bucket launches),
        // so nothing waits for state->pos before the weight stream is requested; chunks >= nc hold stale (finite) partials of
        // earlier frames and are masked below
        // (a) {m, l} of every (head, chunk) + the first four chunks' o values (float4 per thread per chunk)
        const int e0 = threadIdx.x;
        const int eh = e0 / nc_launch, ec = e0 % nc_launch;
        const bool in_ml = e0 < H * nc_launch;
        const float2 mv = *reinterpret_cast<const float2*>(part + ((size_t)(in_ml ? eh : 0) * n_chunks_max + ec) * (DH + 2) + DH);
        const int e4 = threadIdx.x * 4;  // this thread's 4 consecutive attn elements (K <= 4 * NTH for every supported K here)
        const bool has_o = e4 < K;
        const int ho = (has_o ? e4 : 0) / DH, ddo = e4 % DH;
        const float* po = part + (size_t)ho * n_chunks_max * (DH + 2) + ddo;
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(po + (size_t)min(j, nc_launch - 1) * (DH + 2));
        R::load_w(W + (size_t)min(row, n_rows - 1) * K, lane, wv);  // clamped, not predicated: a predicated load costs a wait
        FS_ISSUE_FENCE();
        const int T = state->pos + 1;
        const int nc = (T + chunk - 1) / chunk;  // chunks k_attn_decode produced (<= nc_launch)
        if (in_ml && ec < nc) { ml[2 * (eh * 128 + ec)] = mv.x; ml[2 * (eh * 128 + ec) + 1] = mv.y; }
        for (int e = threadIdx.x + NTH; e < H * nc_launch; e += NTH) {  // only when H * nc_launch > NTH (very long sequences)
            const int h = e / nc_launch, c = e % nc_launch;
            if (c >= nc) continue;
            const float2 t2 = *reinterpret_cast<const float2*>(part + ((size_t)h * n_chunks_max + c) * (DH + 2) + DH);
            ml[2 * (h * 128 + c)] = t2.x;
            ml[2 * (h * 128 + c) + 1] = t2.y;
        }
        __syncthreads();
        // (b) per head: softmax-merge weights wl[h][c] = exp(m_c - M) / sum_c l_c exp(m_c - M)
        for (int h = threadIdx.x; h < H; h += NTH) {
            float mn = -1e30f;
            for (int c = 0; c < nc; ++c) mn = fmaxf(mn, ml[2 * (h * 128 + c)]);
            float L = 0.f;
            for (int c = 0; c < nc; ++c) L += ml[2 * (h * 128 + c) + 1] * __expf(ml[2 * (h * 128 + c)] - mn);
            const float inv = 1.f / L;
            for (int c = 0; c < nc; ++c) wl[h * 128 + c] = __expf(ml[2 * (h * 128 + c)] - mn) * inv;
        }
        __syncthreads();
        // (c) attn[e] = sum_c wl[h][c] * o[h][c][dd]
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = 0; c0 < nc; c0 += 4) {
            if (c0 > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = (has_o && c0 + j < nc) ? *reinterpret_cast<const float4*>(po + (size_t)(c0 + j) * (DH + 2))
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < nc) {
                    const float wj = wl[ho * 128 + c0 + j];
                    acc.x = fmaf(wj, v[j].x, acc.x); acc.y = fmaf(wj, v[j].y, acc.y);
                    acc.z = fmaf(wj, v[j].z, acc.z); acc.w = fmaf(wj, v[j].w, acc.w);
                }
        }
        if (has_o) *reinterpret_cast<float4*>(&attn[e4]) = acc;
    } else if (DH == 64 && K == 4 * NTH && H * 16 == NTH) {
        // Fast decoder, Fish geometry (16 heads x 64 dims, 256 threads): head h is produced AND consumed by the 16 lanes
        // tid = 16h .. 16h+15 (one DPP row of one wave): scores[h][t] by lane pair (2t, 2t+1) (32 dims each), then every lane of
        // the row does softmax . V for 4 of the head's 64 dims.  The score exchange stays inside the wave (8 floats per head in
        // LDS, no block barrier: a wave's LDS operations execute in order) -- one __syncthreads() less than the general path.
        const float scale = 1.0f / sqrtf((float)DH);
        const KT* kbase = reinterpret_cast<const KT*>(kv.k);
        const KT* vbase = reinterpret_cast<const KT*>(kv.v);
        constexpr int QD = DH / 2;
        const int hh = threadIdx.x >> 4, t1 = (threadIdx.x >> 1) & 7, sl = threadIdx.x & 1;
        const int g = hh / n_rep;
        vec kvv[QD / EPL];
        float4 qv[QD / 4];
        {
            const KT* kp = kbase + ((size_t)g * KV_PAGE + t1) * DH + sl * QD;  // rows t >= fused_T are stale but in bounds; masked below
#pragma unroll
            for (int i = 0; i < QD / EPL; ++i) kvv[i] = *reinterpret_cast<const vec*>(kp + i * EPL);
#pragma unroll
            for (int i = 0; i < QD / 4; ++i) qv[i] = *reinterpret_cast<const float4*>(q + hh * DH + sl * QD + i * 4);
        }
        const int ddv = (threadIdx.x & 15) * 4;  // this lane's 4 output dims of head hh
        float vf4[8][4];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const KT* vp = vbase + ((size_t)g * KV_PAGE + t) * DH + ddv;
#pragma unroll
            for (int i = 0; i < 4; ++i) vf4[t][i] = WTr<KT>::to_f32(vp[i]);
        }
        R::load_w(W + (size_t)min(row, n_rows - 1) * K, lane, wv);
        FS_ISSUE_FENCE();
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < QD / EPL; ++i) {
                float kf[EPL];
                WTr<KT>::unpack(kvv[i], kf);
#pragma unroll
                for (int j = 0; j < EPL; j += 4) {
                    const float4 qq = qv[(i * EPL + j) / 4];
                    acc = fmaf(qq.x, kf[j] * scale, acc); acc = fmaf(qq.y, kf[j + 1] * scale, acc);
                    acc = fmaf(qq.z, kf[j + 2] * scale, acc); acc = fmaf(qq.w, kf[j + 3] * scale, acc);
                }
            }
            acc = group_sum<2>(acc);
            if (sl == 0) wl[hh * 8 + t1] = acc;
        }
        // (same wave: no block barrier)
        float mn = -1e30f;
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < fused_T) mn = fmaxf(mn, wl[hh * 8 + t]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < fused_T) {
                const float p = __expf(wl[hh * 8 + t] - mn);
                L += p;
#pragma unroll
                for (int i = 0; i < 4; ++i) O[i] = fmaf(p, vf4[t][i], O[i]);
            }
        const float inv = 1.f / L;
        *reinterpret_cast<float4*>(&attn[threadIdx.x * 4]) = make_float4(O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv);
    } else {
        // Fast decoder: <= 8 cached tokens, all in page 0 of this layer's private KV page (no page-table lookup).
        // step 1: scores[h][t], TWO threads per (h, t) (DH/2 dims each, DPP pair-sum); step 2: softmax . V with one
        // thread per EPL consecutive dims (one 16-B V load per token).
        const float scale = 1.0f / sqrtf((float)DH);
        const KT* kbase = reinterpret_cast<const KT*>(kv.k);
        const KT* vbase = reinterpret_cast<const KT*>(kv.v);
        constexpr int QD = DH / 2;  // dims per step-1 thread
        const int e1 = threadIdx.x >> 1, sl = threadIdx.x & 1;
        const bool has_s = e1 < H * fused_T;
        const int h1 = has_s ? e1 / fused_T : 0, t1 = has_s ? e1 % fused_T : 0;
        vec kvv[QD / EPL];
        float4 qv[QD / 4];
        {
            const KT* kp = kbase + ((size_t)(h1 / n_rep) * KV_PAGE + t1) * DH + sl * QD;
#pragma unroll
            for (int i = 0; i < QD / EPL; ++i) kvv[i] = *reinterpret_cast<const vec*>(kp + i * EPL);
#pragma unroll
            for (int i = 0; i < QD / 4; ++i) qv[i] = *reinterpret_cast<const float4*>(q + h1 * DH + sl * QD + i * 4);
        }
        const int ev = threadIdx.x * EPL;  // this thread's EPL consecutive attn elements
        const bool has_v = ev < K;
        const int hv = (has_v ? ev : 0) / DH, ddv = ev % DH;
        vec vvv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)  // unconditional (rows t >= fused_T of the page are stale but in bounds; masked at use)
            vvv[t] = *reinterpret_cast<const vec*>(vbase + ((size_t)(hv / n_rep) * KV_PAGE + t) * DH + ddv);
        R::load_w(W + (size_t)min(row, n_rows - 1) * K, lane, wv);
        FS_ISSUE_FENCE();
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < QD / EPL; ++i) {
                float kf[EPL];
                WTr<KT>::unpack(kvv[i], kf);
#pragma unroll
                for (int j = 0; j < EPL; j += 4) {
                    const float4 qq = qv[(i * EPL + j) / 4];
                    acc = fmaf(qq.x, kf[j] * scale, acc); acc = fmaf(qq.y, kf[j + 1] * scale, acc);
                    acc = fmaf(qq.z, kf[j + 2] * scale, acc); acc = fmaf(qq.w, kf[j + 3] * scale, acc);
                }
            }
            acc = group_sum<2>(acc);
            if (has_s && sl == 0) wl[h1 * 8 + t1] = acc;
        }
        __syncthreads();
        if (has_v) {
            float mn = -1e30f;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t < fused_T) mn = fmaxf(mn, wl[hv * 8 + t]);
            float L = 0.f, O[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) O[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < fused_T) {
                    const float p = __expf(wl[hv * 8 + t] - mn);
                    L += p;
                    float vf[EPL];
                    WTr<KT>::unpack(vvv[t], vf);
#pragma unroll
                    for (int i = 0; i < EPL; ++i) O[i] = fmaf(p, vf[i], O[i]);
                }
            const float inv = 1.f / L;
#pragma unroll
            for (int i = 0; i < EPL; ++i) attn[ev + i] = O[i] * inv;
        }
    }
    __syncthreads();
    if (row >= n_rows) return;
    float xr[R::NX];
    R::load_x(attn, lane, xr);
    const float d = wave_sum(R::dot(wv, xr)) * sc;
    if (lane == 0) x[row] = xres + d;
}

// ------------------------------------------------------------------------------------------------ SwiGLU up
// One wave per (w1[r], w3[r]) pair of the row-interleaved W13: act[r] = silu(w1[r].xn) * (w3[r].xn)
template <typename WT, int K, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_ffn_up(const float* __restrict__ x, const float* __restrict__ norm_w,
                                                       float eps, const WT* __restrict__ W13, float* __restrict__ act,
                                                       int inter, const float* __restrict__ wscale) {
    using R = Row<WT, K, NT>;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (r >= inter) return;
    float xr[R::NX], nr[R::NX];
    R::load_x(x, lane, xr);
    R::load_x(norm_w, lane, nr);
    const float s1 = row_scale<WT>(wscale, 2 * r), s3 = row_scale<WT>(wscale, 2 * r + 1);  // fp8: in front of the weight stream
    typename R::vec w1[R::NCH], w3[R::NCH];
    R::load_w(W13 + (size_t)(2 * r) * K, lane, w1);
    R::load_w(W13 + (size_t)(2 * r + 1) * K, lane, w3);
    FS_ISSUE_FENCE();
    R::rmsnorm(xr, nr, eps);
    const float a = wave_sum(R::dot(w1, xr)) * s1;
    const float b = wave_sum(R::dot(w3, xr)) * s3;
    if (lane == 0) act[r] = (a / (1.f + __expf(-a))) * b;  // candle silu = x / (1 + exp(-x))
}

// ------------------------------------------------------------------------------------------------ down + residual
// x[r] += W2[r,:] . act with K = inter split over the KS waves of a block (one row per block): every wave streams
// K/KS contiguous weights against its own slice of `act` (registers, straight from L2), the KS partial sums meet in LDS
// and are added in a fixed order.
template <typename WT, int K, int KS, bool NT>
__global__ __launch_bounds__(KS * 64) void k_ffn_down(const float* __restrict__ act, const WT* __restrict__ W2,
                                                      float* __restrict__ x, int n_rows, const float* __restrict__ wscale) {
    using R = Row<WT, K / KS, NT>;
    __shared__ float red[KS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const float xres = (threadIdx.x == 0) ? x[row] : 0.f;
    const float sc = row_scale<WT>(wscale, row);  // fp8: in front of the weight stream
    float xr[R::NX];
    R::load_x(act + wave * (K / KS), lane, xr);
    typename R::vec wv[R::NCH];
    R::load_w(W2 + (size_t)row * K + wave * (K / KS), lane, wv);
    FS_ISSUE_FENCE();
    const float d = wave_sum(R::dot(wv, xr));
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int i = 1; i < KS; ++i) t += red[i];
        x[row] = xres + t * sc;
    }
}

// ------------------------------------------------------------------------------------------------ norm + head GEMV
template <typename WT, int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_head(const float* __restrict__ x, const float* __restrict__ norm_w, float eps,
                                                     const WT* __restrict__ W, int n_rows, float* __restrict__ logits,
                                                     const float* __restrict__ wscale) {
    using R = Row<WT, K>;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    float xr[R::NX], nr[R::NX];
    R::load_x(x, lane, xr);
    R::load_x(norm_w, lane, nr);
    const float sc = row_scale<WT>(wscale, r);  // fp8: in front of the weight stream
    typename R::vec wv[R::NCH];
    R::load_w(W + (size_t)r * K, lane, wv);
    FS_ISSUE_FENCE();
    R::rmsnorm(xr, nr, eps);
    const float d = wave_sum(R::dot(wv, xr)) * sc;
    if (lane == 0) logits[r] = d;
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

template <typename WT>
__global__ void k_gather_rows(const WT* __restrict__ src, int dim, uint32_t r0, uint32_t r1, WT* __restrict__ dst) {
    for (int d = threadIdx.x; d < dim; d += blockDim.x) {
        dst[d] = src[(size_t)r0 * dim + d];
        dst[dim + d] = src[(size_t)r1 * dim + d];
    }
}

__global__ void k_advance(SeqState* state) {
    state->pos += 1;
    state->step += 1;
}

// ------------------------------------------------------------------------------------------------ skinny GEMMs (MFMA)
// Activation ROWS against bf16 weights: the prefill (rows = up to 512 consecutive prompt tokens of one sequence per pass) and the
// batched decode step (rows = sequences).  Y[M, N] = f(A[M, K]) . W[N, K]^T, 32 rows (one MFMA column-tile pair) at a time.
// Numerics: activations stay f32-grade end to end -- every GEMM input is split once into bf16 hi + bf16 lo parts
// (a = hi + lo up to 2^-17 relative; k_prep / producer epilogues) and both parts go through v_mfma_f32_16x16x32_bf16 with
// f32 accumulation, so the MFMA path agrees with the f32-activation GEMV path far below bf16 resolution (the reference's
// own CUDA path rounds activations to bf16).  Kernel: k_gemm3 below (A operand = weights straight from global, non-temporal;
// B operand = fragment-major activations straight from L2; no LDS staging).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int PF_M = 32;    // activation rows per pass

enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_QKV = 3, EPI_RESIDUAL_NORM = 4, EPI_SWIGLU_RMS = 5, EPI_QKV_RMS = 6, EPI_STORE_RMS = 7 };
// epilogues that divide the K-summed accumulators by the row's rms (the producer left split(x * g) and sum-of-squares partials: NormAux)
__host__ __device__ constexpr bool epi_rms(int e) { return e == EPI_SWIGLU_RMS || e == EPI_QKV_RMS || e == EPI_STORE_RMS; }
// RMSNorm folded into the GEMMs around it (saves the k_prep node between Wo and W13): the Wo GEMM's epilogue writes the new
// residual stream, the UN-normalised GEMM input split(x * g) and one sum-of-squares partial per (block, row); the W13 GEMM's
// epilogue divides its accumulators by rms[m] = sqrt(sum_b partial[m][b] / D + eps) (a per-row scalar commutes with the GEMM).
struct NormAux {
    const float* g;   // norm weight [D]                       (EPI_RESIDUAL_NORM)
    float* ss;        // partials [Mcap][nblk], row-major       (written by RESIDUAL_NORM, read by SWIGLU_RMS)
    bf16_t* A2;       // fragment-major output of RESIDUAL_NORM
    int nblk;         // blocks of the producing GEMM (multiple of 8)
    int D;
    float eps;
};

struct RowMap {       // how activation row m maps onto sequences / positions
    int pos_step;     // prefill: 1 (row m = token at pos + m of ONE sequence); batched decode: 0 (every row at pos)
    int pt_stride;    // prefill: 0 (one page table); batched decode: page-table stride between sequences
    int seq_rows;     // > 0: group prefill -- row m = token (m % seq_rows) of sequence (m / seq_rows), page tables pt_stride apart
};

__device__ __forceinline__ void split_bf16(float a, bf16_t& hi, bf16_t& lo) {
    hi = WTr<bf16_t>::from_f32(a);
    lo = WTr<bf16_t>::from_f32(a - WTr<bf16_t>::to_f32(hi));
}

// GEMM-input layout ("fragment-major"): the bf16 hi/lo activations are stored exactly as the MFMA B operand wants them, so
// that every wave-wide operand load of k_gemm3 is ONE contiguous KiB (8 full cache lines) instead of 16 half lines strided
// by a row (measured: the strided form cost 3.3 us of a 7 us GEMM node).  Element (row m, depth k, part hi=0 / lo=1) of a
// [rows][K] activation matrix lives at (in bf16 elements)
//     (((((m / 32) * (K / 32) + k / 32) * 2 + part) * 2 + (m / 16) % 2) * 64 + ((k / 8) % 4) * 16 + m % 16) * 8 + k % 8
// i.e. [32-row panel][32-deep k-step][part][16-row tile][lane = kq*16 + row][8 consecutive k].
__device__ __forceinline__ size_t frag_off(int m, int k, int part, int K) {
    const int p = m >> 5, mt = (m >> 4) & 1, lr = m & 15, kk = k >> 5, lq = (k >> 3) & 3, e = k & 7;
    return (((((size_t)p * (K >> 5) + kk) * 2 + part) * 2 + mt) * 64 + (lq * 16 + lr)) * 8 + e;
}

// X[m] (+= sum_s P[s][m], fixed order) ; optional RMSNorm ; emit bf16 hi/lo (fragment-major).  One block per row.
__global__ __launch_bounds__(256) void k_prep(float* __restrict__ X, int D, const float* __restrict__ P, int S, size_t slab_stride,
                                              const float* __restrict__ norm_w, float eps, bf16_t* __restrict__ Ohi) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    float* xm = X + (size_t)m * D;
    float ss = 0.f;
    // D <= 1024 (every row-path model here): a thread's one float4 stays in registers between the two phases -- the second phase used to
    // read the row back from memory behind the block barrier, one more dependent L2 round trip in a node that is nothing but latency
    const bool one = D <= 1024;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f), wkeep = make_float4(1.f, 1.f, 1.f, 1.f);
    if (one && norm_w && threadIdx.x * 4 < D) wkeep = *reinterpret_cast<const float4*>(norm_w + threadIdx.x * 4);  // (requested with the row, not behind the barrier)
    for (int e = threadIdx.x * 4; e < D; e += 1024) {
        float4 v = *reinterpret_cast<const float4*>(xm + e);
        for (int sI = 0; sI < S; ++sI) {
            const float4 p = *reinterpret_cast<const float4*>(P + (size_t)sI * slab_stride + (size_t)m * D + e);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (S > 0) *reinterpret_cast<float4*>(xm + e) = v;
        ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        keep = v;
    }
    if (!Ohi) return;
    float d = 1.f;
    if (norm_w) {
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        d = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)D + eps);
    }
    for (int e = threadIdx.x * 4; e < D; e += 1024) {
        const float4 v = one ? keep : *reinterpret_cast<const float4*>(xm + e);  // (else: own earlier write, same thread)
        float a[4] = {v.x, v.y, v.z, v.w};
        if (norm_w) {
            const float4 w = one ? wkeep : *reinterpret_cast<const float4*>(norm_w + e);
            a[0] = (a[0] / d) * w.x; a[1] = (a[1] / d) * w.y; a[2] = (a[2] / d) * w.z; a[3] = (a[3] / d) * w.w;
        }
        bf16_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_bf16(a[i], hi[i], lo[i]);
        uint2 ph, pl;
        ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
        pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e, 0, D)) = ph;
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e, 1, D)) = pl;
    }
}

// k_prep's RMSNorm + hi/lo split of ONE row by the block that has just written it (the batched samplers: the row is the fast decoder's
// next input, so its first layer needs no k_prep node in a folded decode step).  Threads 0..255 take one float4 each (D <= 1024) and the sums
// meet in k_prep's order: the fragments are bit-identical to what the node would have produced.
struct PrepOut { const float* g; float eps; bf16_t* A; uint32_t* epoch; };   // g == nullptr: disabled; epoch != nullptr (slow-token sampler): the step epoch of k_gemm_down's tags is bumped here, once per step
__device__ __forceinline__ void block_prep_row(const float* __restrict__ xm, int D, const PrepOut& po, int m, float* red4) {
    __syncthreads();  // the row's stores by the other threads of this block
    const int e = threadIdx.x * 4;
    const bool act = threadIdx.x < 256 && e < D;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act) { v = *reinterpret_cast<const float4*>(xm + e); w = *reinterpret_cast<const float4*>(po.g + e); }
    float ss = 0.f;
    ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
    ss = wave_sum(ss);
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (!act) return;
    const float d = sqrtf(((red4[0] + red4[1]) + (red4[2] + red4[3])) / (float)D + po.eps);
    float a[4] = {(v.x / d) * w.x, (v.y / d) * w.y, (v.z / d) * w.z, (v.w / d) * w.w};
    bf16_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_bf16(a[i], hi[i], lo[i]);
    uint2 ph, pl;
    ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
    pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
    *reinterpret_cast<uint2*>(po.A + frag_off(m, e, 0, D)) = ph;
    *reinterpret_cast<uint2*>(po.A + frag_off(m, e, 1, D)) = pl;
}

// Epilogue of the row-path GEMMs for one (row pair r/r+1, activation row m): a, b are the K-summed (and fp8-scaled) dot products.
// Thread mapping contract: the ROWS / 2 threads of one m are adjacent lanes (RESIDUAL_NORM's sum of squares meets by shuffles).
struct GemmEpi {
    float* Y; int ldy; size_t slab_stride; bf16_t* Of; int ldo;
    const float *cos_t, *sin_t;
    KVView kv; int H, Hk, Dh; RowMap rm; NormAux na;
    int pos0, rope_off, N;
    const SeqState* states = nullptr;  // pos_step < 0: row m reads its own position (session slots)
    bf16_t* o1 = nullptr;              // EPI_QKV*: every row sits at position 0 of an empty cache (the fast decoder's first pass): attention over one
                                       // token is the identity on V, so the V rows also leave the attention output (hi = the cached bf16 value, lo = 0:
                                       // exactly what the attention node computes for T = 1) in fragment-major order and that node is not launched
};
// values the epilogue needs from memory that do not depend on the GEMM: requested at kernel entry by k_gemm3 (one slot per thread, first
// panel) so that they arrive under the weight stream instead of starting a dependent L2 round trip behind the K loop
struct EpiPre { float2 o; float2 g; float cs, sn; };
template <int EPI, int ROWS>
__device__ __forceinline__ void gemm_epilogue(float a, float b, int r, int m, int ml, int pr, const GemmEpi& g, const float* s_rms,
                                              const EpiPre* pre = nullptr) {
    float* const Y = g.Y; const int ldy = g.ldy, ldo = g.ldo, N = g.N, H = g.H, Hk = g.Hk, Dh = g.Dh, pos0 = g.pos0, rope_off = g.rope_off;
    const size_t slab_stride = g.slab_stride;
    bf16_t* const Of = g.Of;
    const float *cos_t = g.cos_t, *sin_t = g.sin_t;
    const KVView& kv = g.kv; const RowMap& rm = g.rm; const NormAux& na = g.na;
    if (EPI == EPI_QKV_RMS || EPI == EPI_STORE_RMS) { const float dn = s_rms[ml]; a /= dn; b /= dn; }  // (a per-row scalar commutes with the GEMM)
    if (EPI == EPI_STORE || EPI == EPI_STORE_RMS) {  // split-K slab blockIdx.y
        float* yp = Y + (size_t)blockIdx.y * slab_stride + (size_t)m * ldy + r;
        yp[0] = a;
        if (r + 1 < N) yp[1] = b;
    } else if (EPI == EPI_RESIDUAL) {
        float2* yp = reinterpret_cast<float2*>(Y + (size_t)m * ldy + r);
        const float2 o = pre ? pre->o : *yp;
        *yp = make_float2(o.x + a, o.y + b);
    } else if (EPI == EPI_RESIDUAL_NORM) {  // the ROWS/2 threads of one m are adjacent lanes (8, 16 or 32 of them)
        float2* yp = reinterpret_cast<float2*>(Y + (size_t)m * ldy + r);
        const float2 o = pre ? pre->o : *yp;
        const float v0 = o.x + a, v1 = o.y + b;
        *yp = make_float2(v0, v1);
        bf16_t h0, l0, h1, l1;
        const float2 gw = pre ? pre->g : make_float2(na.g[r], na.g[r + 1]);
        split_bf16(v0 * gw.x, h0, l0); split_bf16(v1 * gw.y, h1, l1);
        *reinterpret_cast<uint32_t*>(na.A2 + frag_off(m, r, 0, na.D)) = h0 | ((uint32_t)h1 << 16);
        *reinterpret_cast<uint32_t*>(na.A2 + frag_off(m, r, 1, na.D)) = l0 | ((uint32_t)l1 << 16);
        float ssq = group_sum<(ROWS / 2 >= 16 ? 16 : ROWS / 2)>(fmaf(v0, v0, v1 * v1));
        if (ROWS / 2 == 32) ssq += __shfl_xor(ssq, 16, 64);
        if (pr == 0) na.ss[(size_t)m * na.nblk + blockIdx.x] = ssq;
    } else if (EPI == EPI_SWIGLU_RMS) {
        const float dn = s_rms[ml];
        const float an = a / dn, bn = b / dn;
        const float v = (an / (1.f + __expf(-an))) * bn;
        bf16_t hh, ll;
        split_bf16(v, hh, ll);
        Of[frag_off(m, r / 2, 0, ldo)] = hh;
        Of[frag_off(m, r / 2, 1, ldo)] = ll;
    } else if (EPI == EPI_SWIGLU) {  // interleaved rows (2r, 2r+1) = (w1[r], w3[r]) -> act hi/lo for the down GEMM
        const float v = (a / (1.f + __expf(-a))) * b;
        bf16_t h, l;
        split_bf16(v, h, l);
        Of[frag_off(m, r / 2, 0, ldo)] = h;
        Of[frag_off(m, r / 2, 1, ldo)] = l;
    } else {  // EPI_QKV: rope_i + scatter (q -> Y[m][r], k/v -> paged cache of row m's sequence)
        const int sq = rm.seq_rows > 0 ? m / rm.seq_rows : m;  // sequence of row m / its token index within the pass
        int pos = pos0 + (rm.seq_rows > 0 ? m - sq * rm.seq_rows : m * rm.pos_step), rpos = pos + rope_off;
        if (rm.pos_step < 0) { pos = g.states[m].pos; rpos = pos + g.states[m].rope_off; }
        const int* ptab = kv.page_table + (size_t)sq * rm.pt_stride;
        const int qdim = H * Dh, kdim = Hk * Dh, half = Dh / 2;
        if (r < qdim + kdim) {
            const int j = (r % Dh) / 2;
            const float cs = pre ? pre->cs : cos_t[(size_t)rpos * half + j], sn = pre ? pre->sn : sin_t[(size_t)rpos * half + j];
            const float o0 = a * cs - b * sn, o1 = a * sn + b * cs;
            if (r < qdim) { *reinterpret_cast<float2*>(Y + (size_t)m * ldy + r) = make_float2(o0, o1); }
            else {
                const int rk = r - qdim;
                bf16_t* dst = kv_addr<bf16_t>(kv.k, ptab, pos, rk / Dh, Hk, Dh) + rk % Dh;
                *reinterpret_cast<uint32_t*>(dst) = WTr<bf16_t>::from_f32(o0) | ((uint32_t)WTr<bf16_t>::from_f32(o1) << 16);
            }
        } else {
            const int rv = r - qdim - kdim;
            bf16_t* dst = kv_addr<bf16_t>(kv.v, ptab, pos, rv / Dh, Hk, Dh) + rv % Dh;
            const uint32_t vb2 = WTr<bf16_t>::from_f32(a) | ((uint32_t)WTr<bf16_t>::from_f32(b) << 16);
            *reinterpret_cast<uint32_t*>(dst) = vb2;
            if (g.o1) {
                const int n_rep = H / Hk, gk = rv / Dh, dd = rv % Dh;
                for (int hh = 0; hh < n_rep; ++hh) {
                    const int e = (gk * n_rep + hh) * Dh + dd;
                    *reinterpret_cast<uint32_t*>(g.o1 + frag_off(m, e, 0, qdim)) = vb2;
                    *reinterpret_cast<uint32_t*>(g.o1 + frag_off(m, e, 1, qdim)) = 0u;
                }
            }
        }
    }
}

// Block = 16*RT weight rows x one K range of 128*NKS (the 4 waves take a quarter each, NKS k-steps of 32) x ALL activation
// rows, 32 at a time.  No LDS staging and no barrier in front of the MFMAs: a wave puts its whole weight panel in flight at
// kernel entry (A operand, non-temporal, kept in VGPRs for every row panel), then per 32-row panel loads its B operands
// (hi + lo slices of the activations, 16 B per lane per slice, L2-resident) straight into VGPRs and runs NKS * RT * 4 MFMAs.
// The four K-quarter accumulators meet in 8 KB of LDS and are summed in a fixed order; the epilogue gives every thread
// one (row pair, m) so RoPE pairs and SwiGLU (w1[r], w3[r]) pairs stay in one lane.  Small grids were the problem of the
// first version of this kernel (64-row blocks: 20 blocks for Wqkv); 16-row blocks give 80 / 64 / 512 (256 with RT = 2) / 256.
// FP8: the weights are e4m3fn bytes + one f32 scale per row (FS_FP8 handles): a lane's 8 bytes per k-step are widened to bf16 in
// registers (exact: e4m3 has 3 mantissa bits) and the row scale is applied to the K-summed accumulator in the epilogue.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 fp8x8_to_bf16x8(u32x2 v) {
    u32x4 o;
    const uint32_t w[2] = {v.x, v.y};
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
        r[2 * i] = (__float_as_uint(lo.x) >> 16) | (__float_as_uint(lo.y) & 0xFFFF0000u);
        r[2 * i + 1] = (__float_as_uint(hi.x) >> 16) | (__float_as_uint(hi.y) & 0xFFFF0000u);
    }
    o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
    return o;
}
// HALF: M <= 16 (small static batches) -- rows 16..31 of the only panel do not exist: their operand loads and MFMAs are skipped.
template <int EPI, int NKS, int RT, bool FP8, bool HALF = false>
__global__ __launch_bounds__(256) void k_gemm3(const bf16_t* __restrict__ Xf, int M, int K,
                                               const void* __restrict__ Wv, const float* __restrict__ wscale, int N, float* __restrict__ Y, int ldy, size_t slab_stride,
                                               bf16_t* __restrict__ Of, int ldo,
                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                               const SeqState* __restrict__ state, KVView kv, int H, int Hk, int Dh, RowMap rm, NormAux na) {
    constexpr int ROWS = 16 * RT;
    __shared__ float red[4][ROWS][33];
    __shared__ float s_rms[PF_M];
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int n0 = blockIdx.x * ROWS;
    const int kbeg = ((int)blockIdx.y * 4 + kq) * NKS * 32 + (lane >> 4) * 8;  // this lane's first k of every 32-wide step (weights)
    int pos0 = 0, rope_off = 0;
    if (EPI == EPI_QKV || EPI == EPI_QKV_RMS) { pos0 = state->pos; rope_off = state->rope_off; }  // requested up front: the epilogue must not start a dependent chain
    const GemmEpi ge{Y, ldy, slab_stride, Of, ldo, cos_t, sin_t, kv, H, Hk, Dh, rm, na, pos0, rope_off, N, state,
                     (EPI == EPI_QKV || EPI == EPI_QKV_RMS) ? Of : nullptr};  // (Of of a QKV launch: GemmEpi::o1)
    u32x4 wf[RT][NKS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const size_t woff = (size_t)min(n0 + rt * 16 + (lane & 15), N - 1) * K + kbeg;
        if (FP8) {
            const uint8_t* wp = reinterpret_cast<const uint8_t*>(Wv) + woff;
            u32x2 raw[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) raw[ks] = ld_stream(reinterpret_cast<const u32x2*>(wp + ks * 32));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[rt][ks] = fp8x8_to_bf16x8(raw[ks]);
        } else {
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wv) + woff;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[rt][ks] = ld_stream(reinterpret_cast<const u32x4*>(wp + ks * 32));
        }
    }
    // one epilogue slot per thread (RT == 1): what that slot reads from memory is requested now, behind the weight stream (first panel)
    constexpr bool PRE_RES = RT == 1 && (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_NORM), PRE_QKV = RT == 1 && (EPI == EPI_QKV || EPI == EPI_QKV_RMS);
    EpiPre pre{make_float2(0.f, 0.f), make_float2(1.f, 1.f), 1.f, 0.f};
    const int pre_r = n0 + 2 * ((int)threadIdx.x % (ROWS / 2)), pre_m = (int)blockIdx.z * PF_M + (int)threadIdx.x / (ROWS / 2);
    bool use_pre = PRE_RES;
    if (PRE_RES && pre_r < N && pre_m < M) {
        pre.o = *reinterpret_cast<const float2*>(Y + (size_t)pre_m * ldy + pre_r);
        if (EPI == EPI_RESIDUAL_NORM) pre.g = *reinterpret_cast<const float2*>(na.g + pre_r);
    }
    if (PRE_QKV) use_pre = rm.pos_step == 0 && rm.seq_rows == 0;  // lock-step decode rows: every row at state->pos (block-uniform)
    // row panels are spread over blockIdx.z (prefill: many panels -> more blocks; the weight tile is then re-read from L2)
    bool first_panel = true;
    for (int mp = (int)blockIdx.z * PF_M; mp < M; mp += (int)gridDim.z * PF_M) {
        float ss8 = 0.f;
        if (epi_rms(EPI)) {  // this panel's 32 row norms: thread (m = tid/8, j = tid%8) sums every 8th block's partial, then 8 lanes meet
            const float* sp = na.ss + (size_t)(mp + (threadIdx.x >> 3)) * na.nblk + (threadIdx.x & 7);
            for (int q8 = 0; q8 < na.nblk; q8 += 8) ss8 += sp[q8];
        }
        // B operands of this panel (fragment-major: [k-step][hi, lo][16-row tile] blocks of one KiB each, lane-major)
        const bf16_t* xp = Xf + ((size_t)(mp >> 5) * (K >> 5) + ((int)blockIdx.y * 4 + kq) * NKS) * 2048 + lane * 8;
        u32x4 xf[NKS][4];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            xf[ks][0] = *reinterpret_cast<const u32x4*>(xp + ks * 2048);          // hi, rows 0..15
            xf[ks][1] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 1024);   // lo, rows 0..15
            if (!HALF) {
                xf[ks][2] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 512);    // hi, rows 16..31
                xf[ks][3] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 1536);   // lo, rows 16..31
            }
        }
        if (PRE_QKV && first_panel && use_pre && pre_r < (H + Hk) * Dh) {  // (waits for the scalar position load; the operand loads are already out)
            const int j = (pre_r % Dh) / 2;
            pre.cs = cos_t[(size_t)(pos0 + rope_off) * (Dh / 2) + j]; pre.sn = sin_t[(size_t)(pos0 + rope_off) * (Dh / 2) + j];
        }
        FS_ISSUE_FENCE();  // every operand load of the panel is in flight before the first MFMA waits (the scheduler would
                           // otherwise meter them out a dozen at a time, one memory round trip per batch)
        f32x4v acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, wf[rt][ks]);
#pragma unroll
                for (int j = 0; j < (HALF ? 2 : 4); ++j)
                    acc[rt][j >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, xf[ks][j]), acc[rt][j >> 1], 0, 0, 0);
            }
        if (!first_panel) __syncthreads();  // the previous panel's epilogue has read `red` (and s_rms)
        first_panel = false;
        if (epi_rms(EPI)) {
            const float tot = group_sum<8>(ss8);
            if ((threadIdx.x & 7) == 0) s_rms[threadIdx.x >> 3] = sqrtf(tot / (float)na.D + na.eps);
        }
        // lane holds C[row = rt*16 + (lane>>4)*4 + i][m = mt*16 + (lane&15)]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4v c = acc[rt][mt];
                float* rp = &red[kq][rt * 16 + (lane >> 4) * 4][mt * 16 + (lane & 15)];
                rp[0] = c.x; rp[33] = c.y; rp[66] = c.z; rp[99] = c.w;
            }
        __syncthreads();
        // epilogue: slot = (row pair, m); K-quarters summed in a fixed order
#pragma unroll
        for (int sI = 0; sI < RT; ++sI) {
            const int slot = sI * 256 + (int)threadIdx.x;
            const int pr = slot % (ROWS / 2), ml = slot / (ROWS / 2);
            const int r = n0 + 2 * pr, m = mp + ml;
            float a = (red[0][2 * pr][ml] + red[1][2 * pr][ml]) + (red[2][2 * pr][ml] + red[3][2 * pr][ml]);
            float b = (red[0][2 * pr + 1][ml] + red[1][2 * pr + 1][ml]) + (red[2][2 * pr + 1][ml] + red[3][2 * pr + 1][ml]);
            if (r >= N || m >= M) continue;
            if (FP8) { a *= wscale[r]; b *= wscale[min(r + 1, N - 1)]; }
            gemm_epilogue<EPI, ROWS>(a, b, r, m, ml, pr, ge, s_rms, (PRE_RES || PRE_QKV) && use_pre && mp == (int)blockIdx.z * PF_M ? &pre : nullptr);
        }
    }
}

constexpr unsigned DOWN_SPIN_MAX = 1u << 17;  // polls of an in-launch exchange unit before a thread gives up (~0.1 s; reported through epoch[1])
// ---- down projection of a decode step (M <= 32 rows) that closes the layer itself: split-K over gridDim.y blocks per 16-row weight tile
// as before (every CU streams a 32 KB weight tile: the K depth of 4096 needs all of them), but the K partials meet INSIDE the launch
// instead of in slabs + a k_prep node: thread (row pair pr, activation row ml) of block y publishes its two partial sums as one 16-byte
// unit {a, tag, b, tag} (relaxed agent-scope write-through store, each 8-byte half self-validating -- the edge protocol of the persistent
// decode kernels, lm_persist_dev.h) unless block y owns row ml; wave y of block y owns rows [y * 32 / ksplit, (y + 1) * 32 / ksplit):
// it sweeps the other blocks' units of its slots (sc1 loads, retried until both tags match), adds the partials in block order and runs the
// epilogue -- x += sum (residual stream), split(x * g_next) in fragment-major order for the next GEMM and one sum-of-squares partial per
// (tile, row); the next GEMM divides its accumulators by the row's rms (EPI_QKV_RMS / EPI_STORE_RMS).  One of six graph nodes per layer
// (~5 us each at 32 rows whatever they do) becomes a ~1 us in-launch hand-off among the <= 4 blocks of a tile, which share an XCD
// (linear block id = tile + gridDim.x * y, gridDim.x % 8 == 0).  tag = step epoch * 4096 + node id: a unit is rewritten by every node
// that uses the buffer, so a stale unit always carries the tag of the previous node or step.
// Measured and rejected first (round 5): the same layer-closing epilogue on an UN-split block (16 weight rows x 16 activation rows x the
// whole depth, 128 blocks): 9.0 us per node against 5.0 + 5.1 for the split GEMM + k_prep -- 384 KB of operands through ONE CU's vector
// memory path at one wave per SIMD is ~55 GB/s per CU.
template <int NKS, bool FP8>
__global__ __launch_bounds__(256) void k_gemm_down(const bf16_t* __restrict__ Xf, int M, int K, const void* __restrict__ Wv,
                                                   const float* __restrict__ wscale, int N, float* __restrict__ Y, int ldy, NormAux na,
                                                   u32x4* __restrict__ xchg, uint32_t* __restrict__ epoch, uint32_t node_id, int nap) {
    __shared__ float red[4][16][33];
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, y = blockIdx.y, ksplit = gridDim.y, rows_per = PF_M / ksplit;
    const unsigned tag = epoch[0] * 4096u + node_id;
    const int pr = threadIdx.x & 7, ml = threadIdx.x >> 3;   // epilogue slot: rows (r, r + 1) of activation row ml
    const int r = n0 + 2 * pr;
    const bool owner = ml / rows_per == y;                  // == (kq == y) for ksplit == 4; wave-uniform for every ksplit
    float2 xo = make_float2(0.f, 0.f), gw = make_float2(1.f, 1.f), sc = make_float2(1.f, 1.f);
    const int kbeg = (y * 4 + kq) * NKS * 32 + (lane >> 4) * 8;
    u32x4 wf[NKS];
    {
        const size_t woff = (size_t)(n0 + (lane & 15)) * K + kbeg;
        if (FP8) {
            const uint8_t* wp = reinterpret_cast<const uint8_t*>(Wv) + woff;
            u32x2 raw[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) raw[ks] = ld_stream(reinterpret_cast<const u32x2*>(wp + ks * 32));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[ks] = fp8x8_to_bf16x8(raw[ks]);
        } else {
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wv) + woff;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[ks] = ld_stream(reinterpret_cast<const u32x4*>(wp + ks * 32));
        }
    }
    const bf16_t* xp = Xf + (size_t)((y * 4 + kq) * NKS) * 2048 + lane * 8;
    u32x4 xf[NKS][4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        xf[ks][0] = *reinterpret_cast<const u32x4*>(xp + ks * 2048);          // hi, rows 0..15
        xf[ks][1] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 1024);   // lo, rows 0..15
        xf[ks][2] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 512);    // hi, rows 16..31
        xf[ks][3] = *reinterpret_cast<const u32x4*>(xp + ks * 2048 + 1536);   // lo, rows 16..31
    }
    if (owner) {  // what the epilogue reads from memory does not depend on the GEMM: requested behind the operands
        if (ml < M) xo = *reinterpret_cast<const float2*>(Y + (size_t)ml * ldy + r);
        gw = *reinterpret_cast<const float2*>(na.g + r);
    }
    if (FP8) sc = *reinterpret_cast<const float2*>(wscale + r);
    FS_ISSUE_FENCE();
    f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, wf[ks]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, xf[ks][j]), acc[j >> 1], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {  // lane holds C[row = (lane>>4)*4 + i][m = mt*16 + (lane&15)]
        const f32x4v c = acc[mt];
        float* rp = &red[kq][(lane >> 4) * 4][mt * 16 + (lane & 15)];
        rp[0] = c.x; rp[33] = c.y; rp[66] = c.z; rp[99] = c.w;
    }
    __syncthreads();
    float a = (red[0][2 * pr][ml] + red[1][2 * pr][ml]) + (red[2][2 * pr][ml] + red[3][2 * pr][ml]);
    float b = (red[0][2 * pr + 1][ml] + red[1][2 * pr + 1][ml]) + (red[2][2 * pr + 1][ml] + red[3][2 * pr + 1][ml]);
    if (FP8) { a *= sc.x; b *= sc.y; }
    // unit of (tile, source block, row, pair)
    u32x4* const units = xchg + ((size_t)blockIdx.x * ksplit * PF_M + ml) * 8 + pr;
    if (!owner) {
        const u32x4 u = {__float_as_uint(a), tag, __float_as_uint(b), tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(units + (size_t)y * PF_M * 8), "v"(u) : "memory");
        return;
    }
    // the other blocks' units of this slot: all requests in flight together, retried (only the missing ones) until both tags of each match.
    // The partners finish their K quarter when this block does and their stores take ~1 us to become visible: polling from the first instant
    // only re-reads units that cannot be there yet (memory-side traffic in front of the stores everybody waits for), so the sweep starts
    // `nap` x 64 clocks later (as the persistent kernels' pre-sweep naps)
    for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(1);
    u32x4 pv[4];
    {
        bool ok[4];
#pragma unroll
        for (int ys = 0; ys < 4; ++ys) { ok[ys] = ys >= ksplit || ys == y; pv[ys] = u32x4{0u, 0u, 0u, 0u}; }
        for (unsigned spins = 0;; ++spins) {
#pragma unroll
            for (int ys = 0; ys < 4; ++ys)
                if (!ok[ys]) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[ys]) : "v"(units + (size_t)ys * PF_M * 8) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool all = true;
#pragma unroll
            for (int ys = 0; ys < 4; ++ys) {
                // (the asm outputs above are only written when the load was issued; keep the compiler from assuming otherwise)
                asm volatile("" : "+v"(pv[ys]));
                if (!ok[ys]) ok[ys] = pv[ys].y == tag && pv[ys].w == tag;
                all = all && ok[ys];
            }
            if (all) break;
            if (spins > DOWN_SPIN_MAX) { atomicAdd(epoch + 1, 1u); break; }  // (reported: check_rows_xchg)
        }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int ys = 0; ys < 4; ++ys)  // partials in block order (the own one from registers)
        if (ys < ksplit) {
            s0 += ys == y ? a : __uint_as_float(pv[ys].x);
            s1 += ys == y ? b : __uint_as_float(pv[ys].z);
        }
    const float v0 = xo.x + s0, v1 = xo.y + s1;
    const bool live = ml < M;
    const float ssq = group_sum<8>(live ? fmaf(v0, v0, v1 * v1) : 0.f);
    if (!live) return;
    *reinterpret_cast<float2*>(Y + (size_t)ml * ldy + r) = make_float2(v0, v1);
    bf16_t h0, l0, h1, l1;
    split_bf16(v0 * gw.x, h0, l0); split_bf16(v1 * gw.y, h1, l1);
    *reinterpret_cast<uint32_t*>(na.A2 + frag_off(ml, r, 0, na.D)) = h0 | ((uint32_t)h1 << 16);
    *reinterpret_cast<uint32_t*>(na.A2 + frag_off(ml, r, 1, na.D)) = l0 | ((uint32_t)l1 << 16);
    if (pr == 0) na.ss[(size_t)ml * na.nblk + blockIdx.x] = ssq;
}

// ---- large-M variant (M >= GB_MIN_M rows: prefill passes, group prefill, big static batches) ---------------------------------
// k_gemm3 re-reads a panel's activation fragments once per 16 (or 32) weight rows through the CU's vector L1, which bounds it at
// 64 B/clk; with many panels that traffic, not the weight stream, is the cost.  Here a block owns 64 weight rows x one 1024-deep
// K range (wave = (K half, 32-row group): 2 A tiles x 16 k-steps stay in VGPRs for every panel) and the panel's fragments go
// through LDS once per block: 4-k-step chunks (32 KB: both K halves), double-buffered, global loads for chunk i+1 in flight
// during the 32 MFMAs per wave of chunk i; every fragment read from LDS feeds two A tiles.  Per chunk and block: 32 KB through
// L1, 64 KB of LDS reads, 128 MFMAs -- the three pipes are balanced.  The two K halves meet in LDS (aliased onto the stage
// buffer that was just consumed) and the epilogue is the one of k_gemm3 (ROWS = 64).
constexpr int GB_ROWS = 64, GB_CH = 4, GB_MIN_M = 128;
static int gemm_big_min_m() {  // FISHRT_GEMM_BIG_MIN_M: tuning / test hook (rows from which the LDS-staged variant is taken)
    static const int v = [] { const char* e = std::getenv("FISHRT_GEMM_BIG_MIN_M"); return e ? std::atoi(e) : GB_MIN_M; }();
    return v;
}
// measured (tools/ubench_gemm, Fish-1.5 shapes): a block's fixed cost (128 KB weight tile + 4 latency-exposed chunk rounds) is
// ~8 us, so the variant wins from 128 rows for the wide GEMMs (W13: 128 tiles, W2: 16 tiles x 4 K ranges) and only from
// ~512 rows for Wqkv / Wo (20 / 16 tiles)
static bool gemm_big_ok(int M, int N, int K, int ksplit) {
    if (K % ksplit != 0 || K / ksplit != 1024) return false;
    const int m0 = gemm_big_min_m();
    return M >= 4 * m0 || (M >= m0 && (N + GB_ROWS - 1) / GB_ROWS * ksplit >= 64);
}
template <int EPI, bool FP8>
__global__ __launch_bounds__(256, 2) void k_gemm_big(const bf16_t* __restrict__ Xf, int M, int K,
                                                     const void* __restrict__ Wv, const float* __restrict__ wscale, int N, float* __restrict__ Y, int ldy,
                                                     size_t slab_stride, bf16_t* __restrict__ Of, int ldo,
                                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                     const SeqState* __restrict__ state, KVView kv, int H, int Hk, int Dh, RowMap rm, NormAux na) {
    constexpr int ROWS = GB_ROWS, NKS = 16, STAGE = 2 * GB_CH * 4096;  // bytes per stage: 2 K halves x 4 k-steps x 4 KB
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = wave >> 1, rg = wave & 1;
    const int n0 = blockIdx.x * ROWS;
    int mp = (int)blockIdx.z * PF_M;
    if (mp >= M) return;
    int pos0 = 0, rope_off = 0;
    if (EPI == EPI_QKV || EPI == EPI_QKV_RMS) { pos0 = state->pos; rope_off = state->rope_off; }
    const GemmEpi ge{Y, ldy, slab_stride, Of, ldo, cos_t, sin_t, kv, H, Hk, Dh, rm, na, pos0, rope_off, N, state};
    const int ksw = (int)blockIdx.y * 32 + kh * NKS;  // this wave's first 32-deep k-step
    u32x4 wf[2][NKS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const size_t woff = (size_t)min(n0 + rg * 32 + t * 16 + (lane & 15), N - 1) * K + (size_t)ksw * 32 + (lane >> 4) * 8;
        if (FP8) {
            const uint8_t* wp = reinterpret_cast<const uint8_t*>(Wv) + woff;
            u32x2 raw[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) raw[ks] = ld_stream(reinterpret_cast<const u32x2*>(wp + ks * 32));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[t][ks] = fp8x8_to_bf16x8(raw[ks]);
        } else {
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wv) + woff;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[t][ks] = ld_stream(reinterpret_cast<const u32x4*>(wp + ks * 32));
        }
    }
    // chunk c of panel p: thread unit i -> K half i >> 2, k-step i & 3 of the chunk, 16-B slot tid of that k-step's 4 KB block
    u32x4 g[8];
    auto load_chunk = [&](int p, int c) {
        const bf16_t* base = Xf + ((size_t)(p >> 5) * (K >> 5) + (size_t)blockIdx.y * 32 + c * GB_CH) * 2048 + tid * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const u32x4*>(base + (size_t)((i >> 2) * NKS + (i & 3)) * 2048);
    };
    auto store_chunk = [&](int sb) {
        uint8_t* dst = lds + sb * STAGE + tid * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dst + i * 4096) = g[i];
    };
    load_chunk(mp, 0);
    store_chunk(0);
    __syncthreads();
    int sb = 0;
    const int mstep = (int)gridDim.z * PF_M;
    for (; mp < M; mp += mstep) {
        const int mp_next = mp + mstep;
        float ss8 = 0.f;
        if (epi_rms(EPI)) {
            const float* sp = na.ss + (size_t)(mp + (tid >> 3)) * na.nblk + (tid & 7);
            for (int q8 = 0; q8 < na.nblk; q8 += 8) ss8 += sp[q8];
        }
        f32x4v acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < NKS / GB_CH; ++c) {
            const bool has_next = c + 1 < NKS / GB_CH || mp_next < M;
            if (has_next) load_chunk(c + 1 < NKS / GB_CH ? mp : mp_next, (c + 1) % (NKS / GB_CH));
            const uint8_t* src = lds + sb * STAGE + kh * (GB_CH * 4096) + lane * 16;
#pragma unroll
            for (int ks = 0; ks < GB_CH; ++ks) {
                bf16x8 xf[4];  // hi rows 0..15, lo rows 0..15, hi rows 16..31, lo rows 16..31 (order of k_gemm3)
                xf[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ks * 4096));
                xf[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ks * 4096 + 2048));
                xf[2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ks * 4096 + 1024));
                xf[3] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + ks * 4096 + 3072));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, wf[t][c * GB_CH + ks]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[j], acc[t][j >> 1], 0, 0, 0);
                }
            }
            if (c == NKS / GB_CH - 1) {  // panel complete: the K halves meet in the stage buffer that was just consumed
                float (*red)[ROWS][33] = reinterpret_cast<float (*)[ROWS][33]>(lds + sb * STAGE);
                float* s_rms = reinterpret_cast<float*>(lds + sb * STAGE + 2 * ROWS * 33 * sizeof(float));
                __syncthreads();
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const f32x4v cc = acc[t][mt];
                        float* rp = &red[kh][rg * 32 + t * 16 + (lane >> 4) * 4][mt * 16 + (lane & 15)];
                        rp[0] = cc.x; rp[33] = cc.y; rp[66] = cc.z; rp[99] = cc.w;
                    }
                if (epi_rms(EPI)) {
                    const float tot = group_sum<8>(ss8);
                    if ((tid & 7) == 0) s_rms[tid >> 3] = sqrtf(tot / (float)na.D + na.eps);
                }
                __syncthreads();
#pragma unroll
                for (int sI = 0; sI < ROWS / 16; ++sI) {
                    const int slot = sI * 256 + tid;
                    const int pr = slot % (ROWS / 2), ml = slot / (ROWS / 2);
                    const int r = n0 + 2 * pr, m = mp + ml;
                    float a = red[0][2 * pr][ml] + red[1][2 * pr][ml];
                    float b = red[0][2 * pr + 1][ml] + red[1][2 * pr + 1][ml];
                    if (r >= N || m >= M) continue;
                    if (FP8) { a *= wscale[r]; b *= wscale[min(r + 1, N - 1)]; }
                    gemm_epilogue<EPI, ROWS>(a, b, r, m, ml, pr, ge, s_rms);
                }
            }
            if (has_next) store_chunk(sb ^ 1);
            __syncthreads();
            sb ^= 1;
        }
    }
}

// combine the per-chunk attention partials of M rows -> attn hi/lo bf16 [PF_M][H*DH] (input of the Wo GEMM)
template <int DH>
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ part_all, int n_chunks_max, int chunk,
                                                      const SeqState* __restrict__ state, int pos_step, bf16_t* __restrict__ Ohi, int H) {
    const int m = blockIdx.x;
    const int T = row_pos(state, m, pos_step) + 1, nc = (T + chunk - 1) / chunk;
    const float* part = part_all + (size_t)m * H * n_chunks_max * (DH + 2);
    __shared__ float wl[32 * 128];
    for (int h = threadIdx.x; h < H; h += 256) {
        const float* p = part + (size_t)h * n_chunks_max * (DH + 2);
        float mn = -1e30f;
        for (int c = 0; c < nc; ++c) mn = fmaxf(mn, p[c * (DH + 2) + DH]);
        float L = 0.f;
        for (int c = 0; c < nc; ++c) L += p[c * (DH + 2) + DH + 1] * __expf(p[c * (DH + 2) + DH] - mn);
        const float inv = 1.f / L;
        for (int c = 0; c < nc; ++c) wl[h * 128 + c] = __expf(p[c * (DH + 2) + DH] - mn) * inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < H * DH; e += 256) {
        const int h = e / DH, dd = e % DH;
        const float* p = part + (size_t)h * n_chunks_max * (DH + 2) + dd;
        float O = 0.f;
        for (int c = 0; c < nc; ++c) O = fmaf(wl[h * 128 + c], p[c * (DH + 2)], O);
        bf16_t hi, lo;
        split_bf16(O, hi, lo);
        Ohi[frag_off(m, e, 0, H * DH)] = hi;
        Ohi[frag_off(m, e, 1, H * DH)] = lo;
    }
}

// Whole attention of one activation row over <= 8 cached tokens (the fast decoder in the batched row path): all H heads in one
// block, scores by two threads per (head, token), softmax . V with four output dims per thread, result straight into the
// fragment-major hi/lo GEMM input -- replaces k_attn_decode + k_attn_combine (two graph nodes) where one 8-token page is all
// there is.  The row's page is page_table[m * pt_stride] (single page), its length state->pos + 1 + m * pos_step <= 8.
template <int DH>
__global__ __launch_bounds__(256) void k_attn_small_rows(const float* __restrict__ q_all, KVView kv, const SeqState* __restrict__ state,
                                                         int H, int Hk, int pos_step, int pt_stride, bf16_t* __restrict__ Ohi, int identity_pages) {
    __shared__ float sc[32 * 8];
    const int m = blockIdx.x, tid = threadIdx.x;
    // identity_pages: row m's only page IS page m (the batched fast decoder's table) -- one dependent L2 round trip less in a node that is
    // nothing but a chain of them
    const int page = identity_pages ? m : kv.page_table[(size_t)m * pt_stride];
    const int n_rep = H / Hk;
    const float* q = q_all + (size_t)m * H * DH;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(kv.k) + (size_t)page * Hk * KV_PAGE * DH;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(kv.v) + (size_t)page * Hk * KV_PAGE * DH;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int QD = DH / 2;
    // the V values of this thread's first (head, 4 dims) output item are requested with the K rows, not behind the score barrier (all 8
    // token slots of the page exist; slots >= T hold stale finite bf16 and are masked by t < T below)
    uint2 vpre[8];
    {
        const int e4 = tid * 4;
        if (e4 < H * DH) {
            const int h = e4 / DH, dd = e4 % DH;
#pragma unroll
            for (int t = 0; t < 8; ++t) vpre[t] = *reinterpret_cast<const uint2*>(vb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + dd);
        }
    }
    const int T = row_pos(state, m, pos_step) + 1;
    for (int e1 = tid >> 1; e1 < H * 8; e1 += 128) {
        const int h = e1 >> 3, t = e1 & 7, sl = tid & 1;
        const bf16_t* kp = kb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + sl * QD;
        const float* qp = q + h * DH + sl * QD;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < QD; i += 8) {
            float kf[8];
            WTr<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(kp + i), kf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(qp[i + j], kf[j] * scale, acc);  // q . (k^T * scale)  (dual_ar.rs:260)
        }
        acc += dpp_mov<DPP_XOR1>(acc);
        if (sl == 0) sc[e1] = acc;
    }
    __syncthreads();
    for (int e4 = tid * 4; e4 < H * DH; e4 += 1024) {
        const int h = e4 / DH, dd = e4 % DH;
        const bool firstit = e4 == tid * 4;
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < T) mx = fmaxf(mx, sc[h * 8 + t]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < T) {
                const float p = __expf(sc[h * 8 + t] - mx);
                L += p;
                const uint2 vv = firstit ? vpre[t] : *reinterpret_cast<const uint2*>(vb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + dd);
                O[0] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.x & 0xFFFFu)), O[0]); O[1] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.x >> 16)), O[1]);
                O[2] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.y & 0xFFFFu)), O[2]); O[3] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.y >> 16)), O[3]);
            }
        const float inv = 1.f / L;
        bf16_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_bf16(O[i] * inv, hi[i], lo[i]);
        uint2 ph, pl;
        ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
        pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e4, 0, H * DH)) = ph;
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e4, 1, H * DH)) = pl;
    }
}

// k_attn_small_rows for the FIRST fast layer of the codebook passes 1.. (round 6): that layer's input row is fast_embeddings[code] of the code the
// previous pass's sampler picked (static_batch.rs:236-241 / single_batch.rs:181-183), so attention_norm + Wqkv of it is row `code` of the qkv
// table the persistent fast decoder builds at load time (lm_persist.hip k_pf_qkv0_table: 1024 x 1280 f32, pre-RoPE).  The node takes q / k / v of
// the new token from that row instead of from a Wqkv GEMM node in front of it (7 nodes of a step disappear): RoPE at the pass's position, K / V
// appended to the row's page exactly as the GEMM epilogue would (bf16, EPI_QKV), and the new position's K / V used from LDS in their cached
// (bf16-rounded) form.  code = row_states[m].cur[code_slot] (written by the sampler node in front of this one).
template <int DH>
__global__ __launch_bounds__(256) void k_attn_small_rows_tbl(const float* __restrict__ tbl, const SeqState* __restrict__ row_states, int code_slot, KVView kv,
                                                             const SeqState* __restrict__ state, int H, int Hk, int pos_step, int pt_stride,
                                                             const float* __restrict__ cos_t, const float* __restrict__ sin_t, bf16_t* __restrict__ Ohi,
                                                             int identity_pages) {
    __shared__ float sc[32 * 8];
    __shared__ __attribute__((aligned(16))) float qS[32 * DH];
    __shared__ __attribute__((aligned(16))) float knS[4 * DH], vnS[4 * DH];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int page = identity_pages ? m : kv.page_table[(size_t)m * pt_stride];
    const int n_rep = H / Hk;
    bf16_t* kb = reinterpret_cast<bf16_t*>(kv.k) + (size_t)page * Hk * KV_PAGE * DH;
    bf16_t* vb = reinterpret_cast<bf16_t*>(kv.v) + (size_t)page * Hk * KV_PAGE * DH;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int QD = DH / 2;
    uint2 vpre[8];  // (as k_attn_small_rows: requested up front; the slot of the new position is stale here and replaced below)
    {
        const int e4 = tid * 4;
        if (e4 < H * DH) {
            const int h = e4 / DH, dd = e4 % DH;
#pragma unroll
            for (int t = 0; t < 8; ++t) vpre[t] = *reinterpret_cast<const uint2*>(vb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + dd);
        }
    }
    const int pos = row_pos(state, m, pos_step), T = pos + 1;
    const int rpos = pos + (pos_step < 0 ? state[m].rope_off : state->rope_off);
    const uint32_t code = row_states[m].cur[code_slot];
    const int qdim = H * DH, kdim = Hk * DH;
    const float* row = tbl + (size_t)code * (qdim + 2 * kdim);
    for (int i = tid; i < (qdim + 2 * kdim) / 2; i += 256) {
        const int r = 2 * i;
        const float2 ab = *reinterpret_cast<const float2*>(row + r);
        if (r < qdim + kdim) {  // rope_i (dual_ar.rs:246-247)
            const int j = (r % DH) / 2;
            const float cs = cos_t[(size_t)rpos * QD + j], sn = sin_t[(size_t)rpos * QD + j];
            const float o0 = ab.x * cs - ab.y * sn, o1 = ab.x * sn + ab.y * cs;
            if (r < qdim) { qS[r] = o0; qS[r + 1] = o1; }
            else {
                const int rk = r - qdim;
                const bf16_t b0 = WTr<bf16_t>::from_f32(o0), b1 = WTr<bf16_t>::from_f32(o1);
                *reinterpret_cast<uint32_t*>(kb + ((size_t)(rk / DH) * KV_PAGE + pos) * DH + rk % DH) = b0 | ((uint32_t)b1 << 16);
                knS[rk] = bf16_bits_to_f32(b0); knS[rk + 1] = bf16_bits_to_f32(b1);
            }
        } else {
            const int rv = r - qdim - kdim;
            const bf16_t b0 = WTr<bf16_t>::from_f32(ab.x), b1 = WTr<bf16_t>::from_f32(ab.y);
            *reinterpret_cast<uint32_t*>(vb + ((size_t)(rv / DH) * KV_PAGE + pos) * DH + rv % DH) = b0 | ((uint32_t)b1 << 16);
            vnS[rv] = bf16_bits_to_f32(b0); vnS[rv + 1] = bf16_bits_to_f32(b1);
        }
    }
    __syncthreads();
    for (int e1 = tid >> 1; e1 < H * 8; e1 += 128) {
        const int h = e1 >> 3, t = e1 & 7, sl = tid & 1;
        const bf16_t* kp = kb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + sl * QD;
        const float* kn = knS + (h / n_rep) * DH + sl * QD;
        const float* qp = qS + h * DH + sl * QD;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < QD; i += 8) {
            float kf[8];
            WTr<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(kp + i), kf);
            if (t == pos) {
#pragma unroll
                for (int j = 0; j < 8; ++j) kf[j] = kn[i + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(qp[i + j], kf[j] * scale, acc);  // q . (k^T * scale)  (dual_ar.rs:260)
        }
        acc += dpp_mov<DPP_XOR1>(acc);
        if (sl == 0) sc[e1] = acc;
    }
    __syncthreads();
    for (int e4 = tid * 4; e4 < H * DH; e4 += 1024) {
        const int h = e4 / DH, dd = e4 % DH;
        const bool firstit = e4 == tid * 4;
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < T) mx = fmaxf(mx, sc[h * 8 + t]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < T) {
                const float p = __expf(sc[h * 8 + t] - mx);
                L += p;
                if (t == pos) {
                    const float* vn = vnS + (h / n_rep) * DH + dd;
                    O[0] = fmaf(p, vn[0], O[0]); O[1] = fmaf(p, vn[1], O[1]); O[2] = fmaf(p, vn[2], O[2]); O[3] = fmaf(p, vn[3], O[3]);
                } else {
                    const uint2 vv = firstit ? vpre[t] : *reinterpret_cast<const uint2*>(vb + ((size_t)(h / n_rep) * KV_PAGE + t) * DH + dd);
                    O[0] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.x & 0xFFFFu)), O[0]); O[1] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.x >> 16)), O[1]);
                    O[2] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.y & 0xFFFFu)), O[2]); O[3] = fmaf(p, bf16_bits_to_f32((bf16_t)(vv.y >> 16)), O[3]);
                }
            }
        const float inv = 1.f / L;
        bf16_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_bf16(O[i] * inv, hi[i], lo[i]);
        uint2 ph, pl;
        ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
        pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e4, 0, H * DH)) = ph;
        *reinterpret_cast<uint2*>(Ohi + frag_off(m, e4, 1, H * DH)) = pl;
    }
}

// Static-batch decode attention in ONE node: block = (kv head g, activation row m) walks the row's whole KV prefix chunk by
// chunk (same wave / lane-group geometry as k_attn_decode; the next chunk's K/V tiles are in flight while the current one is
// scored), every lane group keeps a running (max, sum, o) (online softmax), the waves meet once in LDS and the normalised result
// goes straight into the fragment-major hi/lo input of the Wo GEMM -- replaces k_attn_decode over (chunks x rows) blocks +
// k_attn_combine (two graph nodes and the partials round trip) for the rows-are-sequences passes.
// PART (batch-1 decode over a LONG prefix, > 8 chunks of 128 tokens): blockIdx.z = super-chunk of `tpb` consecutive chunks; instead
// of the normalised hi/lo row the block leaves {o, m, l} in slot blockIdx.z of k_attn_decode's partials layout, so that k_wo always
// merges <= 8 partials in registers (its general LDS merge over 32..64 chunks costs 9..16 us per layer at 4..8 k tokens).
template <typename WT, int DH, int NREP, bool PART = false>
__global__ __launch_bounds__((AttnGeom<WT, DH>::NW * 64)) void k_attn_rows(const float* __restrict__ q_all, KVView kv,
                                                   const SeqState* __restrict__ state, int Hk, int pos_step, int pt_stride,
                                                   bf16_t* __restrict__ Ohi, int hsplit, float* __restrict__ part_all = nullptr,
                                                   int n_chunks_max = 0, int tpb = 1 << 30) {
    // hsplit > 1: the query heads of a kv group are spread over hsplit blocks of NREP heads each (small batches: more blocks,
    // less VALU work per wave; the K/V tiles are then read hsplit times, from L2)
    int g = blockIdx.x / hsplit, hb = (blockIdx.x % hsplit) * NREP, mrow = blockIdx.y;
    if (!PART && hsplit > 1) {
        // XCD-aware placement: workgroups go round-robin over the 8 XCDs by linear id, and each XCD has its own L2 -- the hsplit blocks
        // that share one (kv head, row) K/V stream must land on ONE XCD or the stream crosses the fabric hsplit times (PMC at B = 32:
        // 711 MB per step for 162 MB of K/V with the plain mapping)
        const int total = gridDim.x * gridDim.y, per_xcd = total / (8 * hsplit);
        if (per_xcd * 8 * hsplit == total) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.y, k = lin >> 3, grp = (lin & 7) * per_xcd + k / hsplit;
            g = grp % Hk; mrow = grp / Hk; hb = (k % hsplit) * NREP;
        }
    }
    const int GH = NREP * hsplit;  // query heads per kv head
    kv.page_table += (size_t)mrow * pt_stride;
    const float* q = q_all + (size_t)mrow * Hk * GH * DH;
    constexpr int EPL = WTr<WT>::EPL;
    constexpr int LPT = DH / EPL, G = 64 / LPT, NRP = NREP < G ? NREP : G, NTS = G / NRP, NHP = NREP / NRP;
    constexpr int TW = AttnGeom<WT, DH>::TW, NW = AttnGeom<WT, DH>::NW, CH = NW * TW, TPG = TW / NTS, NLD = TW * LPT / 64;
    static_assert(TW % NTS == 0 && NLD >= 1 && KV_PAGE % TW == 0, "attention geometry");
    using vec = typename WTr<WT>::vec;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) WT sk[NW][TW * DH];
    __shared__ __attribute__((aligned(16))) WT sv[NW][TW * DH];
    __shared__ float sp[NW][NREP][NTS][DH + 2];
    const int T = row_pos(state, mrow, pos_step) + 1;  // the row's own K/V were appended by the qkv stage
    const int nc_all = (T + CH - 1) / CH;
    const int c0 = PART ? (int)blockIdx.z * tpb : 0, nc = PART ? min(nc_all, c0 + tpb) : nc_all;  // this block's chunks [c0, nc)
    if (c0 >= nc) return;  // super-chunk past the current length (the graph bucket launches a power of two of them)
    vec kreg[NLD], vreg[NLD];
    auto load_tiles = [&](int c) {  // chunk c: this wave's TW tokens live in one page
        const int t_base = __builtin_amdgcn_readfirstlane(c * CH + wave * TW);
        const int page = kv.page_table[t_base / KV_PAGE];
        const WT* kpage = reinterpret_cast<const WT*>(kv.k) + (size_t)(page * Hk + g) * KV_PAGE * DH;
        const WT* vpage = reinterpret_cast<const WT*>(kv.v) + (size_t)(page * Hk + g) * KV_PAGE * DH;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int tl = (i * 64 + lane) / LPT, sl = lane % LPT;
            const int t = t_base + tl;
            kreg[i] = *reinterpret_cast<const vec*>(kpage + (size_t)(t % KV_PAGE) * DH + sl * EPL);
            vreg[i] = *reinterpret_cast<const vec*>(vpage + (size_t)(t % KV_PAGE) * DH + sl * EPL);
        }
    };
    load_tiles(c0);
    const int gi = lane / LPT, sub = lane % LPT;
    const int rl = gi % NRP, ts = gi / NRP;
    constexpr bool POW2 = (DH == 64 || DH == 16 || DH == 256);
    const float scale = 1.0f / sqrtf((float)DH);
    float qr[NHP][EPL];
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {
        const float* qp = q + (size_t)(g * GH + hb + hp * NRP + rl) * DH + sub * EPL;
#pragma unroll
        for (int i = 0; i < EPL; ++i) qr[hp][i] = POW2 ? qp[i] * scale : qp[i];  // 2^-k scale folded into q (exact), see k_attn_decode
    }
    float mr[NHP], lr[NHP], orun[NHP][EPL];
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {
        mr[hp] = -1e30f; lr[hp] = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) orun[hp][i] = 0.f;
    }
    for (int c = c0; c < nc; ++c) {
        const int t_base = c * CH + wave * TW;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *reinterpret_cast<vec*>(&sk[wave][(size_t)(i * 64 + lane) * EPL]) = kreg[i];
            *reinterpret_cast<vec*>(&sv[wave][(size_t)(i * 64 + lane) * EPL]) = vreg[i];
        }
        if (c + 1 < nc) load_tiles(c + 1);
        // (each wave reads only the tile it wrote: no block barrier, LDS operations of one wave stay in order)
#pragma unroll
        for (int hp = 0; hp < NHP; ++hp) {
            float sc[TPG];
            float mc = -1e30f;
#pragma unroll
            for (int j = 0; j < TPG; ++j) {
                const int tl = ts + j * NTS;
                float kf[EPL];
                WTr<WT>::unpack(*reinterpret_cast<const vec*>(&sk[wave][(size_t)tl * DH + sub * EPL]), kf);
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) a = POW2 ? fmaf(qr[hp][i], kf[i], a) : fmaf(qr[hp][i], kf[i] * scale, a);
                a = group_sum<LPT>(a);
                sc[j] = (t_base + tl < T) ? a : -1e30f;
                mc = fmaxf(mc, sc[j]);
            }
            const float mn = fmaxf(mr[hp], mc), f = __expf(mr[hp] - mn);
            float l = lr[hp] * f, o[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) o[i] = orun[hp][i] * f;
#pragma unroll
            for (int j = 0; j < TPG; ++j) {
                const int tl = ts + j * NTS;
                const float p = (t_base + tl < T) ? __expf(sc[j] - mn) : 0.f;
                l += p;
                float vf[EPL];
                WTr<WT>::unpack(*reinterpret_cast<const vec*>(&sv[wave][(size_t)tl * DH + sub * EPL]), vf);
#pragma unroll
                for (int i = 0; i < EPL; ++i) o[i] = fmaf(p, vf[i], o[i]);
            }
            mr[hp] = mn; lr[hp] = l;
#pragma unroll
            for (int i = 0; i < EPL; ++i) orun[hp][i] = o[i];
        }
    }
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {
        float* dst = sp[wave][hp * NRP + rl][ts];
#pragma unroll
        for (int i = 0; i < EPL; ++i) dst[sub * EPL + i] = orun[hp][i];
        if (sub == 0) { dst[DH] = mr[hp]; dst[DH + 1] = lr[hp]; }
    }
    __syncthreads();
    const int H = Hk * GH;
    for (int e = threadIdx.x; e < NREP * DH; e += NW * 64) {
        const int r = e / DH, dd = e % DH;
        float mn = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int k = 0; k < NTS; ++k) mn = fmaxf(mn, sp[w][r][k][DH]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int k = 0; k < NTS; ++k) {
                const float cf = __expf(sp[w][r][k][DH] - mn);
                L += sp[w][r][k][DH + 1] * cf;
                O += sp[w][r][k][dd] * cf;
            }
        if (PART) {
            float* dst = part_all + (((size_t)mrow * H + g * GH + hb + r) * n_chunks_max + blockIdx.z) * (DH + 2);
            dst[dd] = O;
            if (dd == 0) { dst[DH] = mn; dst[DH + 1] = L; }
        } else {
            bf16_t hi, lo;
            split_bf16(O / L, hi, lo);
            const int col = (g * GH + hb + r) * DH + dd;
            Ohi[frag_off(mrow, col, 0, H * DH)] = hi;
            Ohi[frag_off(mrow, col, 1, H * DH)] = lo;
        }
    }
}

// ------------------------------------------------------------------------------------------------ prefill attention (MFMA)
// Causal flash attention for the rows of a prefill pass (consecutive tokens of ONE sequence, bf16 KV, head_dim 64): block =
// (query head, 16-row tile); its 4 waves split the sequence page-wise (wave w takes KV pages w, w+4, ...), each producing a
// local (max, sum, O) with a two-pass softmax over its own pages; one LDS merge, and the normalised result goes straight into
// the fragment-major hi/lo GEMM input (replaces k_attn_decode over (chunks x rows) blocks + k_attn_combine: 57 + 7 us per
// layer at 384 rows).  Per page (64 tokens = 4 MFMA token tiles) every operand load is in flight before the first MFMA.
//   S^T[token][row] = K_tile[16 x 64] . Q^T[64 x 16]   v_mfma_f32_16x16x32_bf16, A = K straight from the paged cache (one 16-B load
//                                                      per lane per 32 dims), B = the rows' q split bf16 hi + lo (held in VGPRs)
//   two-pass softmax: pass 1 only takes the column maxima; pass 2 recomputes S, p = exp(s - max) -- the S^T accumulator layout
//   (lane: row l&15, tokens (l>>4)*4..+3) IS the A-operand layout of the 16x16x16 MFMA, so
//   O[row][dim] += P[row][16 tokens] . V[16 tokens][dim]   v_mfma_f32_16x16x16_bf16, p split hi + lo, B = V gathered from the
//                                                      wave's private LDS copy of the tile (token-major -> 4 strided bf16)
// The softmax scale 2^-3 is folded into q (exact).  Rows >= M and tokens past a row's position are masked.
typedef short short4v __attribute__((ext_vector_type(4)));
// blockIdx.y = sequence of a group pass (rows [y * M, (y + 1) * M), page table y * pt_stride further on); one sequence: gridDim.y = 1.
__global__ __launch_bounds__(256) void k_attn_prefill_mfma(const float* __restrict__ q_all, KVView kv, const SeqState* __restrict__ state,
                                                           int M, int H, int Hk, bf16_t* __restrict__ Ohi, int pt_stride) {
    constexpr int DH = 64, VLD = DH + 8;
    __shared__ __attribute__((aligned(16))) bf16_t vt[4][KV_PAGE * VLD];
    __shared__ __attribute__((aligned(16))) float sm_o[4][16][DH + 4];
    __shared__ float sm_m[4][16], sm_l[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x % H, rt = blockIdx.x / H;  // query head, row tile
    const int g = h / (H / Hk);
    const int row0 = rt * 16, pos0 = state->pos;        // row m of this sequence sits at position pos0 + m
    const int mbase = (int)blockIdx.y * M;              // first activation row of this sequence
    const int* ptab = kv.page_table + (size_t)blockIdx.y * pt_stride;
    const int c16 = lane & 15, q4 = lane >> 4;
    // B operand of QK^T: q[row0 + c16][h][ks*32 + q4*8 ..+8], scaled, split hi/lo
    bf16x8 qh[2], ql[2];
    {
        const int m = min(row0 + c16, M - 1);
        const float* qp = q_all + ((size_t)(mbase + m) * H + h) * DH + q4 * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(qp + ks * 32), b = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16_t h0, l0, h1, l1;
                split_bf16(v[2 * i] * 0.125f, h0, l0); split_bf16(v[2 * i + 1] * 0.125f, h1, l1);
                hw[i] = h0 | ((uint32_t)h1 << 16); lw[i] = l0 | ((uint32_t)l1 << 16);
            }
            u32x4 hv, lv; hv.x = hw[0]; hv.y = hw[1]; hv.z = hw[2]; hv.w = hw[3]; lv.x = lw[0]; lv.y = lw[1]; lv.z = lw[2]; lv.w = lw[3];
            qh[ks] = __builtin_bit_cast(bf16x8, hv); ql[ks] = __builtin_bit_cast(bf16x8, lv);
        }
    }
    const int my_pos = pos0 + row0 + c16;                       // last token this lane's row may see
    const int n_tok = pos0 + min(row0 + 15, M - 1) + 1;           // tokens the tile's LAST row needs
    const int n_groups = (n_tok + KV_PAGE - 1) / KV_PAGE;         // one group = one KV page = 4 token tiles of 16
    const bf16_t* kpool = reinterpret_cast<const bf16_t*>(kv.k);
    const bf16_t* vpool = reinterpret_cast<const bf16_t*>(kv.v);
    // S^T of the 4 token tiles of page `grp`: all 8 K loads (A operands, straight from the cache) are in flight together
    auto scores = [&](int grp, f32x4v (&sacc)[4]) {
        const int page = ptab[grp];
        const bf16_t* kp = kpool + ((size_t)(page * Hk + g) * KV_PAGE + c16) * DH + q4 * 8;
        u32x4 k0[4], k1[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            k0[tt] = *reinterpret_cast<const u32x4*>(kp + (size_t)tt * 16 * DH);
            k1[tt] = *reinterpret_cast<const u32x4*>(kp + (size_t)tt * 16 * DH + 32);
        }
        FS_ISSUE_FENCE();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4v a = f32x4v{0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k0[tt]), qh[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k0[tt]), ql[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k1[tt]), qh[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k1[tt]), ql[1], a, 0, 0, 0);
            const int t0 = grp * KV_PAGE + tt * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (t0 + q4 * 4 + i > my_pos) a[i] = -1e30f;  // causal mask (also hides stale rows of the page)
            sacc[tt] = a;
        }
    };
    // pass 1: column maxima
    float mx = -1e30f;
    for (int grp = wave; grp < n_groups; grp += 4) {
        f32x4v s4[4];
        scores(grp, s4);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) mx = fmaxf(fmaxf(mx, fmaxf(s4[tt][0], s4[tt][1])), fmaxf(s4[tt][2], s4[tt][3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // pass 2: P . V
    f32x4v o[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    bf16_t* myv = vt[wave];
    for (int grp = wave; grp < n_groups; grp += 4) {
        {   // stage the page's V (64 tokens x 64 dims) token-major into the wave's private LDS region: 8 x 16-B loads per lane
            const int page = ptab[grp];
            const bf16_t* vp = vpool + (size_t)(page * Hk + g) * KV_PAGE * DH;
            u32x4 vr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) vr[j] = *reinterpret_cast<const u32x4*>(vp + (size_t)(j * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = j * 64 + lane, tk = e >> 3, d8 = (e & 7) * 8;
                *reinterpret_cast<u32x4*>(myv + tk * VLD + d8) = vr[j];
            }
        }
        f32x4v s4[4];
        scores(grp, s4);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            uint32_t ph[2], pl[2];
            {
                bf16_t hb[4], lb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = s4[tt][i] > -1e29f ? __expf(s4[tt][i] - mx) : 0.f;
                    lsum += p;
                    split_bf16(p, hb[i], lb[i]);
                }
                ph[0] = hb[0] | ((uint32_t)hb[1] << 16); ph[1] = hb[2] | ((uint32_t)hb[3] << 16);
                pl[0] = lb[0] | ((uint32_t)lb[1] << 16); pl[1] = lb[2] | ((uint32_t)lb[3] << 16);
            }
            uint2 phv, plv; phv.x = ph[0]; phv.y = ph[1]; plv.x = pl[0]; plv.y = pl[1];
            const short4v pa_h = __builtin_bit_cast(short4v, phv), pa_l = __builtin_bit_cast(short4v, plv);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16_t* vq = myv + (tt * 16 + q4 * 4) * VLD + nt * 16 + c16;  // V[token tt*16 + q4*4 + j][dim nt*16 + c16]
                uint2 bv;
                bv.x = vq[0] | ((uint32_t)vq[VLD] << 16);
                bv.y = vq[2 * VLD] | ((uint32_t)vq[3 * VLD] << 16);
                const short4v vb = __builtin_bit_cast(short4v, bv);
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa_h, vb, o[nt], 0, 0, 0);
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa_l, vb, o[nt], 0, 0, 0);
            }
        }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    // merge the 4 page-splits: o[nt][i] = O_w[row q4*4 + i][dim nt*16 + c16]; max / sum live in column layout (row = c16)
    if (q4 == 0) { sm_m[wave][c16] = mx; sm_l[wave][c16] = lsum; }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) sm_o[wave][q4 * 4 + i][nt * 16 + c16] = o[nt][i];
    __syncthreads();
    {
        const int r = threadIdx.x >> 4, d4 = (threadIdx.x & 15) * 4, m = row0 + r;
        if (m < M) {
            const float mg = fmaxf(fmaxf(sm_m[0][r], sm_m[1][r]), fmaxf(sm_m[2][r], sm_m[3][r]));
            float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const float cf = __expf(sm_m[w2][r] - mg);  // a split without pages has max -1e30, sum 0, O 0
                L = fmaf(sm_l[w2][r], cf, L);
                const float4 ov = *reinterpret_cast<const float4*>(&sm_o[w2][r][d4]);
                O[0] = fmaf(ov.x, cf, O[0]); O[1] = fmaf(ov.y, cf, O[1]); O[2] = fmaf(ov.z, cf, O[2]); O[3] = fmaf(ov.w, cf, O[3]);
            }
            const float inv = 1.f / L;
            bf16_t hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split_bf16(O[i] * inv, hi[i], lo[i]);
            uint2 phv, plv;
            phv.x = hi[0] | ((uint32_t)hi[1] << 16); phv.y = hi[2] | ((uint32_t)hi[3] << 16);
            plv.x = lo[0] | ((uint32_t)lo[1] << 16); plv.y = lo[2] | ((uint32_t)lo[3] << 16);
            const int e = h * DH + d4;
            *reinterpret_cast<uint2*>(Ohi + frag_off(mbase + m, e, 0, H * DH)) = phv;
            *reinterpret_cast<uint2*>(Ohi + frag_off(mbase + m, e, 1, H * DH)) = plv;
        }
    }
}

template <typename WT>
__global__ void k_embed_rows(const WT* __restrict__ tok_emb, const WT* __restrict__ cb_emb, int dim, int n_cb, int cb_size,
                             const SampleCfg* __restrict__ cfg, const uint32_t* __restrict__ prompt,
                             const SeqState* __restrict__ state, float* __restrict__ X, int seq_rows, size_t prompt_stride) {
    // seq_rows > 0: group pass -- row m = prompt column step + (m % seq_rows) of the (m / seq_rows)-th staged prompt
    const int m = blockIdx.x, sq = seq_rows > 0 ? m / seq_rows : 0, j = seq_rows > 0 ? m - sq * seq_rows : m;
    embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, cfg->sem_lo, cfg->sem_hi, prompt + (size_t)sq * prompt_stride + state->step + j,
                     state->prompt_L, X + (size_t)m * dim, threadIdx.x, blockDim.x);
}

__global__ void k_advance_n(SeqState* state, int n) {
    state->pos += n;
    state->step += n;
}

// ------------------------------------------------------------------------------------------------ sampling
#if defined(FS_SAMPLE_DBG) && FS_SAMPLE_DBG == 9
__device__ unsigned long long g_dbg_ts[64];
#define FS_TS(i) do { if (threadIdx.x == 0) g_dbg_ts[i] = clock64(); } while (0)
void fs_dbg_read_ts(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_ts), sizeof(unsigned long long) * 64); }
#else
#define FS_TS(i) do {} while (0)
#endif
#include "lm_bsample_dev.h"  // chacha12_word + the block-parallel sampler

// ---- single-thread sequential f32 chains over an LDS array (the sampler's sums must not depend on a reduction order, so
// they are evaluated exactly as the scalar reference does: one running f32 sum in ascending order).  A naive loop pays the
// LDS latency (~100 cycles) per element; these helpers fetch 32 values per step with eight independent 16-byte reads and
// then run the dependent adds out of registers (~10 cycles per element).  `a` must be 16-byte aligned and readable up to
// the next multiple of 32; entries >= n count as +0.0 (x + 0.0f == x exactly for the non-negative sums used here).
__device__ __forceinline__ void lds_fetch32(const float* a, int j, int n, float (&v)[32]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(a + j + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    if (j + 32 > n) {  // tail chunk only: the full chunks run without per-element masking
#pragma unroll
        for (int e = 0; e < 32; ++e) if (j + e >= n) v[e] = 0.f;
    }
}
__device__ float seq_sum(const float* a, int n) {
    float sum = 0.f;
    for (int j = 0; j < n; j += 32) {
        float v[32];
        lds_fetch32(a, j, n, v);
#pragma unroll
        for (int e = 0; e < 32; ++e) sum += v[e];
    }
    return sum;
}
// same chain, also leaving the running sums in cum[0..n) (cum may be written up to the next multiple of 32)
__device__ float seq_sum_prefix(const float* a, int n, float* cum) {
    float sum = 0.f;
    for (int j = 0; j < n; j += 32) {
        float v[32];
        lds_fetch32(a, j, n, v);
#pragma unroll
        for (int e = 0; e < 32; ++e) { sum += v[e]; v[e] = sum; }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(cum + j + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    return sum;
}
// top-p cut over probabilities sorted in descending order: the first rank r at which the running sum of sp[0..r) has
// reached top_p (sampling/mod.rs:117-126); n when it never does
__device__ int seq_topp_cut(const float* sp, int n, float top_p) {
    float cumsum = 0.f;
    for (int j = 0; j < n; j += 32) {
        float v[32];
        lds_fetch32(sp, j, n, v);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            if (j + e >= n) return n;
            if (cumsum >= top_p) return j + e;
            cumsum += v[e];
        }
    }
    return n;
}

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_MAXN = 4096;  // candidates handled by the sampler (audio range 2037, codebook 1024)

// WeightedIndex::new + sample over the contiguous weights w[0..cnt) (ascending token index; zero weights do not move the
// cumulative sum): rand 0.8.5 UniformFloat<f32>::sample_single over [0, total) + partition_point on the cumulative weights.
// Block-wide: thread 0 runs the one sequential f32 chain (leaving the running sums in `cum`), then every thread tests its
// own entries -- the pick is the FIRST non-zero entry whose inclusive running sum exceeds the draw, else the last non-zero
// entry.  `word` = the StdRng word for this draw (computed off the critical path by a side wave).  All threads must call.
__device__ int block_weighted_pick(const float* w, int cnt, float* cum, RngState* rng, uint32_t word) {
    __shared__ float s_chosen;
    __shared__ int s_first, s_last, s_any;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const float total = seq_sum_prefix(w, cnt, cum);
        s_any = total > 0.f ? 1 : 0;
        if (total > 0.f) {
            const float max_rand = __uint_as_float((0xFFFFFFFFu >> 9) | (127u << 23)) - 1.0f;
            float scale = total;
            while (scale * max_rand + 0.f >= total) scale = __uint_as_float(__float_as_uint(scale) - 1u);
            rng->consumed += 1;
            s_chosen = (__uint_as_float((word >> 9) | (127u << 23)) - 1.0f) * scale + 0.f;
        }
        s_first = 0x7FFFFFFF; s_last = -1;
    }
    __syncthreads();
    if (s_any) {
        const float chosen = s_chosen;
        int first = 0x7FFFFFFF, last = -1;
        for (int j = tid; j < cnt; j += SAMPLE_THREADS) {
            if (w[j] == 0.f) continue;
            last = j;
            if (cum[j] > chosen && first == 0x7FFFFFFF) first = j;
        }
        if (first != 0x7FFFFFFF) atomicMin(&s_first, first);
        if (last >= 0) atomicMax(&s_last, last);
    }
    __syncthreads();
    const int res = !s_any ? 0 : (s_first != 0x7FFFFFFF ? s_first : s_last);
    __syncthreads();
    return res;
}

// ---- top-k (k <= 256) sampling of n <= 64 * EPL candidates by ONE wave, no block barrier inside (a barrier phase of a
// 16-wave block costs ~0.4 us on this chip and the sort-based version needed ~40 of them; measured 36 us per call).  Lane l
// owns the EPL consecutive candidates l*EPL .. l*EPL+EPL-1 in registers (ascending index == lane-major order):
//   1. softmax in registers (DPP max, f64 sum);
//   2. the k-th largest probability T by radix select on its bit pattern (8 + 8 + 8 + 6 bits, LDS histogram per pass);
//   3. keep p > T and the first k - #{p > T} ties in index order (v_mbcnt prefix counts), compact the kept set into the
//      contiguous index-ordered arrays kp / ki and 64-bit keys (p bits : 255 - position);
//   4. sort the <= 256 keys descending in registers (4 per lane: in-lane swaps, DPP for lane^1 / lane^2, ds_bpermute above);
//   5. lane 0 runs the ascending-index sum of the kept probabilities while lane 1 runs the descending-order top-p cumsum --
//      two sequential f32 chains in one instruction stream; entries ranked at or after the cut are zeroed.
// wave_topk_select leaves kp / ki in LDS; wave_pick then draws from them.  Decisions are identical to the sorted version
// (top-k ties: lower index first; sequential f32 sums in the reference's order).
// (implementation note, measured with tools/ubench_valu.hip: a lone wave retires a dependent VALU op every ~6 cycles, but a
// VALU result consumed by the SCALAR unit -- v_cmp -> s_bcnt1, ballot -> s_and -- costs ~32 cycles per hop, and a dependent
// ds_bpermute ~70.  Hence: counts and prefix sums stay in vector registers (v_addc, DPP scans), compare-exchanges are
// written as max / min selects, and the sequential sums are pure add chains whose comparisons happen afterwards in parallel.)
// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts, no LDS, no scalar hop)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned long long dpp_xor_lane_u64(unsigned long long v, int which /*1: lane^1, 2: lane^2*/) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (which == 1) { lo = __builtin_amdgcn_mov_dpp(lo, DPP_XOR1, 0xF, 0xF, false); hi = __builtin_amdgcn_mov_dpp(hi, DPP_XOR1, 0xF, 0xF, false); }
    else { lo = __builtin_amdgcn_mov_dpp(lo, DPP_XOR2, 0xF, 0xF, false); hi = __builtin_amdgcn_mov_dpp(hi, DPP_XOR2, 0xF, 0xF, false); }
    return ((unsigned long long)hi << 32) | lo;
}

template <int EPL>
__device__ void wave_topk_select(const float* lg, int n, int kk, float inv_t, float top_p, float* kp, int* ki, float* sp,
                                 unsigned long long* keyb, float* cumsp, bool batch, double top_p64) {
    const int lane = threadIdx.x & 63;
    const int base = lane * EPL;
    uint32_t u[EPL];
    {
        float v[EPL];
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(lg + base + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            v[s] = (base + s < n) ? v[s] * inv_t : -INFINITY;
            mx = fmaxf(mx, v[s]);
        }
        mx = fmaxf(mx, dpp_mov<DPP_XOR1>(mx)); mx = fmaxf(mx, dpp_mov<DPP_XOR2>(mx));
        mx = fmaxf(mx, dpp_mov<DPP_HALF_MIRROR>(mx)); mx = fmaxf(mx, dpp_mov<DPP_MIRROR>(mx));
        mx = fmaxf(fmaxf(readlane(mx, 15), readlane(mx, 31)), fmaxf(readlane(mx, 47), readlane(mx, 63)));
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            v[s] = (base + s < n) ? expf(v[s] - mx) : 0.f;
            part += (double)v[s];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
        const float denom = (float)part;
#pragma unroll
        for (int s = 0; s < EPL; ++s) u[s] = __float_as_uint(v[s] / denom);  // 0 for slots past n
    }
    FS_TS(1);
    // k-th largest value T (#{u > T} < k <= #{u >= T}) by radix select over the 30-bit patterns, 8 + 8 + 8 + 6 bits from the top:
    // per pass the candidates whose higher bits match the prefix are counted into a 256-bin LDS histogram (ds_add, no return),
    // every lane takes 4 bins, a DPP scan gives the counts above each lane, and the bin holding the rank-th candidate extends
    // the prefix -- 4 passes of ~1000 cycles instead of 30 bisection steps of ~380 (each a count over all candidates + a scalar hop)
    uint32_t* hist = reinterpret_cast<uint32_t*>(keyb);  // 256 bins; keyb is only written after T is known
    uint32_t prefix = 0u;
    int krem = kk;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass == 0 ? 22 : (pass == 1 ? 14 : (pass == 2 ? 6 : 0)), bits = pass == 3 ? 6 : 8;
        *reinterpret_cast<uint4*>(hist + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int s2 = 0; s2 < EPL; ++s2)
            if (pass == 0 || (u[s2] >> (shift + bits)) == prefix) atomicAdd(&hist[(u[s2] >> shift) & ((1u << bits) - 1u)], 1u);
        const uint4 hv = *reinterpret_cast<const uint4*>(hist + lane * 4);  // bins 4 * lane .. 4 * lane + 3 (one wave: LDS ops stay in order)
        const int h4[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
        const int mine = (h4[0] + h4[1]) + (h4[2] + h4[3]);
        const int incl = wave_incl_scan(mine);
        int above = __builtin_amdgcn_readlane(incl, 63) - incl;  // candidates in bins of higher lanes
        int found_bin = -1, found_above = 0;
#pragma unroll
        for (int j = 3; j >= 0; --j) {  // from this lane's top bin down: the bin with above < rank <= above + count
            if (found_bin < 0 && above < krem && krem <= above + h4[j]) { found_bin = lane * 4 + j; found_above = above; }
            above += h4[j];
        }
        const unsigned long long m = __ballot(found_bin >= 0);  // exactly one lane (rank <= number of candidates)
        const int src = __builtin_ctzll(m);
        prefix = (prefix << bits) | (uint32_t)__builtin_amdgcn_readlane(found_bin, src);
        krem -= __builtin_amdgcn_readlane(found_above, src);
    }
    const uint32_t lo = prefix;
    const uint32_t T = lo;
    FS_TS(2);
    // keep p > T and the first k - #{p > T} ties in index order (lane-major, then slot)
    const int nvalid = min(max(n - base, 0), EPL);  // only matters for ties at T == 0 (slots past n hold 0 as well)
    int my_gt = 0, my_eq = 0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) { my_gt += u[s] > T ? 1 : 0; my_eq += (u[s] == T && s < nvalid) ? 1 : 0; }
    const int sc_gt = wave_incl_scan(my_gt), sc_eq = wave_incl_scan(my_eq);
    const int r_ties = kk - __builtin_amdgcn_readlane(sc_gt, 63);  // ties to keep (>= 1)
    int run_eq = sc_eq - my_eq;                                    // ties in lower lanes
    uint32_t keepbits = 0u;
    int my_keep = 0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool eq = u[s] == T && s < nvalid;
        const bool keep = u[s] > T || (eq && run_eq < r_ties);
        run_eq += eq ? 1 : 0;
        keepbits |= keep ? (1u << s) : 0u;
        my_keep += keep ? 1 : 0;
    }
    int pos = wave_incl_scan(my_keep) - my_keep;
#pragma unroll
    for (int s = 0; s < EPL; ++s)
        if (keepbits & (1u << s)) {
            kp[pos] = __uint_as_float(u[s]);
            ki[pos] = base + s;
            keyb[pos] = ((unsigned long long)u[s] << 32) | (unsigned long long)(255 - pos);
            ++pos;
        }
    FS_TS(3);
    // sort the kept keys (descending): position i = lane * 4 + s.  Bitonic network with the direction folded into the keys
    // (keys of "ascending" regions are complemented for the duration of a merge phase), so every compare-exchange is the
    // same max / min select.
    unsigned long long k[4];
    {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(keyb + lane * 4), b = *reinterpret_cast<const ulonglong2*>(keyb + lane * 4 + 2);
        k[0] = lane * 4 + 0 < kk ? a.x : 0ull; k[1] = lane * 4 + 1 < kk ? a.y : 0ull;
        k[2] = lane * 4 + 2 < kk ? b.x : 0ull; k[3] = lane * 4 + 3 < kk ? b.y : 0ull;
    }
#pragma unroll
    for (int lk = 1; lk <= 8; ++lk) {       // merge phase K = 1 << lk
        const int K = 1 << lk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // complement the regions that this phase sorts ascending
            const uint32_t f = 0u - (uint32_t)(((lane * 4 + s) >> lk) & 1);
            k[s] ^= ((unsigned long long)f << 32) | f;
        }
#pragma unroll
        for (int j = K >> 1; j > 0; j >>= 1) {
            if (j >= 4) {
                const int lm = j >> 2;
                const bool lower = (lane & lm) == 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned long long o = lm == 1 ? dpp_xor_lane_u64(k[s], 1) : (lm == 2 ? dpp_xor_lane_u64(k[s], 2) : __shfl_xor(k[s], lm, 64));
                    const bool g = k[s] > o;
                    const unsigned long long mxk = g ? k[s] : o, mnk = g ? o : k[s];
                    k[s] = lower ? mxk : mnk;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s & j) continue;
                    const unsigned long long a = k[s], b = k[s ^ j];
                    const bool g = a > b;
                    k[s] = g ? a : b;
                    k[s ^ j] = g ? b : a;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t f = 0u - (uint32_t)(((lane * 4 + s) >> lk) & 1);
            k[s] ^= ((unsigned long long)f << 32) | f;
        }
    }
    *reinterpret_cast<float4*>(sp + lane * 4) = make_float4(__uint_as_float((uint32_t)(k[0] >> 32)), __uint_as_float((uint32_t)(k[1] >> 32)),
                                                            __uint_as_float((uint32_t)(k[2] >> 32)), __uint_as_float((uint32_t)(k[3] >> 32)));
    FS_TS(4);
    // two sequential f32 chains in one instruction stream, pure adds: lane 0 sums kp (ascending index), lane 1 walks sp
    // (descending order) and leaves its running sums in cumsp; the top-p cut is then found in parallel:
    // cut = first rank r whose EXCLUSIVE running sum is >= top_p  ==  1 + first q with inclusive sum[q] >= top_p
    const float* arr = lane == 1 ? sp : kp;
    float cum = 0.f;
    for (int j = 0; j < kk; j += 32) {
        float v[32];
        lds_fetch32(arr, j, kk, v);
#pragma unroll
        for (int e = 0; e < 32; ++e) { cum += v[e]; v[e] = cum; }
        if (lane == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(cumsp + j + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
    const float sum_p = readlane(cum, 0);
    const bool do_topp = batch ? !(top_p64 <= 0.0 || top_p64 >= (double)sum_p) : !(top_p <= 0.f || top_p >= sum_p);  // sampling/mod.rs:68
    FS_TS(5);
    if (do_topp) {  // zero every prob once the running cumsum (descending order) reached top_p
        const float4 cq = *reinterpret_cast<const float4*>(cumsp + lane * 4);
        const float cs[4] = {cq.x, cq.y, cq.z, cq.w};
        int first = 0x7FFFFFFF;
#pragma unroll
        for (int s = 3; s >= 0; --s) if (lane * 4 + s < kk && cs[s] >= top_p) first = lane * 4 + s;
        const unsigned long long mh = __ballot(first != 0x7FFFFFFF);
        const int cutv = mh ? __builtin_amdgcn_readlane(first, __builtin_ctzll(mh)) + 1 : kk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = lane * 4 + s;
            if (i < kk && i >= cutv) kp[255 - (int)(k[s] & 0xFFull)] = 0.f;
        }
    }
}

// WeightedIndex::new + sample over the contiguous weights w[0..cnt), cnt <= 256, by one wave.  Zero weights (entries cut by top-p)
// do not move the cumulative f32 sum (x + 0 == x exactly) and are never picked, so the sequential chain only walks the NON-ZERO
// weights, compacted in ascending index order first (DPP prefix scan; `cval` / `cpos` = LDS scratch for <= 256 floats / ints):
// after a top-p cut that is typically a few dozen of the 256 entries.  Every lane runs the same chain (lane 0 leaves the running
// sums in `cum`), then each lane tests its four compacted entries.
__device__ int wave_pick(const float* w, int cnt, float* cum, RngState* rng, uint32_t word, float* cval, int* cpos) {
    const int lane = threadIdx.x & 63;
    const float4 wv = *reinterpret_cast<const float4*>(w + lane * 4);
    const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
    int mine = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) mine += (lane * 4 + s < cnt && ws[s] != 0.f) ? 1 : 0;
    const int incl = wave_incl_scan(mine);
    const int m = __builtin_amdgcn_readlane(incl, 63);  // non-zero weights
    if (m == 0) return 0;
    int pos = incl - mine;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (lane * 4 + s < cnt && ws[s] != 0.f) { cval[pos] = ws[s]; cpos[pos] = lane * 4 + s; ++pos; }
    float total = 0.f;
    for (int j = 0; j < m; j += 32) {
        float v[32];
        lds_fetch32(cval, j, m, v);
#pragma unroll
        for (int e = 0; e < 32; ++e) { total += v[e]; v[e] = total; }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(cum + j + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
    if (!(total > 0.f)) return 0;
    const float max_rand = __uint_as_float((0xFFFFFFFFu >> 9) | (127u << 23)) - 1.0f;
    float scale = total;
    while (scale * max_rand + 0.f >= total) scale = __uint_as_float(__float_as_uint(scale) - 1u);
    if (lane == 0) rng->consumed += 1;
    const float chosen = (__uint_as_float((word >> 9) | (127u << 23)) - 1.0f) * scale + 0.f;
    const float4 cv = *reinterpret_cast<const float4*>(cum + lane * 4);
    const float cs[4] = {cv.x, cv.y, cv.z, cv.w};
    int first = 0x7FFFFFFF;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int j = lane * 4 + s;
        if (j < m && cs[s] > chosen && first == 0x7FFFFFFF) first = j;  // first kept item whose inclusive cumulative weight is > chosen
    }
    const unsigned long long mh = __ballot(first != 0x7FFFFFFF);
    const int jsel = mh ? __builtin_amdgcn_readlane(first, __builtin_ctzll(mh)) : m - 1;  // else the last non-zero item
    return cpos[jsel];
}

// Block-wide selection of one index from `n` logits held in LDS (already penalised / masked).
//  temp == 0: host-ArgMax rule of candle's LogitsProcessor (max_by(total_cmp)): LAST maximal index wins.
//  temp  > 0: softmax(logits / temp) -> top-k (ties: lower index first) -> top-p -> WeightedIndex draw, evaluated in
//             ascending-index order with the StdRng stream (sampling/mod.rs:51-132).  Identical decision procedure to
//             oracle::LogitsProcessor::sample; the softmax denominator is accumulated in f64 on both sides so that
//             the result does not depend on reduction order.
__device__ int block_sample(float* lg /*LDS [n]*/, int n, const SampleCfg& c, RngState* rng, float* sp /*LDS [SAMPLE_MAXN]*/,
                            int* si /*LDS [SAMPLE_MAXN]*/, double* red /*LDS [SAMPLE_THREADS]*/, bool first_max = false) {
    const int tid = threadIdx.x;
    __shared__ int s_result;
    if (c.temp == 0.f) {
        // argmax with the host rule (LAST maximal index) or the device rule (FIRST): per-thread scan, DPP/readlane wave
        // reduction of the value, ballot-free index pick, then one LDS hop across the waves
        float bv = -INFINITY;
        int bi = -1;
        for (int i = tid; i < n; i += SAMPLE_THREADS) {
            const float v = lg[i];
            if (bi < 0 || (first_max ? (v > bv) : !(v < bv))) { bv = v; bi = i; }  // ascending i per thread
        }
        float wm = bv;
        wm = fmaxf(wm, dpp_mov<DPP_XOR1>(wm)); wm = fmaxf(wm, dpp_mov<DPP_XOR2>(wm));
        wm = fmaxf(wm, dpp_mov<DPP_HALF_MIRROR>(wm)); wm = fmaxf(wm, dpp_mov<DPP_MIRROR>(wm));
        wm = fmaxf(fmaxf(readlane(wm, 15), readlane(wm, 31)), fmaxf(readlane(wm, 47), readlane(wm, 63)));
        int cand = (bi >= 0 && bv == wm) ? bi : (first_max ? 0x7FFFFFFF : -1);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(cand, m, 64);
            cand = first_max ? min(cand, o) : max(cand, o);
        }
        float* rv = reinterpret_cast<float*>(red);
        int* ri = reinterpret_cast<int*>(red) + 64;
        const int wv = tid >> 6;
        if ((tid & 63) == 0) { rv[wv] = wm; ri[wv] = cand; }
        __syncthreads();
        if (tid == 0) {
            float gv = rv[0];
            int gi = ri[0];
            for (int w2 = 1; w2 < SAMPLE_THREADS / 64; ++w2) {
                const float v2 = rv[w2];
                const int i2 = ri[w2];
                if (v2 > gv || (v2 == gv && (first_max ? i2 < gi : i2 > gi))) { gv = v2; gi = i2; }
            }
            s_result = gi;
        }
        __syncthreads();
        const int res = s_result;
        __syncthreads();
        return res;
    }
    {
        const bool use_k0 = c.top_k > 0 && c.top_k < n;
        if (use_k0 && c.top_k <= 256 && n <= 2048) {
            // one-wave path (see wave_topk_select): wave 0 selects, the last wave computes this draw's StdRng word meanwhile
            const float inv_t0 = (float)(1.0 / (double)c.temp);
            __shared__ uint32_t s_word0;
            __shared__ __attribute__((aligned(16))) float w_kp[256 + 32];
            __shared__ __attribute__((aligned(16))) float w_cum[256 + 32];
            __shared__ __attribute__((aligned(16))) unsigned long long w_key[256];
            __shared__ int w_ki[256];
            const int kk0 = c.top_k;
            FS_TS(0);
            if (tid < 64) {
                if (n <= 1024) wave_topk_select<16>(lg, n, kk0, inv_t0, c.top_p, w_kp, w_ki, sp, w_key, w_cum, first_max, c.top_p64);
                else wave_topk_select<32>(lg, n, kk0, inv_t0, c.top_p, w_kp, w_ki, sp, w_key, w_cum, first_max, c.top_p64);
            } else if (tid == SAMPLE_THREADS - 1) {
                s_word0 = chacha12_word(rng->key, rng->consumed);
            }
            FS_TS(6);
            __syncthreads();
            FS_TS(7);
            if (tid < 64) {
                const int pick = wave_pick(w_kp, kk0, w_cum, rng, s_word0, sp, si);  // sp / si: free after the sort
                if (tid == 0) s_result = w_ki[pick];
            }
            FS_TS(8);
            __syncthreads();
            const int res = s_result;
            __syncthreads();
            return res;
        }
    }
    // softmax(logits * (1/temp)).  Blocked ownership: thread t owns the `ept` consecutive candidates t*ept .. t*ept+ept-1
    // (ascending index order == thread-major order, which the index-order prefix scans below rely on).
    FS_TS(0);
    const float inv_t = (float)(1.0 / (double)c.temp);
    const int ept = (n + SAMPLE_THREADS - 1) / SAMPLE_THREADS;  // 1..4
    const int lane = tid & 63, wv = tid >> 6;
    // the StdRng word of this call's draw: ~800 dependent integer ops, computed by the last wave while the others select
    __shared__ uint32_t s_word;
    if (tid == SAMPLE_THREADS - 1) s_word = chacha12_word(rng->key, rng->consumed);
    float pv[4];
    bool valid[4];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = tid * ept + s;
        valid[s] = s < ept && i < n;
        pv[s] = valid[s] ? lg[i] * inv_t : -INFINITY;
        mx = fmaxf(mx, pv[s]);
    }
    float* rv = reinterpret_cast<float*>(red);
    mx = fmaxf(mx, dpp_mov<DPP_XOR1>(mx)); mx = fmaxf(mx, dpp_mov<DPP_XOR2>(mx));
    mx = fmaxf(mx, dpp_mov<DPP_HALF_MIRROR>(mx)); mx = fmaxf(mx, dpp_mov<DPP_MIRROR>(mx));
    mx = fmaxf(fmaxf(readlane(mx, 15), readlane(mx, 31)), fmaxf(readlane(mx, 47), readlane(mx, 63)));
    if (lane == 0) rv[wv] = mx;
    __syncthreads();
#pragma unroll
    for (int w2 = 0; w2 < SAMPLE_THREADS / 64; ++w2) mx = fmaxf(mx, rv[w2]);
    __syncthreads();
    double part = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        pv[s] = valid[s] ? expf(pv[s] - mx) : 0.f;
        part += (double)pv[s];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    double dsum = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < SAMPLE_THREADS / 64; ++w2) dsum += red[w2];
    const float denom = (float)dsum;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) pv[s] = pv[s] / denom;  // probabilities (by index, in registers)
    FS_TS(1);
    const bool use_k = c.top_k > 0 && c.top_k < n;
    const int kk = use_k ? c.top_k : n;
    __shared__ int s_cut;       // number of leading sorted entries that survive top-p
    __shared__ int s_do_topp;
    // ---- general path (no top-k, k > 256, or more than 2048 candidates): full bitonic sort by (prob desc, index asc) over the next power of two
#pragma unroll
    for (int s = 0; s < 4; ++s) if (valid[s]) lg[tid * ept + s] = pv[s];
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    __syncthreads();
    for (int i = tid; i < np2; i += SAMPLE_THREADS) {
        if (i < n) { sp[i] = lg[i]; si[i] = i; } else { sp[i] = -1.f; si[i] = 0x7FFFFFFF; }
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
    // ---- tail.  Decision procedure and f32 rounding identical to the oracle restatement: sums and the WeightedIndex scan
    // run in ascending token-index order, the top-p cut in descending probability order, every sum is a sequential f32
    // chain.  To keep those chains short and free of dependent LDS indirections, the kept set (top_k entries, or all n)
    // is materialised as CONTIGUOUS arrays: kp[j] = probability of the j-th kept token in index order, ki[j] = its index.
    int* ki = reinterpret_cast<int*>(red);  // 8 KB scratch: up to 2048 ints
    float* kp = lg;                          // the by-index array is no longer needed once (sp, si) are sorted
    const bool small = use_k && kk <= 2 * SAMPLE_THREADS;
    int cnt;                                 // entries of (ki, kp)
    if (small) {
        int kp2 = 1;
        while (kp2 < kk) kp2 <<= 1;
        __syncthreads();
        for (int r = tid; r < kp2; r += SAMPLE_THREADS) { ki[r] = r < kk ? si[r] : 0x7FFFFFFF; kp[r] = r < kk ? sp[r] : 0.f; }
        __syncthreads();
        for (int k2 = 2; k2 <= kp2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < kp2; i += SAMPLE_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const int a0 = ki[i], a1 = ki[ixj];
                        const bool up = (i & k2) == 0;
                        if ((a0 > a1) == up) {
                            ki[i] = a1; ki[ixj] = a0;
                            const float t0 = kp[i]; kp[i] = kp[ixj]; kp[ixj] = t0;
                        }
                    }
                }
                __syncthreads();
            }
        cnt = kk;
    } else {  // all n tokens (or a top-k too large for the scratch): index order is the identity
        __syncthreads();
        for (int i = tid; i < n; i += SAMPLE_THREADS) kp[i] = 0.f;
        __syncthreads();
        for (int r = tid; r < kk; r += SAMPLE_THREADS) kp[si[r]] = sp[r];
        __syncthreads();
        cnt = n;
    }
    if (tid == 0) {
        bool do_topp = true;
        if (use_k) {
            const float sum_p = seq_sum(kp, cnt);  // ascending index; entries outside the top-k are 0 (or absent)
            do_topp = first_max ? !(c.top_p64 <= 0.0 || c.top_p64 >= (double)sum_p) : !(c.top_p <= 0.f || c.top_p >= sum_p);  // (first_max == batch semantics)
        }
        // zero every prob once the running cumsum (descending order) reached top_p
        s_cut = do_topp ? seq_topp_cut(sp, kk, c.top_p) : kk;
        s_do_topp = do_topp ? 1 : 0;
    }
    __syncthreads();
    if (s_do_topp && s_cut < kk) {  // entries sorting at or after rank `cut` are zeroed (parallel predicate on (prob, index))
        const float pc = sp[s_cut];
        const int ic = si[s_cut];
        for (int j = tid; j < cnt; j += SAMPLE_THREADS) {
            const float pj = kp[j];
            const int ij = small ? ki[j] : j;
            if (pj < pc || (pj == pc && ij >= ic)) kp[j] = 0.f;
        }
    }
    __syncthreads();
    {
        const int r = block_weighted_pick(kp, cnt, sp, rng, s_word);
        const int res = small ? ki[r] : r;
        __syncthreads();
        return res;
    }
}

// Greedy pick (temp == 0, host ArgMax rule: LAST maximal index) without the three block barriers of block_sample: a thread's
// candidates (indices tid + j * SAMPLE_THREADS) stay in registers, a wave reduces (value, then index) by DPP, lane 0 of every wave
// does one 64-bit LDS atomicMax on {order-preserving value bits : index}, ONE barrier, everybody reads the winner.  *s_key must
// have been zeroed before the previous barrier.  (A 16-wave barrier phase costs ~0.4 us on this chip.)
__device__ __forceinline__ int dpp_wave_max_int(int v) {
    v = max(v, __builtin_amdgcn_mov_dpp(v, DPP_XOR1, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_mov_dpp(v, DPP_XOR2, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_mov_dpp(v, DPP_HALF_MIRROR, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_mov_dpp(v, DPP_MIRROR, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)), max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
__device__ __forceinline__ int greedy_pick(const float (&val)[SAMPLE_MAXN / SAMPLE_THREADS], int n, unsigned long long* s_key) {
    const int tid = threadIdx.x;
    float bv = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int j = 0; j < SAMPLE_MAXN / SAMPLE_THREADS; ++j) {
        const int i = tid + j * SAMPLE_THREADS;
        if (i < n && (bi < 0 || !(val[j] < bv))) { bv = val[j]; bi = i; }  // ascending i per thread: the later equal value wins
    }
    float wm = bv;
    wm = fmaxf(wm, dpp_mov<DPP_XOR1>(wm)); wm = fmaxf(wm, dpp_mov<DPP_XOR2>(wm));
    wm = fmaxf(wm, dpp_mov<DPP_HALF_MIRROR>(wm)); wm = fmaxf(wm, dpp_mov<DPP_MIRROR>(wm));
    wm = fmaxf(fmaxf(readlane(wm, 15), readlane(wm, 31)), fmaxf(readlane(wm, 47), readlane(wm, 63)));
    const int ci = dpp_wave_max_int((bi >= 0 && bv == wm) ? bi : -1);
    if ((tid & 63) == 0 && ci >= 0) {
        uint32_t u = __float_as_uint(wm);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // unsigned order == float order
        atomicMax(s_key, ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)ci);
    }
    __syncthreads();
    return (int)(*s_key & 0xFFFFFFFFull);
}

__device__ inline void child_rng(const RngState* master, unsigned long long n64, RngState* out);

template <typename WT>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_slow(const float* __restrict__ logits, int n,
                                                                const SampleCfg* __restrict__ cp, RngState* rng, SeqState* __restrict__ state,
                                                                const float* __restrict__ x, float* __restrict__ xf, int dim,
                                                                float* const* __restrict__ hid_slot) {
    __shared__ __attribute__((aligned(16))) float lg[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) float sp[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    const int tid = threadIdx.x;
    const SampleCfg c = *cp;
    // generate_blocking_with_hidden (single_batch.rs:250,264-266): the hidden state of every generator iteration, the terminating one
    // included, row = iteration index; replays after termination (done != 0 on entry) write nothing
    float* hid = hid_slot ? *hid_slot : nullptr;
    if (hid && state->done != 0) hid = nullptr;
    if (hid) hid += (size_t)state->frame * dim;
    __shared__ unsigned long long s_key;
    if (c.batch_rows > 0 && !c.legacy) {  // row `batch_row` of a static batch on the single-sequence path: sampling/mod.rs:77-109
        __shared__ RngState lrng;
        SampleCfg cb = c;
        if (cb.temp <= 1e-7f) cb.temp = 0.f;
        for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = logits[i];
        for (int i = tid; i < dim; i += SAMPLE_THREADS) { const float h = x[i]; xf[i] = h; if (hid) hid[i] = h; }
        if (tid == 0 && cb.temp != 0.f)
            child_rng(rng, (unsigned long long)state->frame * (unsigned long long)c.batch_calls * c.batch_rows + c.batch_row, &lrng);
        __syncthreads();
        if (cb.ignore_eos && tid == 0) lg[0] = -INFINITY;
        __syncthreads();
        const int idx = block_sample(lg, n, cb, &lrng, sp, si, red, /*first_max=*/true);
        if (tid == 0) {
            uint32_t tok = audio_tok(c, idx);
            if (state->done) tok = c.im_end_id;
            state->cur[0] = tok;
            if (tok == c.im_end_id && state->done == 0) state->done = 1;
        }
        return;
    }
    if (c.temp == 0.f && !c.legacy) {  // greedy: two barriers instead of five (see greedy_pick)
        if (tid == 0) s_key = 0ull;
        float val[SAMPLE_MAXN / SAMPLE_THREADS];
#pragma unroll
        for (int j = 0; j < SAMPLE_MAXN / SAMPLE_THREADS; ++j) {
            const int i = tid + j * SAMPLE_THREADS;
            val[j] = i < n ? logits[i] : -INFINITY;
            if (i == 0 && c.ignore_eos) val[j] = -INFINITY;
        }
        for (int i = tid; i < dim; i += SAMPLE_THREADS) { const float h = x[i]; xf[i] = h; if (hid) hid[i] = h; }  // hidden_states -> fast decoder input (:149)
        __syncthreads();
        const int idx = greedy_pick(val, n, &s_key);
        if (tid == 0) {
            uint32_t tok = audio_tok(c, idx);  // rescale_semantic_tokens (utils.rs:45-46)
            if (state->done) tok = c.im_end_id;
            state->cur[0] = tok;
            if (tok == c.im_end_id && state->done == 0) state->done = 1;
        }
        return;
    }
    for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = logits[i];
    for (int i = tid; i < dim; i += SAMPLE_THREADS) { const float h = x[i]; xf[i] = h; if (hid) hid[i] = h; }  // hidden_states -> fast decoder input (:149)
    __syncthreads();
    if (c.legacy) {
        // legacy_softmax_sample (sampling/mod.rs:8-26): P(pad) = softmax([pad, eos])[0]; u ~ U[0,1) = (next_u32 >> 8) * 2^-24
        // (rand Standard<f32>).  The reference draws from an unseeded thread_rng; here the draw comes from the request's
        // seeded StdRng stream so that runs are reproducible.
        if (tid == 0) {
            const float pad = lg[0], eos = lg[1], m = fmaxf(pad, eos);
            const float e_pad = expf(pad - m), e_eos = expf(eos - m);
            const float p_pad = e_pad / (e_pad + e_eos);
            const uint32_t w = chacha12_word(rng->key, rng->consumed);
            rng->consumed += 1;
            const float u = (float)(w >> 8) * (1.0f / 16777216.0f);
            uint32_t tok = (u < p_pad || c.ignore_eos) ? c.pad_id : c.im_end_id;
            if (state->done) tok = c.im_end_id;
            state->cur[0] = tok;
            if (tok == c.im_end_id && state->done == 0) state->done = 1;
        }
        return;
    }
    if (c.ignore_eos && tid == 0) lg[0] = -INFINITY;
    __syncthreads();
    const int idx = block_sample(lg, n, c, rng, sp, si, red);
    if (tid == 0) {
        uint32_t tok = audio_tok(c, idx);  // rescale_semantic_tokens (utils.rs:45-46)
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
    __shared__ __attribute__((aligned(16))) float lg[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) float sp[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    const int tid = threadIdx.x;
    const int n = cb_size;
    const SampleCfg c = *cp;
    // one round trip for everything the decision needs: state words, the repetition-penalty ring, logits and mask
    __shared__ int s_ring[17], s_meta[2];
    __shared__ uint32_t s_prev, s_cur0, s_have_prev;
    __shared__ unsigned long long s_key;
    // batch_rows > 0: BatchedLogitsProcessor semantics (see k_sample_slow): first-max argmax at temp <= 1e-7, else the child StdRng of
    // (frame, codebook call, row)
    const bool bm = c.batch_rows > 0;
    SampleCfg cc = c;
    if (bm && cc.temp <= 1e-7f) cc.temp = 0.f;
    const bool greedy = cc.temp == 0.f && !bm;
    __shared__ RngState lrng;
    if (bm && tid == 23 && cc.temp != 0.f)
        child_rng(rng, ((unsigned long long)state->frame * (unsigned long long)(n_cb + 1) + 1ull + (unsigned long long)cb) * c.batch_rows + c.batch_row, &lrng);
    if (tid == 22) s_key = 0ull;
    if (tid < 17) s_ring[tid] = rp.ring[cb * 17 + tid];
    else if (tid < 19) s_meta[tid - 17] = rp.ring_meta[cb * 2 + tid - 17];
    else if (tid == 19) s_prev = state->prev[cb + 1];
    else if (tid == 20) s_cur0 = state->cur[0];
    else if (tid == 21) s_have_prev = (uint32_t)state->have_prev;
    float* mask = rp.mask + (size_t)cb * cb_size;
    float lv[SAMPLE_MAXN / SAMPLE_THREADS], mv[SAMPLE_MAXN / SAMPLE_THREADS];
#pragma unroll
    for (int j = 0; j < SAMPLE_MAXN / SAMPLE_THREADS; ++j) {
        const int i = tid + j * SAMPLE_THREADS;
        lv[j] = i < n ? logits[i] : 0.f;
        mv[j] = i < n ? mask[i] : 1.f;
    }
    __syncthreads();
    const bool eos = s_cur0 == c.im_end_id;  // single_batch.rs:153-156: push 0, skip the fast step
    if (!eos) {
        const bool pen = s_have_prev != 0;
        // SingleBatchedRepPenProcessor::apply (rep_pen.rs:37-65) on the register copy of the mask.  "token in tokens_seen"
        // == "mask[token] == penalty" (set on insert, reset to 1 on removal; with penalty == 1 the mask never changes).
        int last = -1, dropped = -1;
        if (pen) {
            last = (int)s_prev;
            const int head = (s_meta[0] + 16) % 17, len = s_meta[1] + 1;  // push_front
            const bool drop = len > 16;
            if (drop) dropped = s_ring[(head + len - 1) % 17];            // pop_back (never the slot just written)
            if (tid == 0) { rp.ring[cb * 17 + head] = last; rp.ring_meta[cb * 2] = head; rp.ring_meta[cb * 2 + 1] = drop ? 16 : len; }
        }
#pragma unroll
        for (int j = 0; j < SAMPLE_MAXN / SAMPLE_THREADS; ++j) {
            const int i = tid + j * SAMPLE_THREADS;
            if (i < n) {
                float m = mv[j];
                if (pen) {
                    const float m0 = m;
                    if (i == last) m = c.rep_pen;
                    if (i == dropped && m == c.rep_pen) m = 1.0f;
                    if (m != m0) mask[i] = m;
                }
                lv[j] = pen ? lv[j] / m : lv[j];
                if (!greedy) lg[i] = lv[j];
            }
        }
        if (!greedy) __syncthreads();
    }
    int code = 0;
    if (!eos) code = greedy ? greedy_pick(lv, n, &s_key) : block_sample(lg, n, cc, bm ? &lrng : rng, sp, si, red, /*first_max=*/bm);
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

// ------------------------------------------------------------------------------------------------ batched (static-batch) sampling
// generate/static_batch.rs:117-274 + sampling/mod.rs:77-109: one block per batch row.  temp <= 1e-7 -> device argmax
// (FIRST maximal index); else softmax(logits / temp) and, per sample() call, every row draws from its OWN child StdRng
// seeded with the next u64 of the master StdRng (sampling/mod.rs:93-95): call c of the request, row b uses master u64
// number c * B + b, and the single WeightedIndex draw consumes word 0 of the child stream.
__device__ inline void child_rng(const RngState* master, unsigned long long n64, RngState* out) {
    const unsigned long long lo = chacha12_word(master->key, 2 * n64), hi = chacha12_word(master->key, 2 * n64 + 1);
    unsigned long long state = (hi << 32) | lo;
    for (int i = 0; i < 8; ++i) {  // rand_core seed_from_u64 (PCG32 expansion)
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        out->key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    out->consumed = 0;
}

template <typename WT>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_slow_rows(const float* __restrict__ logits, int ld, int n,
                                                                     const SampleCfg* __restrict__ cp, const RngState* __restrict__ master,
                                                                     int B, int calls_per_frame, SeqState* __restrict__ states,
                                                                     const float* __restrict__ X, float* __restrict__ XF, int dim, PrepOut po) {
    __shared__ float red4[4];
    __shared__ __attribute__((aligned(16))) float lg[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) float sp[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    __shared__ RngState lrng;
    const int tid = threadIdx.x, b = blockIdx.x;
    SeqState* st = states + b;
    SampleCfg c = *cp;
    if (c.temp <= 1e-7f) c.temp = 0.f;  // sampling/mod.rs:80
    for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = logits[(size_t)b * ld + i];
    for (int i = tid; i < dim; i += SAMPLE_THREADS) XF[(size_t)b * dim + i] = X[(size_t)b * dim + i];  // hidden_states (:175)
    __syncthreads();
    if (c.ignore_eos && tid == 0) lg[0] = -INFINITY;
    __syncthreads();
    // the child StdRng (two ChaCha12 blocks + the PCG expansion: ~3 us of dependent integer work) is derived by the thread that also
    // computes the draw's word inside block_sample -- the last one -- while wave 0 already selects; nobody else reads lrng before the
    // barrier in front of the pick
    if (tid == SAMPLE_THREADS - 1 && c.temp != 0.f) child_rng(master, (unsigned long long)st->frame * calls_per_frame * B + b, &lrng);
    const int idx = block_sample(lg, n, c, &lrng, sp, si, red, /*first_max=*/true);
    if (tid == 0) {
        const uint32_t tok = audio_tok(c, idx);  // rescale_semantic_tokens (utils.rs:45-46)
        st->cur[0] = tok;
        if (tok == c.im_end_id) st->done = 1;  // batch_item_is_dead |= newly dead (:160-173)
    }
    if (po.epoch && b == 0 && tid == 0) po.epoch[0] += 1;
    if (po.g) block_prep_row(XF + (size_t)b * dim, dim, po, b, red4);
}

template <typename WT>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_fast_rows(const float* __restrict__ logits, int cb, int n_cb, int cb_size,
                                                                     const SampleCfg* __restrict__ cp, const RngState* __restrict__ master,
                                                                     int B, SeqState* __restrict__ states, const WT* __restrict__ fast_emb,
                                                                     float* __restrict__ XF, const WT* __restrict__ tok_emb,
                                                                     const WT* __restrict__ cb_emb, float* __restrict__ X, int dim,
                                                                     uint32_t* __restrict__ out_codes, int out_cap, PrepOut po) {
    __shared__ float red4[4];
    __shared__ __attribute__((aligned(16))) float lg[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) float sp[SAMPLE_MAXN];
    __shared__ __attribute__((aligned(16))) int si[SAMPLE_MAXN];
    __shared__ double red[SAMPLE_THREADS];
    __shared__ RngState lrng;
    const int tid = threadIdx.x, b = blockIdx.x, n = cb_size;
    SeqState* st = states + b;
    SampleCfg c = *cp;
    if (c.temp <= 1e-7f) c.temp = 0.f;
    // the batch repetition-penalty mask is never updated for Fish models (static_batch.rs:204-206): logits / 1.0
    for (int i = tid; i < n; i += SAMPLE_THREADS) lg[i] = logits[(size_t)b * n + i];
    __syncthreads();
    if (tid == SAMPLE_THREADS - 1 && c.temp != 0.f)  // (see k_sample_slow_rows: off wave 0's critical path)
        child_rng(master, ((unsigned long long)st->frame * (n_cb + 1) + 1 + cb) * B + b, &lrng);
    const int code = block_sample(lg, n, c, &lrng, sp, si, red, /*first_max=*/true);
    if (tid == 0) st->cur[cb + 1] = (uint32_t)code;
    if (cb != n_cb - 1) {
        for (int d = tid; d < dim; d += SAMPLE_THREADS) XF[(size_t)b * dim + d] = WTr<WT>::to_f32(fast_emb[(size_t)code * dim + d]);
        if (po.g) block_prep_row(XF + (size_t)b * dim, dim, po, b, red4);
        return;
    }
    // ---- end of frame (static_batch.rs:224-267 + generate_static_batch :305-338)
    __syncthreads();
    __shared__ uint32_t cur[16];
    if (tid <= n_cb) {
        const uint32_t slow = st->cur[0];
        const bool is_audio = slow >= c.sem_lo;  // :229 (non-audio rows carry zero codes)
        uint32_t v = tid == 0 ? slow : (tid == n_cb ? (uint32_t)code : st->cur[tid]);
        if (tid > 0 && !is_audio) v = 0;
        cur[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const int frame = st->frame;
        // session slots (SampleCfg::session): a dead slot is frozen -- it neither emits nor advances -- from the frame after its last one
        const bool frozen = c.session != 0 && st->done != 0 && frame > 0;
        if (!frozen) {
            if (frame == 0 || !st->done) {  // first position unconditionally, then only while the row is active
                if (frame == 0 && cur[0] < c.sem_lo) st->step = -1;  // BatchPosition::is_audio of the first position (static_batch.rs:229): `step` is free during decode
                const int o = st->n_out;
                uint32_t* oc = out_codes + (size_t)b * n_cb * out_cap;
                if (o < out_cap)
                    for (int cc = 0; cc < n_cb; ++cc) oc[(size_t)cc * out_cap + o] = cur[cc + 1];
                st->n_out = o + 1;
            }
            for (int i = 0; i <= n_cb; ++i) { st->prev[i] = cur[i]; st->cur[i] = cur[i]; }
            st->have_prev = 1;
            st->pos += 1;  // dead rows keep stepping in lock-step (:255-261)
            st->frame = frame + 1;
        }
    }
    embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, c.sem_lo, c.sem_hi, cur, 1, X + (size_t)b * dim, tid, SAMPLE_THREADS);
}

// ---- the batched samplers on the block-parallel sampler (lm_bsample_dev.h): 512 threads per row, for temp > 1e-7 with 0 < top_k <= 256
// (BASELINE configs[2]: top-k 256 / top-p 0.8).  The per-(call, row) child StdRng derivation -- two ChaCha12 blocks, the PCG expansion
// and the word of the draw: ~5 us of dependent integer work -- no longer hides behind a 13 us one-wave selection, so it runs once per
// step for all the step's calls: k_rows_rng_words, thread c of block b = call c of row b (the same master u64 numbers as above).
constexpr int ROWS_WORDS_LD = 16;
__global__ __launch_bounds__(64) void k_rows_rng_words(const RngState* __restrict__ master, int B, int calls_per_frame,
                                                       const SeqState* __restrict__ states, uint32_t* __restrict__ words) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= calls_per_frame) return;
    RngState child;
    child_rng(master, ((unsigned long long)states[b].frame * calls_per_frame + c) * B + b, &child);
    words[b * ROWS_WORDS_LD + c] = chacha12_word(child.key, 0);
}
constexpr int PAR_THREADS = 512;
template <typename WT>
__global__ __launch_bounds__(PAR_THREADS) void k_sample_slow_rows_par(const float* __restrict__ logits, int ld, int n, const SampleCfg* __restrict__ cp,
                                                                       const uint32_t* __restrict__ words, SeqState* __restrict__ states,
                                                                       const float* __restrict__ X, float* __restrict__ XF, int dim, PrepOut po) {
    __shared__ float red4[4];
    __shared__ BSampLds S;
    const int tid = threadIdx.x, b = blockIdx.x;
    SeqState* st = states + b;
    const SampleCfg c = *cp;
    float lv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = tid * 4 + s;
        lv[s] = i < n ? logits[(size_t)b * ld + i] : 0.f;
        if (i == 0 && c.ignore_eos) lv[s] = -INFINITY;
    }
    for (int i = tid; i < dim; i += PAR_THREADS) XF[(size_t)b * dim + i] = X[(size_t)b * dim + i];  // hidden_states (:175)
    int used = 0;
    const int idx = bsample<PAR_THREADS, 4>(lv, n, c.top_k, (float)(1.0 / (double)c.temp), c.top_p, words[b * ROWS_WORDS_LD], &used, S, true, c.top_p64);
    if (tid == 0) {
        const uint32_t tok = audio_tok(c, idx);  // rescale_semantic_tokens (utils.rs:45-46)
        st->cur[0] = tok;
        if (tok == c.im_end_id) st->done = 1;  // batch_item_is_dead |= newly dead (:160-173)
    }
    if (po.epoch && b == 0 && tid == 0) po.epoch[0] += 1;
    if (po.g) block_prep_row(XF + (size_t)b * dim, dim, po, b, red4);
}
template <typename WT>
__global__ __launch_bounds__(PAR_THREADS) void k_sample_fast_rows_par(const float* __restrict__ logits, int cb, int n_cb, int cb_size,
                                                                       const SampleCfg* __restrict__ cp, const uint32_t* __restrict__ words, SeqState* __restrict__ states,
                                                                       const WT* __restrict__ fast_emb, float* __restrict__ XF, const WT* __restrict__ tok_emb,
                                                                       const WT* __restrict__ cb_emb, float* __restrict__ X, int dim,
                                                                       uint32_t* __restrict__ out_codes, int out_cap, PrepOut po) {
    __shared__ float red4[4];
    __shared__ BSampLds S;
    const int tid = threadIdx.x, b = blockIdx.x, n = cb_size;
    SeqState* st = states + b;
    const SampleCfg c = *cp;
    // the batch repetition-penalty mask is never updated for Fish models (static_batch.rs:204-206): logits / 1.0
    float lv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { const int i = tid * 2 + s; lv[s] = i < n ? logits[(size_t)b * n + i] : 0.f; }
    int used = 0;
    const int code = bsample<PAR_THREADS, 2>(lv, n, c.top_k, (float)(1.0 / (double)c.temp), c.top_p, words[b * ROWS_WORDS_LD + 1 + cb], &used, S, true, c.top_p64);
    if (tid == 0) st->cur[cb + 1] = (uint32_t)code;
    if (cb != n_cb - 1) {
        for (int d = tid; d < dim; d += PAR_THREADS) XF[(size_t)b * dim + d] = WTr<WT>::to_f32(fast_emb[(size_t)code * dim + d]);
        if (po.g) block_prep_row(XF + (size_t)b * dim, dim, po, b, red4);
        return;
    }
    // ---- end of frame (static_batch.rs:224-267 + generate_static_batch :305-338): as k_sample_fast_rows
    __syncthreads();
    __shared__ uint32_t cur[16];
    if (tid <= n_cb) {
        const uint32_t slow = st->cur[0];
        const bool is_audio = slow >= c.sem_lo;  // :229 (non-audio rows carry zero codes)
        uint32_t v = tid == 0 ? slow : (tid == n_cb ? (uint32_t)code : st->cur[tid]);
        if (tid > 0 && !is_audio) v = 0;
        cur[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const int frame = st->frame;
        const bool frozen = c.session != 0 && st->done != 0 && frame > 0;
        if (!frozen) {
            if (frame == 0 || !st->done) {
                if (frame == 0 && cur[0] < c.sem_lo) st->step = -1;  // (see k_sample_fast_rows)
                const int o = st->n_out;
                uint32_t* oc = out_codes + (size_t)b * n_cb * out_cap;
                if (o < out_cap)
                    for (int cc = 0; cc < n_cb; ++cc) oc[(size_t)cc * out_cap + o] = cur[cc + 1];
                st->n_out = o + 1;
            }
            for (int i = 0; i <= n_cb; ++i) { st->prev[i] = cur[i]; st->cur[i] = cur[i]; }
            st->have_prev = 1;
            st->pos += 1;
            st->frame = frame + 1;
        }
    }
    embed_tokens<WT>(tok_emb, cb_emb, dim, n_cb, cb_size, c.sem_lo, c.sem_hi, cur, 1, X + (size_t)b * dim, tid, PAR_THREADS);
}

// test hook of the block-parallel sampler (lm_bsample_dev.h) with the static-batch RNG derivation of k_sample_slow_rows
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_bsample_rows_test(const float* __restrict__ logits, int n, const SampleCfg* __restrict__ cp,
                                                          const RngState* __restrict__ master, int B, int call, uint32_t* __restrict__ out) {
    __shared__ BSampLds S;
    __shared__ RngState lrng;
    __shared__ uint32_t s_word;
    const int tid = threadIdx.x, b = blockIdx.x;
    const SampleCfg c = *cp;
    float lv[EPT];
#pragma unroll
    for (int s = 0; s < EPT; ++s) { const int i = tid * EPT + s; lv[s] = i < n ? logits[(size_t)b * n + i] : 0.f; }
    if (tid == NT - 1) { child_rng(master, (unsigned long long)call * B + b, &lrng); s_word = chacha12_word(lrng.key, 0); }
    __syncthreads();
    int consumed = 0;
    const int idx = bsample<NT, EPT>(lv, n, c.top_k, (float)(1.0 / (double)c.temp), c.top_p, s_word, &consumed, S, /*batch=*/true, c.top_p64);
    if (tid == 0) out[b] = (uint32_t)idx;
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

// fp8 quantiser: one block per source row.  scale = amax / 448 (1 for an all-zero row); byte = e4m3fn_rne(v / scale) with an
// IEEE-exact f32 division: pure integer / correctly-rounded f32 arithmetic, so any IEEE-754 host reproduces the stored
// bytes and scales bit-for-bit (tests/test_fp8_gpu.py checks that against the CPU restatement).
template <bool SYNTH>
__global__ __launch_bounds__(256) void k_quant_fp8(uint8_t* __restrict__ dst, float* __restrict__ scales, const float* __restrict__ src,
                                                   uint64_t key, long long n_cols, int row_mul, int row_off, float mean, float gscale) {
    __shared__ float red[256];
    const long long r = blockIdx.x;
    auto elem = [&](long long c) -> float {
        if (SYNTH) return synth_elem(key, (uint64_t)(r * n_cols + c), mean, gscale);
        return src[r * n_cols + c];
    };
    float amax = 0.f;
    for (long long c = threadIdx.x; c < n_cols; c += 256) amax = fmaxf(amax, fabsf(elem(c)));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    amax = red[0];
    const float sc = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
    const long long orow = r * row_mul + row_off;
    if (threadIdx.x == 0) scales[orow] = sc;
    for (long long c = threadIdx.x; c < n_cols; c += 256) dst[orow * n_cols + c] = f32_to_e4m3(__fdiv_rn(elem(c), sc));
}

// out[b] = the value the GEMV kernels' unpack path (v_cvt_pk_f32_fp8) assigns to byte b, in each of the 16 lane slots
__global__ void k_fp8_decode_table(float* __restrict__ out) {
    const unsigned b = threadIdx.x & 255u, slot = blockIdx.x;  // slot 0..15: position of the byte inside the 16-byte lane load
    unsigned w[4] = {0u, 0u, 0u, 0u};
    w[slot >> 2] = b << (8 * (slot & 3));
    u32x4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    float f[16];
    WTr<fp8_t>::unpack(v, f);
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i == (int)slot) r = f[i];
    out[slot * 256 + b] = r;
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
    constexpr int WAVES = 2;  // one row per wave, one RoPE pair per block
    const int n_rows = (d.H + 2 * d.Hk) * d.Dh;
    const int grid = (n_rows + WAVES - 1) / WAVES;
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        if (w.cache_resident)
            hipLaunchKernelGGL((k_qkv<WT, K, WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.attn_norm, d.eps,
                               (const WT*)w.wqkv, cos_t, sin_t, state, pos_static, rope_static, q_out, kv, d.H, d.Hk, d.Dh, w.s_qkv);
        else
            hipLaunchKernelGGL((k_qkv<WT, K, WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.attn_norm, d.eps,
                               (const WT*)w.wqkv, cos_t, sin_t, state, pos_static, rope_static, q_out, kv, d.H, d.Hk, d.Dh, w.s_qkv);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
int LmKernels<WT>::attn_chunk() { return AttnGeom<KVT<WT>, 64>::NW * AttnGeom<KVT<WT>, 64>::TW; }  // same for every head_dim

// batch-1 decode over more than 8 chunks: attention blocks take `tpb` consecutive chunks each so that k_wo merges 8 partials
// (FISHRT_ATTN_SUPERCHUNK=0 disables: tuning / test hook)
template <typename WT>
static int attn_tiles_per_block(const ModelDims& d, int nc_launch) {
    static const bool on = [] { const char* e = std::getenv("FISHRT_ATTN_SUPERCHUNK"); return !e || std::atoi(e) != 0; }();
    if (!on || std::is_same<WT, float>::value || d.Dh != 64 || d.n_rep != 8 || nc_launch <= 8) return 1;
    return (nc_launch + 7) / 8;
}
template <typename WT>
void LmKernels<WT>::attn_decode(const ModelDims& d, const float* q, KVView kv, const SeqState* state, float* part,
                                int n_chunks_max, int nc_launch, hipStream_t st) {
    FS_REQUIRE(nc_launch >= 1 && nc_launch <= n_chunks_max, "bad attention chunk count");
    const int grid = d.Hk * nc_launch;
    FS_REQUIRE(n_chunks_max <= 128, "attention supports at most 128 chunks per sequence");
    static const int hs8 = [] { const char* e = std::getenv("FISHRT_ATTN_HSPLIT"); return e ? std::atoi(e) : 4; }();  // tuning hook: 1, 2 or 4
    if (const int tpb = attn_tiles_per_block<WT>(d, nc_launch); tpb > 1) {
        if constexpr (!std::is_same<WT, float>::value)
            hipLaunchKernelGGL((k_attn_rows<KVT<WT>, 64, 2, true>), dim3(d.Hk * 4, 1, (nc_launch + tpb - 1) / tpb), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0,
                               st, q, kv, state, d.Hk, 0, 0, (bf16_t*)nullptr, 4, part, n_chunks_max, tpb);
        FS_LAUNCH_CHECK();
        return;
    }
    if (d.Dh == 64 && d.n_rep == 8 && hs8 == 8 && !std::is_same<WT, float>::value)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 64, 1>), dim3(grid * 8), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 8);
    else if (d.Dh == 64 && d.n_rep == 8 && hs8 == 4 && !std::is_same<WT, float>::value)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 64, 2>), dim3(grid * 4), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 4);
    else if (d.Dh == 64 && d.n_rep == 8 && hs8 == 2 && !std::is_same<WT, float>::value)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 64, 4>), dim3(grid * 2), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 2);
    else if (d.Dh == 64 && d.n_rep == 8)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 64, 8>), dim3(grid), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 1);
    else if (d.Dh == 32 && d.n_rep == 2)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 32, 2>), dim3(grid), dim3(AttnGeom<KVT<WT>, 32>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 1);
    else if (d.Dh == 64 && d.n_rep == 2)
        hipLaunchKernelGGL((k_attn_decode<KVT<WT>, 64, 2>), dim3(grid), dim3(AttnGeom<KVT<WT>, 64>::NW * 64), 0, st, q, kv, state, part, d.Hk, n_chunks_max, nc_launch, 0, 0, 1);
    else
        throw Error("unsupported attention geometry (head_dim, n_rep) = (" + std::to_string(d.Dh) + ", " +
                    std::to_string(d.n_rep) + ")");
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::wo(const ModelDims& d, const float* part, int n_chunks_max, int nc_launch, const SeqState* state, const float* q,
                       KVView kv, int fused_T, const LayerW& w, float* x, hipStream_t st) {
    int chunk = attn_chunk();
    if (fused_T <= 0) {
        FS_REQUIRE(nc_launch >= 1 && nc_launch <= n_chunks_max, "bad attention chunk count");
        const int tpb = attn_tiles_per_block<WT>(d, nc_launch);  // attn_decode left one partial per super-chunk
        chunk *= tpb;
        nc_launch = (nc_launch + tpb - 1) / tpb;
    } else nc_launch = 1;
    constexpr int WAVES = 4;
    const int grid = (d.dim + WAVES - 1) / WAVES;
    FS_REQUIRE(fused_T <= 8 && d.H <= 32 && d.H * 8 * 2 <= WAVES * 64 && d.dim <= 4 * WAVES * 64, "fused attention supports at most 8 cached tokens and 16 heads");
    FS_REQUIRE(d.Dh == 64 || d.Dh == 32, "unsupported head_dim");
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        auto go = [&](auto fused, auto dh) {
            if (w.cache_resident)
                hipLaunchKernelGGL((k_wo<WT, K, WAVES, decltype(fused)::value, decltype(dh)::value, false>), dim3(grid), dim3(WAVES * 64), 0,
                                   st, part, n_chunks_max, chunk, state, q, kv, fused_T, (const WT*)w.wo, x, d.H, d.Hk, d.dim, w.s_o, nc_launch);
            else
                hipLaunchKernelGGL((k_wo<WT, K, WAVES, decltype(fused)::value, decltype(dh)::value, true>), dim3(grid), dim3(WAVES * 64), 0,
                                   st, part, n_chunks_max, chunk, state, q, kv, fused_T, (const WT*)w.wo, x, d.H, d.Hk, d.dim, w.s_o, nc_launch);
        };
        using T = std::true_type; using F = std::false_type;
        using D64 = std::integral_constant<int, 64>; using D32 = std::integral_constant<int, 32>;
        if (fused_T > 0) { if (d.Dh == 64) go(T(), D64()); else go(T(), D32()); }
        else { if (d.Dh == 64) go(F(), D64()); else go(F(), D32()); }
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::ffn_up(const ModelDims& d, const float* x, const LayerW& w, float* act, hipStream_t st) {
    constexpr int WAVES = 4;
    const int grid = (d.inter + WAVES - 1) / WAVES;
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        if (w.cache_resident)
            hipLaunchKernelGGL((k_ffn_up<WT, K, WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.ffn_norm, d.eps,
                               (const WT*)w.w13, act, d.inter, w.s_13);
        else
            hipLaunchKernelGGL((k_ffn_up<WT, K, WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, st, x, w.ffn_norm, d.eps,
                               (const WT*)w.w13, act, d.inter, w.s_13);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::ffn_down(const ModelDims& d, const float* act, const LayerW& w, float* x, hipStream_t st) {
    constexpr int KS = 4;
    dispatch_k(d.inter, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        if (w.cache_resident)
            hipLaunchKernelGGL((k_ffn_down<WT, K, KS, false>), dim3(d.dim), dim3(KS * 64), 0, st, act, (const WT*)w.w2, x, d.dim, w.s_2);
        else
            hipLaunchKernelGGL((k_ffn_down<WT, K, KS, true>), dim3(d.dim), dim3(KS * 64), 0, st, act, (const WT*)w.w2, x, d.dim, w.s_2);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::head(const ModelDims& d, const float* x, const float* norm_w, const void* W, const float* wscale, int n_rows,
                         float* logits, hipStream_t st) {
    constexpr int WAVES = 4;
    const int grid = (n_rows + WAVES - 1) / WAVES;
    dispatch_k(d.dim, [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        hipLaunchKernelGGL((k_head<WT, K, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, x, norm_w, d.eps, (const WT*)W,
                           n_rows, logits, wscale);
    });
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::embed(const ModelDims& d, const void* tok_emb, const void* cb_emb, int n_cb, int cb_size,
                          const SampleCfg* cfg, const uint32_t* prompt, SeqState* state, float* x, hipStream_t st) {
    hipLaunchKernelGGL((k_embed<KVT<WT>>), dim3(1), dim3(256), 0, st, (const KVT<WT>*)tok_emb, (const KVT<WT>*)cb_emb, d.dim, n_cb, cb_size,
                       cfg, prompt, state, x);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void LmKernels<WT>::fast_embed(const ModelDims& d, const void* fast_emb, const uint32_t* ids, int n, float* out,
                               hipStream_t st) {
    hipLaunchKernelGGL((k_fast_embed<KVT<WT>>), dim3(n), dim3(256), 0, st, (const KVT<WT>*)fast_emb, d.dim, ids, out);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void SampleKernels<WT>::sample_slow(const ModelDims& d, const float* logits, int n, const SampleCfg* c, RngState* rng,
                                    SeqState* state, const float* x, float* xf, hipStream_t st, float* const* hid_slot) {
    FS_REQUIRE(n <= SAMPLE_MAXN, "audio-range vocabulary larger than the sampler capacity");
    hipLaunchKernelGGL((k_sample_slow<KVT<WT>>), dim3(1), dim3(SAMPLE_THREADS), 0, st, logits, n, c, rng, state, x, xf, d.dim, hid_slot);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void SampleKernels<WT>::sample_fast(const ModelDims& d, const float* logits, int cb, int n_cb, int cb_size, const SampleCfg* c,
                                    RngState* rng, RepPenState rp, SeqState* state, const void* fast_emb, float* xf,
                                    const void* tok_emb, const void* cb_emb, float* x, uint32_t* out_codes, int out_cap,
                                    hipStream_t st) {
    FS_REQUIRE(cb_size <= SAMPLE_MAXN, "codebook larger than the sampler capacity");
    hipLaunchKernelGGL((k_sample_fast<KVT<WT>>), dim3(1), dim3(SAMPLE_THREADS), 0, st, logits, cb, n_cb, cb_size, c, rng, rp, state,
                       (const KVT<WT>*)fast_emb, xf, (const KVT<WT>*)tok_emb, (const KVT<WT>*)cb_emb, x, d.dim, out_codes, out_cap);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void SampleKernels<WT>::rows_rng_words(const RngState* master, int B, int calls_per_frame, const SeqState* states, uint32_t* words, hipStream_t st) {
    FS_REQUIRE(calls_per_frame <= ROWS_WORDS_LD, "too many sample() calls per frame");
    hipLaunchKernelGGL(k_rows_rng_words, dim3(B), dim3(64), 0, st, master, B, calls_per_frame, states, words);
    FS_LAUNCH_CHECK();
}
bool rows_par_sampler_ok(double temp, uint64_t top_k, int n_slow, int cb_size) {
    return temp > 1e-7 && top_k > 0 && top_k <= (uint64_t)BS_MAXK && (int)top_k < cb_size && (int)top_k < n_slow && n_slow <= 2048 && cb_size <= 1024;
}
template <typename WT>
void SampleKernels<WT>::sample_slow_rows(const ModelDims& d, const float* logits, int ld, int n, const SampleCfg* c, const RngState* master,
                                         int B, int calls_per_frame, SeqState* states, const float* X, float* XF, hipStream_t st, const uint32_t* words,
                                         const float* prep_g, uint16_t* prep_A, uint32_t* epoch) {
    FS_REQUIRE(n <= SAMPLE_MAXN, "audio-range vocabulary larger than the sampler capacity");
    FS_REQUIRE(!prep_g || (d.dim <= 1024 && d.dim % 4 == 0), "sampler-side RMSNorm of the next input row: dim <= 1024");
    const PrepOut po{prep_g, d.eps, prep_A, epoch};
    if (words) {
        hipLaunchKernelGGL((k_sample_slow_rows_par<KVT<WT>>), dim3(B), dim3(PAR_THREADS), 0, st, logits, ld, n, c, words, states, X, XF, d.dim, po);
        FS_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL((k_sample_slow_rows<KVT<WT>>), dim3(B), dim3(SAMPLE_THREADS), 0, st, logits, ld, n, c, master, B, calls_per_frame, states,
                       X, XF, d.dim, po);
    FS_LAUNCH_CHECK();
}
template <typename WT>
void SampleKernels<WT>::sample_fast_rows(const ModelDims& d, const float* logits, int cb, int n_cb, int cb_size, const SampleCfg* c,
                                         const RngState* master, int B, SeqState* states, const void* fast_emb, float* XF,
                                         const void* tok_emb, const void* cb_emb, float* X, uint32_t* out_codes, int out_cap,
                                         hipStream_t st, const uint32_t* words, const float* prep_g, uint16_t* prep_A) {
    FS_REQUIRE(cb_size <= SAMPLE_MAXN, "codebook larger than the sampler capacity");
    FS_REQUIRE(!prep_g || (d.dim <= 1024 && d.dim % 4 == 0), "sampler-side RMSNorm of the next input row: dim <= 1024");
    const PrepOut po{prep_g, d.eps, prep_A, nullptr};
    if (words) {
        hipLaunchKernelGGL((k_sample_fast_rows_par<KVT<WT>>), dim3(B), dim3(PAR_THREADS), 0, st, logits, cb, n_cb, cb_size, c, words, states,
                           reinterpret_cast<const KVT<WT>*>(fast_emb), XF, reinterpret_cast<const KVT<WT>*>(tok_emb), reinterpret_cast<const KVT<WT>*>(cb_emb), X, d.dim,
                           out_codes, out_cap, po);
        FS_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL((k_sample_fast_rows<KVT<WT>>), dim3(B), dim3(SAMPLE_THREADS), 0, st, logits, cb, n_cb, cb_size, c, master, B, states,
                       (const KVT<WT>*)fast_emb, XF, (const KVT<WT>*)tok_emb, (const KVT<WT>*)cb_emb, X, d.dim, out_codes, out_cap, po);
    FS_LAUNCH_CHECK();
}

template <typename WT>
void launch_gather_rows(const void* src, int dim, uint32_t r0, uint32_t r1, void* dst, hipStream_t st) {
    hipLaunchKernelGGL((k_gather_rows<WT>), dim3(1), dim3(256), 0, st, (const WT*)src, dim, r0, r1, (WT*)dst);
    FS_LAUNCH_CHECK();
}
template void launch_gather_rows<bf16_t>(const void*, int, uint32_t, uint32_t, void*, hipStream_t);
template void launch_gather_rows<float>(const void*, int, uint32_t, uint32_t, void*, hipStream_t);
template void launch_gather_rows<fp8_t>(const void*, int, uint32_t, uint32_t, void*, hipStream_t);

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

void launch_synth_quant_fp8(uint8_t* dst, float* scales, uint64_t key, int64_t n_rows, int64_t n_cols, int row_mul, int row_off,
                            float mean, float scale, hipStream_t st) {
    hipLaunchKernelGGL((k_quant_fp8<true>), dim3((unsigned)n_rows), dim3(256), 0, st, dst, scales, (const float*)nullptr, key,
                       (long long)n_cols, row_mul, row_off, mean, scale);
    FS_LAUNCH_CHECK();
}
void launch_fp8_decode_table(float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_fp8_decode_table, dim3(16), dim3(256), 0, st, out);
    FS_LAUNCH_CHECK();
}
void launch_quant_rows_fp8(uint8_t* dst, float* scales, const float* src, int64_t n_rows, int64_t n_cols, int row_mul, int row_off,
                           hipStream_t st) {
    hipLaunchKernelGGL((k_quant_fp8<false>), dim3((unsigned)n_rows), dim3(256), 0, st, dst, scales, src, (uint64_t)0,
                       (long long)n_cols, row_mul, row_off, 0.f, 0.f);
    FS_LAUNCH_CHECK();
}

// ---- MFMA row-path launchers (bf16 and fp8 weights; f32 handles take the sequential decode-kernel path)
template <typename WT>
bool LmKernels<WT>::has_mfma_prefill() { return std::is_same<WT, bf16_t>::value || std::is_same<WT, fp8_t>::value; }

template <typename WT>
void LmKernels<WT>::prefill_embed(const ModelDims& d, const void* tok_emb, const void* cb_emb, int n_cb, int cb_size,
                                  const SampleCfg* cfg, const uint32_t* prompt, const SeqState* state, int M, float* X,
                                  hipStream_t st, int seq_rows, size_t prompt_stride) {
    hipLaunchKernelGGL((k_embed_rows<KVT<WT>>), dim3(M), dim3(256), 0, st, (const KVT<WT>*)tok_emb, (const KVT<WT>*)cb_emb, d.dim, n_cb, cb_size,
                       cfg, prompt, state, X, seq_rows, prompt_stride);
    FS_LAUNCH_CHECK();
}

// Y[M, N] (+)= f(X[M, K]) . W[N, K]^T over `ksplit` K ranges (slabs Y + s * slab_stride for EPI_STORE); rt = 16-row tiles
// per wave (2 halves the L2 re-reads of the activations for the wide W13 GEMM)
template <int EPI>
static void launch_gemm3(int N, int ksplit, int rt, hipStream_t st, const bf16_t* Xf, int M, int K, const void* W, const float* wscale,
                         float* Y, int ldy, size_t slab_stride, bf16_t* Of, int ldo, const float* cos_t, const float* sin_t,
                         const SeqState* state, KVView kv, int H, int Hk, int Dh, RowMap rm, NormAux na = NormAux{}) {
    FS_REQUIRE(K % (ksplit * 128) == 0, "GEMM depth must be a multiple of 128 per K range");
    if (gemm_big_ok(M, N, K, ksplit)) {
        const int tiles = (N + GB_ROWS - 1) / GB_ROWS, panels_b = (M + PF_M - 1) / PF_M;
        const int gzb = std::max(1, std::min(panels_b, 512 / std::max(1, tiles * ksplit)));
        const dim3 gridb(tiles, ksplit, gzb);
        if (wscale)
            hipLaunchKernelGGL((k_gemm_big<EPI, true>), gridb, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, ldo, cos_t, sin_t,
                               state, kv, H, Hk, Dh, rm, na);
        else
            hipLaunchKernelGGL((k_gemm_big<EPI, false>), gridb, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, ldo, cos_t, sin_t,
                               state, kv, H, Hk, Dh, rm, na);
        return;
    }
    const int nks = K / ksplit / 128;
    // enough blocks to fill 256 CUs a few times over: spread the row panels over blockIdx.z until ~1024 blocks
    const int nb = (N + 16 * rt - 1) / (16 * rt) * ksplit, panels = (M + PF_M - 1) / PF_M;
    const int gz = std::max(1, std::min(panels, 1024 / std::max(1, nb)));
    const dim3 grid((N + 16 * rt - 1) / (16 * rt), ksplit, gz);
#define FS_GEMM_CASE(nk, r)                                                                                                          \
    do {                                                                                                                             \
        if (wscale && M <= 16)                                                                                                       \
            hipLaunchKernelGGL((k_gemm3<EPI, nk, r, true, true>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, \
                               ldo, cos_t, sin_t, state, kv, H, Hk, Dh, rm, na);                                                 \
        else if (wscale)                                                                                                             \
            hipLaunchKernelGGL((k_gemm3<EPI, nk, r, true>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, ldo,  \
                               cos_t, sin_t, state, kv, H, Hk, Dh, rm, na);                                                      \
        else if (M <= 16)                                                                                                            \
            hipLaunchKernelGGL((k_gemm3<EPI, nk, r, false, true>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, \
                               ldo, cos_t, sin_t, state, kv, H, Hk, Dh, rm, na);                                                 \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_gemm3<EPI, nk, r, false>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, slab_stride, Of, ldo, \
                               cos_t, sin_t, state, kv, H, Hk, Dh, rm, na);                                                      \
    } while (0)
    if (rt == 4) {
        switch (nks) {
            case 8: FS_GEMM_CASE(8, 4); break;
            case 2: FS_GEMM_CASE(2, 4); break;
            case 1: FS_GEMM_CASE(1, 4); break;
            default: throw Error("unsupported GEMM depth per K range " + std::to_string(K / ksplit) + " (supported: 1024, 256, 128)");
        }
    } else if (rt == 2) {
        switch (nks) {
            case 8: FS_GEMM_CASE(8, 2); break;
            case 2: FS_GEMM_CASE(2, 2); break;
            case 1: FS_GEMM_CASE(1, 2); break;
            default: throw Error("unsupported GEMM depth per K range " + std::to_string(K / ksplit) + " (supported: 1024, 256, 128)");
        }
    } else {
        switch (nks) {
            case 8: FS_GEMM_CASE(8, 1); break;
            case 2: FS_GEMM_CASE(2, 1); break;
            case 1: FS_GEMM_CASE(1, 1); break;
            default: throw Error("unsupported GEMM depth per K range " + std::to_string(K / ksplit) + " (supported: 1024, 256, 128)");
        }
    }
#undef FS_GEMM_CASE
}

// layer-closing down projection of a decode step (k_gemm_down): M <= 32 rows, depth = NKS * 128 * ksplit
static bool gemm_down_ok(int M, int N, int K) { return M <= PF_M && N % 128 == 0 && (K == 4096 || K == 1024 || K == 256); }
size_t rows_xchg_bytes(int dim) { return (size_t)(dim / 16) * 4 * PF_M * 8 * 16; }
static void launch_gemm_down(hipStream_t st, const bf16_t* Xf, int M, int K, const void* W, const float* wscale, int N, float* Y, int ldy, NormAux na,
                             void* xchg, uint32_t* epoch, uint32_t node_id) {
    FS_REQUIRE(xchg && epoch, "folded decode step without an exchange buffer");
    const int nks = K == 4096 ? 8 : (K == 1024 ? 2 : 1), ksplit = K / (nks * 128);   // 4, 4 (mid-size tests), 2 (tiny tests)
    static const int nap = [] { const char* e = std::getenv("FISHRT_DOWN_NAP"); return e ? std::atoi(e) : 8; }();  // 64-clock units before the first sweep (0 / 8 / 16 / 24 / 40: 1820 / 1813 / 1820 / 1853 / 1890 us per B = 32 step)
    const dim3 grid(N / 16, ksplit);
#define FS_DOWN_CASE(nk)                                                                                                              \
    do {                                                                                                                              \
        if (wscale) hipLaunchKernelGGL((k_gemm_down<nk, true>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, na, (u32x4*)xchg, epoch, node_id, nap);  \
        else hipLaunchKernelGGL((k_gemm_down<nk, false>), grid, dim3(256), 0, st, Xf, M, K, W, wscale, N, Y, ldy, na, (u32x4*)xchg, epoch, node_id, nap);        \
    } while (0)
    switch (nks) {
        case 8: FS_DOWN_CASE(8); break;
        case 2: FS_DOWN_CASE(2); break;
        default: FS_DOWN_CASE(1); break;
    }
#undef FS_DOWN_CASE
}
// FISHRT_ROWS_NO_FOLD=1: A/B hook -- decode steps keep the split-K slabs + k_prep nodes
static bool rows_fold_enabled() {
    static const bool v = [] { const char* e = std::getenv("FISHRT_ROWS_NO_FOLD"); return !(e && std::atoi(e) != 0); }();
    return v;
}
// k_gemm_down's owner waves spin on units published by the other K-quarter blocks of the SAME launch: every block of the grid must be
// resident at once.  The grid is dispatched y = 0 first; on a device (or partition: CPX, a masked-off part) where
// occupancy x CUs < N / 16 x ksplit the owners of resident tiles would wait for blocks that cannot be scheduled.  Checked once per
// (device, instantiation) with the runtime's occupancy query; such devices keep the slab + k_prep step.
template <int NKS, bool FP8>
static bool gemm_down_resident_inst(int blocks) {
    static int cached_dev = -1, cached_cap = 0;
    int dev = 0;
    FS_HIP(hipGetDevice(&dev));
    if (dev != cached_dev) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        FS_HIP(hipGetDeviceProperties(&prop, dev));
        FS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_gemm_down<NKS, FP8>), 256, 0));
        cached_cap = per_cu * prop.multiProcessorCount;
        cached_dev = dev;
    }
    if (const char* e = std::getenv("FISHRT_DOWN_FAKE_CAPACITY")) return std::atoi(e) >= blocks;  // (test hook: pretend a smaller device)
    return cached_cap >= blocks;
}
static bool gemm_down_resident(int N, int K, bool fp8) {
    const int nks = K == 4096 ? 8 : (K == 1024 ? 2 : 1), ksplit = K / (nks * 128), blocks = (N / 16) * ksplit;
    switch (nks) {
        case 8: return fp8 ? gemm_down_resident_inst<8, true>(blocks) : gemm_down_resident_inst<8, false>(blocks);
        case 2: return fp8 ? gemm_down_resident_inst<2, true>(blocks) : gemm_down_resident_inst<2, false>(blocks);
        default: return fp8 ? gemm_down_resident_inst<1, true>(blocks) : gemm_down_resident_inst<1, false>(blocks);
    }
}
template <typename WT>
bool LmKernels<WT>::rows_fold_ok(const ModelDims& d, int M, const RowsCtx& c) {
    if constexpr (std::is_same<WT, float>::value) return false;
    else
        return rows_fold_enabled() && c.A2 != nullptr && c.ss != nullptr && c.xchg != nullptr && c.epoch != nullptr && gemm_down_ok(M, d.dim, d.inter) &&
               !gemm_big_ok(M, d.dim, d.dim, 1) && c.stage_mask == 0xFFu && gemm_down_resident(d.dim, d.inter, std::is_same<WT, fp8_t>::value);
}

template <typename WT>
void LmKernels<WT>::rows_layer(const ModelDims& d, int M, const RowsCtx& c, const LayerW& w, KVView kv, bool first, hipStream_t st, const float* next_norm) {
    if constexpr (std::is_same<WT, float>::value) {
        throw Error("the MFMA row path needs bf16 or fp8 weights");
    } else {
        using KT = KVT<WT>;  // bf16 KV cache for both weight types
        FS_REQUIRE(M >= 1 && M <= c.Mcap, "more activation rows than the row buffers hold");
        FS_REQUIRE(d.dim % 128 == 0 && d.inter % 128 == 0, "row path needs dim % 128 == 0 and intermediate_size % 128 == 0");
        const int qkv_rows = (d.H + 2 * d.Hk) * d.Dh;
        // 16-row-tiles per wave: 1 everywhere, 2 for the wide W13 (halves the L2 re-reads of the activation fragments).  64-row
        // blocks (rt = 4) were measured for prefill-sized passes and are SLOWER (W13 at 384 rows: 33 -> 41 us): the lost
        // occupancy costs more than the saved fragment traffic.
        static const int rt13_env = [] { const char* e = std::getenv("FISHRT_RT13"); return e ? std::atoi(e) : 0; }();  // experiment hook (1, 2 or 4)
        const int rt_qkv = 1, rt_o = 1, rt_13 = (rt13_env == 1 || rt13_env == 2 || rt13_env == 4) ? rt13_env : 2, rt_2 = 1;
        const int nblk_o = gemm_big_ok(M, d.dim, d.dim, 1) ? d.dim / GB_ROWS : d.dim / (16 * rt_o);  // blocks of the Wo GEMM (sum-of-squares partials)
        const RowMap rm{c.pos_step, c.pt_stride, c.seq_rows};
        const RowMap none{0, 0, 0};
        KVView nokv = {};
        const size_t slab = (size_t)c.Mcap * d.dim;
        const int DOWN_SPLIT = c.down_split;
        // decode steps (c.fold): the previous layer's un-split down projection (k_gemm_down) closed the residual stream and left
        // split(x * attention_norm) + sum-of-squares partials in c.A / c.ss: no k_prep node, Wqkv divides by the row's rms instead
        const bool fold = c.fold && next_norm != nullptr;
        if (fold) FS_REQUIRE(rows_fold_ok(d, M, c), "folded decode step on a shape the un-split down projection does not take");
        const int nblk_d = d.dim / 16;
        // (1) x += previous layer's down-proj slabs ; RMSNorm(attention_norm) -> hi/lo
        // c.qkv0_tbl (fold, small_attn, first layer of a codebook pass >= 1): q / k / v of the new token are a row of the qkv table -- no k_prep, no Wqkv
        // GEMM; the attention node reads the table itself (k_attn_small_rows_tbl)
        const bool tbl0 = fold && first && c.qkv0_tbl != nullptr && c.small_attn && !c.attn_t1 && d.Dh == 64 && d.H <= 32 && d.Hk <= 4 && (c.stage_mask & 4u);
        if (tbl0) {}
        else if (fold && (!first || c.first_prepped)) {}   // (first_prepped: the sampler that wrote the input row left its normalised fragments in c.A)
        else if (c.stage_mask & 1u) hipLaunchKernelGGL(k_prep, dim3(M), dim3(256), 0, st, c.X, d.dim, c.P, first ? 0 : DOWN_SPLIT, slab, w.attn_norm, d.eps, c.A);
        // (2) Wqkv + rope + KV scatter
        // c.attn_t1 (fold, small_attn; every row at position 0 of an empty cache): the Wqkv epilogue leaves the attention output itself
        // (GemmEpi::o1) in c.C -- free until this layer's W13 -- and the attention node is not launched
        const bool t1 = fold && c.attn_t1 && c.small_attn && !gemm_big_ok(M, qkv_rows, d.dim, 1);
        bf16_t* const o1 = t1 ? c.C : nullptr;
        // (Measured and rejected, round 5: Wqkv + the row attention of the later passes in ONE launch -- the GEMM blocks publish q / k / v as
        // tagged units, block m then attends for row m, bit-identical results: 10.1-10.4 us per node against 5.7 + 5.6 for the two nodes under
        // rocprofv3, 1856-1877 vs 1825-1833 us per step -- a consumer that waits for units from all 80 blocks on all 8 XCDs pays the slowest
        // block's finish + cross-XCD visibility, unlike k_gemm_down's four same-XCD partners; profiles/r05_rows_fold.txt)
        if (tbl0) {}
        else if (fold && !first)
            launch_gemm3<EPI_QKV_RMS>(qkv_rows, 1, rt_qkv, st, c.A, M, d.dim, w.wqkv, w.s_qkv, c.Q, d.dim, 0, o1, 0,
                                      c.cos_t, c.sin_t, c.state, kv, d.H, d.Hk, d.Dh, rm, NormAux{nullptr, c.ss, nullptr, nblk_d, d.dim, d.eps});
        else if (c.stage_mask & 2u) launch_gemm3<EPI_QKV>(qkv_rows, 1, rt_qkv, st, c.A, M, d.dim, w.wqkv, w.s_qkv, c.Q, d.dim, 0, o1, 0,
                              c.cos_t, c.sin_t, c.state, kv, d.H, d.Hk, d.Dh, rm);
        // (3) attention over each row's KV prefix + chunk combine -> hi/lo
        if (t1) {}
        else if (tbl0)
            hipLaunchKernelGGL((k_attn_small_rows_tbl<64>), dim3(M), dim3(256), 0, st, c.qkv0_tbl, c.row_states, c.code_slot, kv, c.state, d.H, d.Hk, c.pos_step,
                               c.pt_stride, c.cos_t, c.sin_t, c.A, (int)c.identity_pages);
        else if (c.seq_rows > 0) {
            // group prefill: M = n_seq * seq_rows rows, every sequence starts at state->pos; flash attention per sequence
            FS_REQUIRE(d.Dh == 64 && M % c.seq_rows == 0 && !c.no_flash, "group prefill needs head_dim 64 and whole sequences");
            if (c.stage_mask & 4u)
                hipLaunchKernelGGL(k_attn_prefill_mfma, dim3(d.H * ((c.seq_rows + 15) / 16), M / c.seq_rows), dim3(256), 0, st, c.Q, kv, c.state,
                                   c.seq_rows, d.H, d.Hk, c.A, c.pt_stride);
        } else if (c.pos_step == 1 && c.pt_stride == 0 && (c.stage_mask & 4u) && d.Dh == 64 && M > 1 && !c.no_flash) {
            // prefill: causal flash attention on the matrix cores, result straight into the Wo GEMM's input
            hipLaunchKernelGGL(k_attn_prefill_mfma, dim3(d.H * ((M + 15) / 16)), dim3(256), 0, st, c.Q, kv, c.state, M, d.H, d.Hk, c.A, 0);
        } else if (c.small_attn && (c.stage_mask & 4u) && d.H <= 32 && (d.Dh == 64 || d.Dh == 32)) {
            // fast decoder: <= 8 tokens in one page -> one node instead of two
            if (d.Dh == 64)
                hipLaunchKernelGGL((k_attn_small_rows<64>), dim3(M), dim3(256), 0, st, c.Q, kv, c.state, d.H, d.Hk, c.pos_step, c.pt_stride, c.A, (int)c.identity_pages);
            else
                hipLaunchKernelGGL((k_attn_small_rows<32>), dim3(M), dim3(256), 0, st, c.Q, kv, c.state, d.H, d.Hk, c.pos_step, c.pt_stride, c.A, (int)c.identity_pages);
        } else if (c.pos_step <= 0 && !c.chunked_attn && ((d.Dh == 64 && (d.n_rep == 8 || d.n_rep == 2)) || (d.Dh == 32 && d.n_rep == 2))) {
            // static-batch decode: one fused node per layer (whole KV prefix per (kv head, row) block)
            // few rows: split the 8 query heads of a kv group over two blocks (64 -> 128 blocks at 32 rows)
            const int hs = (d.Dh == 64 && d.n_rep == 8) ? (d.Hk * M <= 64 ? 4 : (d.Hk * M < 256 ? 2 : 1)) : 1;
            const dim3 gr(d.Hk * hs, M);
            if (!(c.stage_mask & 4u)) {}
            else if (d.Dh == 64 && d.n_rep == 8 && hs == 4)
                hipLaunchKernelGGL((k_attn_rows<KT, 64, 2>), gr, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, d.Hk, c.pos_step, c.pt_stride, c.A, 4);
            else if (d.Dh == 64 && d.n_rep == 8 && hs == 2)
                hipLaunchKernelGGL((k_attn_rows<KT, 64, 4>), gr, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, d.Hk, c.pos_step, c.pt_stride, c.A, 2);
            else if (d.Dh == 64 && d.n_rep == 8)
                hipLaunchKernelGGL((k_attn_rows<KT, 64, 8>), gr, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, d.Hk, c.pos_step, c.pt_stride, c.A, 1);
            else if (d.Dh == 64)
                hipLaunchKernelGGL((k_attn_rows<KT, 64, 2>), gr, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, d.Hk, c.pos_step, c.pt_stride, c.A, 1);
            else
                hipLaunchKernelGGL((k_attn_rows<KT, 32, 2>), gr, dim3(AttnGeom<KT, 32>::NW * 64), 0, st, c.Q, kv, c.state, d.Hk, c.pos_step, c.pt_stride, c.A, 1);
        } else {
        FS_REQUIRE(M <= c.part_rows, "more rows than the attention-partials buffer holds");
        const dim3 ga(d.Hk * c.nc_launch, M);
            if (!(c.stage_mask & 4u)) {}
            else if (d.Dh == 64 && d.n_rep == 8)
                hipLaunchKernelGGL((k_attn_decode<KT, 64, 8>), ga, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, c.part, d.Hk, c.n_chunks_max, c.nc_launch, c.pos_step, c.pt_stride, 1);
            else if (d.Dh == 32 && d.n_rep == 2)
                hipLaunchKernelGGL((k_attn_decode<KT, 32, 2>), ga, dim3(AttnGeom<KT, 32>::NW * 64), 0, st, c.Q, kv, c.state, c.part, d.Hk, c.n_chunks_max, c.nc_launch, c.pos_step, c.pt_stride, 1);
            else if (d.Dh == 64 && d.n_rep == 2)
                hipLaunchKernelGGL((k_attn_decode<KT, 64, 2>), ga, dim3(AttnGeom<KT, 64>::NW * 64), 0, st, c.Q, kv, c.state, c.part, d.Hk, c.n_chunks_max, c.nc_launch, c.pos_step, c.pt_stride, 1);
            else
                throw Error("unsupported attention geometry");
            if (!(c.stage_mask & 8u)) {}
            else if (d.Dh == 64)
                hipLaunchKernelGGL((k_attn_combine<64>), dim3(M), dim3(256), 0, st, c.part, c.n_chunks_max, attn_chunk(), c.state, c.pos_step, c.A, d.H);
            else
                hipLaunchKernelGGL((k_attn_combine<32>), dim3(M), dim3(256), 0, st, c.part, c.n_chunks_max, attn_chunk(), c.state, c.pos_step, c.A, d.H);
        }
        // (4) Wo + residual (each output element owned by one lane: deterministic)
        const bool fuse_norm = c.A2 != nullptr && c.ss != nullptr && nblk_o % 8 == 0 && d.dim % (16 * rt_o) == 0;
        NormAux na{w.ffn_norm, c.ss, c.A2, nblk_o, d.dim, d.eps};
        if (!(c.stage_mask & 16u)) {}
        else if (fuse_norm)
            launch_gemm3<EPI_RESIDUAL_NORM>(d.dim, 1, rt_o, st, t1 ? c.C : c.A, M, d.dim, w.wo, w.s_o, c.X, d.dim, 0, nullptr, 0, nullptr, nullptr,
                                            nullptr, nokv, 0, 0, 0, none, na);
        else launch_gemm3<EPI_RESIDUAL>(d.dim, 1, rt_o, st, c.A, M, d.dim, w.wo, w.s_o, c.X, d.dim, 0, nullptr, 0,
                                   nullptr, nullptr, nullptr, nokv, 0, 0, 0, none);
        // (5) RMSNorm(ffn_norm) -> hi/lo ; W1||W3 + SwiGLU -> act hi/lo ; W2 split-K slabs (summed by the next k_prep)
        if (fuse_norm) {
            if (c.stage_mask & 64u) launch_gemm3<EPI_SWIGLU_RMS>(2 * d.inter, 1, rt_13, st, c.A2, M, d.dim, w.w13, w.s_13, nullptr, 0, 0, c.C, d.inter,
                                     nullptr, nullptr, nullptr, nokv, 0, 0, 0, none, na);
        } else {
            if (c.stage_mask & 32u) hipLaunchKernelGGL(k_prep, dim3(M), dim3(256), 0, st, c.X, d.dim, (const float*)nullptr, 0, slab, w.ffn_norm, d.eps, c.A);
            if (c.stage_mask & 64u) launch_gemm3<EPI_SWIGLU>(2 * d.inter, 1, rt_13, st, c.A, M, d.dim, w.w13, w.s_13, nullptr, 0, 0, c.C, d.inter,
                                     nullptr, nullptr, nullptr, nokv, 0, 0, 0, none);
        }
        if (fold)  // x += W2 . act ; split(x * next_norm) -> c.A ; sum-of-squares partials -> c.ss  (the attention output in c.A and Wo's partials are consumed)
            launch_gemm_down(st, c.C, M, d.inter, w.w2, w.s_2, d.dim, c.X, d.dim, NormAux{next_norm, c.ss, c.A, nblk_d, d.dim, d.eps}, c.xchg, c.epoch, c.node_id);
        else if (c.stage_mask & 128u) launch_gemm3<EPI_STORE>(d.dim, DOWN_SPLIT, rt_2, st, c.C, M, d.inter, w.w2, w.s_2, c.P, d.dim, slab, nullptr, 0, nullptr, nullptr,
                                nullptr, nokv, 0, 0, 0, none);
        FS_LAUNCH_CHECK();
    }
}

// (kept for API stability: the row kernels need no one-time setup any more)
template <typename WT>
void LmKernels<WT>::rows_warmup() {}

// x += last layer's down-proj slabs (closes a rows_layer chain); optionally RMSNorm -> hi/lo for a head GEMM
template <typename WT>
void LmKernels<WT>::rows_finish(const ModelDims& d, int M, const RowsCtx& c, const float* norm_w, hipStream_t st) {
    hipLaunchKernelGGL(k_prep, dim3(M), dim3(256), 0, st, c.X, d.dim, c.P, c.down_split, (size_t)c.Mcap * d.dim, norm_w, d.eps,
                       norm_w ? c.A : (bf16_t*)nullptr);
    FS_LAUNCH_CHECK();
}

// head GEMM of the MFMA row path: logits[m][0..n_rows) = W[n_rows, dim] . (hi + lo)[m]  (input = rows_finish(norm_w) output)
template <typename WT>
void LmKernels<WT>::rows_head(const ModelDims& d, int M, const RowsCtx& c, const void* W, const float* wscale, int n_rows, float* logits, int ld,
                              hipStream_t st, bool rms) {
    if constexpr (std::is_same<WT, float>::value) {
        throw Error("the MFMA row path needs bf16 or fp8 weights");
    } else {
        FS_REQUIRE(ld >= n_rows, "logits row stride must cover n_rows");
        KVView nokv = {};
        // EPI_STORE with one K range writes slab 0 == the logits matrix itself (row stride ld)
        if (rms)  // folded decode step: c.A holds split(x * norm_w), c.ss the sum-of-squares partials of the last down projection
            launch_gemm3<EPI_STORE_RMS>(n_rows, 1, 1, st, c.A, M, d.dim, W, wscale, logits, ld, 0, nullptr, 0, nullptr, nullptr, nullptr, nokv, 0, 0,
                                        0, RowMap{0, 0}, NormAux{nullptr, c.ss, nullptr, d.dim / 16, d.dim, d.eps});
        else
        launch_gemm3<EPI_STORE>(n_rows, 1, 1, st, c.A, M, d.dim, W, wscale, logits, ld, 0, nullptr, 0, nullptr, nullptr, nullptr, nokv, 0, 0,
                                0, RowMap{0, 0});
        FS_LAUNCH_CHECK();
    }
}

// ---- decision capture on the row path (fs_lm_debug_capture for generate_static_batch / sessions): what every row's decision saw and picked,
// in the layout of the request-row kernels' record: cap[row][cap_frames][9][2048], logits at [0, n), the pick at [2047] (slow) / [1024] (fast)
__global__ __launch_bounds__(256) void k_cap_rows_logits(const float* __restrict__ logits, int ld, int n, const SeqState* __restrict__ states,
                                                         const SampleCfg* __restrict__ cp, float* __restrict__ cap, int cap_frames, int decision) {
    const int b = blockIdx.x, frame = states[b].frame;
    if (frame >= cap_frames) return;
    float* dst = cap + (((size_t)b * cap_frames + frame) * 9 + decision) * 2048;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = logits[(size_t)b * ld + i];
        if (decision == 0 && i == 0 && cp->ignore_eos) v = -INFINITY;  // (what the slow samplers do to <|im_end|> before they select)
        dst[i] = v;
    }
}
__global__ __launch_bounds__(64) void k_cap_rows_picks(const SeqState* __restrict__ states, const SampleCfg* __restrict__ cp, float* __restrict__ cap,
                                                       int cap_frames, int n_cb) {
    const int b = blockIdx.x, frame = states[b].frame - 1, t = threadIdx.x;  // (the last sampler of the frame advanced the counter)
    if (frame < 0 || frame >= cap_frames || t > n_cb) return;
    float* dst = cap + (((size_t)b * cap_frames + frame) * 9 + t) * 2048;
    const uint32_t v = states[b].cur[t];
    if (t == 0) dst[2047] = v == cp->im_end_id ? 0.f : (float)(v - cp->audio_base);
    else dst[1024] = (float)v;
}
void launch_cap_rows_logits(const float* logits, int ld, int n, const SeqState* states, const SampleCfg* cfg, int B, float* cap, int cap_frames, int decision,
                            hipStream_t st) {
    hipLaunchKernelGGL(k_cap_rows_logits, dim3(B), dim3(256), 0, st, logits, ld, n, states, cfg, cap, cap_frames, decision);
}
void launch_cap_rows_picks(const SeqState* states, const SampleCfg* cfg, int B, float* cap, int cap_frames, int n_cb, hipStream_t st) {
    hipLaunchKernelGGL(k_cap_rows_picks, dim3(B), dim3(64), 0, st, states, cfg, cap, cap_frames, n_cb);
}

void launch_advance_n(SeqState* state, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_advance_n, dim3(1), dim3(1), 0, st, state, n);
    FS_LAUNCH_CHECK();
}

// ---- sampler test hook (fs_selftest_sample_rows): the static-batch slow sampler on caller-provided logits, B rows of n candidates,
// as sample() call number `call_index` of a request (child StdRng of row b = master u64 number call_index * B + b)
void debug_sample_rows(int device, const float* logits, int B, int n, double temp, double top_p, uint64_t top_k, uint64_t seed,
                       int call_index, uint32_t* out) {
    FS_REQUIRE(B >= 1 && n >= 1 && n <= SAMPLE_MAXN, "bad sampler test shape");
    FS_HIP(hipSetDevice(device));
    float* d_logits = nullptr; SampleCfg* d_cfg = nullptr; RngState* d_rng = nullptr; SeqState* d_st = nullptr;
    FS_HIP(hipMalloc(&d_logits, sizeof(float) * (size_t)B * n));
    FS_HIP(hipMalloc(&d_cfg, sizeof(SampleCfg))); FS_HIP(hipMalloc(&d_rng, sizeof(RngState))); FS_HIP(hipMalloc(&d_st, sizeof(SeqState) * B));
    FS_HIP(hipMemcpy(d_logits, logits, sizeof(float) * (size_t)B * n, hipMemcpyHostToDevice));
    SampleCfg c = {};
    c.temp = (float)temp; c.top_p = (float)top_p; c.top_k = (int)std::min<uint64_t>(top_k, 1u << 30); c.rep_pen = 1.f; c.top_p64 = top_p;
    FS_HIP(hipMemcpy(d_cfg, &c, sizeof(c), hipMemcpyHostToDevice));
    RngState r = {};
    unsigned long long state = seed;
    for (int i = 0; i < 8; ++i) {  // rand_core seed_from_u64 (PCG32 expansion)
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        r.key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    FS_HIP(hipMemcpy(d_rng, &r, sizeof(r), hipMemcpyHostToDevice));
    std::vector<SeqState> hs(B);
    for (auto& s : hs) { s = SeqState{}; s.frame = call_index; }
    FS_HIP(hipMemcpy(d_st, hs.data(), sizeof(SeqState) * B, hipMemcpyHostToDevice));
    // FISHRT_SAMPLER_IMPL=par512: the block-parallel sampler the persistent fast decoder uses (lm_bsample_dev.h), 512 threads per row
    const char* impl = getenv("FISHRT_SAMPLER_IMPL");
    const bool par = impl && std::string(impl) == "par512" && temp > 1e-7 && top_k > 0 && top_k <= 256 && (int)top_k < n && n <= 2048;
    if (par) {
        uint32_t* d_out = nullptr;
        FS_HIP(hipMalloc(&d_out, sizeof(uint32_t) * B));
        if (n <= 1024) hipLaunchKernelGGL((k_bsample_rows_test<512, 2>), dim3(B), dim3(512), 0, nullptr, d_logits, n, d_cfg, d_rng, B, call_index, d_out);
        else hipLaunchKernelGGL((k_bsample_rows_test<512, 4>), dim3(B), dim3(512), 0, nullptr, d_logits, n, d_cfg, d_rng, B, call_index, d_out);
        FS_LAUNCH_CHECK();
        FS_HIP(hipDeviceSynchronize());
        FS_HIP(hipMemcpy(out, d_out, sizeof(uint32_t) * B, hipMemcpyDeviceToHost));
        (void)hipFree(d_out);
    } else {
    hipLaunchKernelGGL((k_sample_slow_rows<bf16_t>), dim3(B), dim3(SAMPLE_THREADS), 0, nullptr, d_logits, n, n, d_cfg, d_rng, B, 1, d_st,
                       (const float*)nullptr, (float*)nullptr, 0, PrepOut{nullptr, 0.f, nullptr, nullptr});
    FS_LAUNCH_CHECK();
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMemcpy(hs.data(), d_st, sizeof(SeqState) * B, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) out[b] = hs[b].cur[0];
    }
    (void)hipFree(d_logits); (void)hipFree(d_cfg); (void)hipFree(d_rng); (void)hipFree(d_st);
}

template struct LmKernels<bf16_t>;
template struct LmKernels<float>;
template struct SampleKernels<bf16_t>;
template struct SampleKernels<float>;
template struct LmKernels<fp8_t>;
template struct SampleKernels<fp8_t>;
template void launch_synth_fill<bf16_t>(bf16_t*, uint64_t, int64_t, int64_t, int, int, float, float, int, hipStream_t);
template void launch_synth_fill<float>(float*, uint64_t, int64_t, int64_t, int, int, float, float, int, hipStream_t);
template void launch_convert_rows<bf16_t>(bf16_t*, const float*, int64_t, int64_t, int, int, hipStream_t);
template void launch_convert_rows<float>(float*, const float*, int64_t, int64_t, int, int, hipStream_t);

}  // namespace fs
