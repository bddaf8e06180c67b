// Persistent slow-transformer kernel: one launch = DualARTransformer::forward_generate for ONE new token at batch 1 (24 blocks over
// the paged KV cache + final norm + audio-range head), 256 workgroups x 512 threads, every workgroup co-resident.
//
// Reference semantics (same arithmetic as the per-node kernels of lm_kernels.hip, other summation order):
//   forward_generate   fish_speech_core/lib/lm/dual_ar.rs:574-635 (L == 1: no mask, :360)
//   Attention::forward dual_ar.rs:281-384 (interleaved RoPE on q / k, scale on K, K/V appended to the cache, GQA by head index)
//   FeedForward        dual_ar.rs:160-165;  TransformerBlock :429-440;  constrain_probs_to_audio generate/utils.rs:13-16
//
// Unlike the fast decoder (lm_persist.hip) the 717 MB of slow weights cannot stay on chip: every stage's weight slice (its
// workgroup's rows, 8-64 KB per CU) is requested into VGPRs ONE STAGE AHEAD, right after the current stage's sweep has completed, so
// the HBM stream of stage s+1 runs under stage s's arithmetic, publish and the propagation of its edge; a CU's loads return in issue
// order, so the sweep of stage s+1 simply completes when both the weights and the edge are there.
//
// Stages per block (edges are all-gathers of 8-byte {value, tag} granules, see lm_persist_dev.h):
//   S1  (gather x) -> RMSNorm folded -> Wqkv rows [5b, 5b+5)                                  -> 1280 granules
//   S2  attention: workgroup (h, s) = (b / n_sl, b % n_sl), b < 16 n_sl, owns query head h over token slice s of the cache (K/V tile
//       prefetched during S1; the new token's k / v come from the qkv edge and are appended to the cache by the workgroup of the
//       last slice of heads 0 and 8) -> un-normalised partial {o[64], m, l} per (head, slice)   -> 16 n_sl 66 granules
//   S3  every workgroup merges the slices of all heads (flash-decoding combine), Wo rows [4b, 4b+4) + residual -> 1024 granules
//   S4  gather h -> RMSNorm folded -> 16 SwiGLU pairs of W13                                   -> 4096 granules
//   S5  gather the activations -> W2 rows [4b, 4b+4) + residual                                 -> 1024 granules
//   head: gather x (= the pre-norm hidden state, also written to A.x) -> norm folded -> head rows [8b, 8b+8) -> A.logits (plain stores)
//
// Round 3 additions: k_slow_persist<true> streams e4m3 byte images (half the bytes; a dword holds the pairs of two
// rows, v_cvt_pk_f32_fp8, per-row scales applied by the publishing lanes); rsq / rcp epilogues; every stage sleeps A.naps[kind] x 64 clocks
// before its first sweep (pf_nap_before_sweep).  Measured and rejected: an LDS-ring loader wave for the weight stream, fewer attention
// slices, retry naps inside the per-lane spin loop (profiles/r03_loader_engine.txt, r03_poll_naps.txt).
#include "lm_persist.h"

#include <hip/hip_runtime.h>

#include <type_traits>
#include <vector>

#include "fs_common.h"

namespace fs {

namespace {

#include "lm_persist_dev.h"
#include "lm_persist_rows_dev.h"  // pr_sweep_att8: 16 x 16-byte sweep loads of one lane in flight (uniform bases in SGPRs)

constexpr int PS_DROR8 = 0x128;  // DPP row_ror:8 -- lane l <- lane l ^ 8 inside a row of 16

// byte offsets inside a (layer, workgroup) weight image; every region is [chunk][512 lanes] x 16 B (B: x 4 B)
constexpr size_t IM_QKV4 = 0, IM_QKV1 = 8192, IM_WO = 10240, IM_W13 = 18432, IM_W2 = 83968;
static_assert(IM_W2 + 4 * 8192 == PS_LAYER_IMAGE, "image layout");
// FS_FP8 images: one e4m3 byte per weight, so a lane's (2 t, 2 t + 1) pair of a row is 16 bits and a dword holds the pairs of TWO rows
// (low half: the even row).  Regions: Wqkv rows 0..3 [512] x 8 B, row 4 [512] x 4 B, Wo [512] x 8 B, W13 [4][512] x 16 B (chunk c = rows
// 8 c .. 8 c + 7), W2 [2][512] x 16 B (chunk h = K quarters 2 h, 2 h + 1; per quarter rows 0..3 in two dwords).  The per-row f32
// scales (45 per layer and workgroup, padded to 48: Wqkv 5 at [0], Wo 4 at [8], W13 32 at [12], W2 4 at [44]) travel separately and
// multiply the K-summed row result in the publishing lanes.
constexpr size_t I8_QKV4 = 0, I8_QKV1 = 4096, I8_WO = 6144, I8_W13 = 10240, I8_W2 = 43008;
static_assert(I8_W2 + 2 * 8192 == PS_LAYER_IMAGE_FP8, "fp8 image layout");
constexpr int PS_SC = 48, SC_QKV = 0, SC_WO = 8, SC_W13 = 12, SC_W2 = 44;

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 ps_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float ps_f32x4_t __attribute__((ext_vector_type(4)));
// FISHRT_SLOW_S4_VALU=1: A/B hook -- the bf16 image keeps the row-pair layout of W13 and S4 runs on the VALU (the round-4 stage)
static bool slow_s4_mfma() {
    static const bool v = [] { const char* e = std::getenv("FISHRT_SLOW_S4_VALU"); return !(e && std::atoi(e) != 0); }();
    return v;
}
// two rows' (2 t, 2 t + 1) pairs in one dword: acc_even += row_even . c, acc_odd += row_odd . c
__device__ __forceinline__ void pf_dot2x2_fp8(uint32_t w, float c0, float c1, float& acc_even, float& acc_odd) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), b2 = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    acc_even = fmaf(a.x, c0, acc_even); acc_even = fmaf(a.y, c1, acc_even);
    acc_odd = fmaf(b2.x, c0, acc_odd); acc_odd = fmaf(b2.y, c1, acc_odd);
}

// LDS carve
constexpr int S_XS = 0;                       // residual stream copy f32 [1024]
constexpr int S_RED = S_XS + 4096;            // [2][8][PS_RED] row partials
constexpr int PS_RED = 40;
constexpr int S_QS = S_RED + 2 * 8 * PS_RED * 4;   // rope'd, pre-scaled q of this workgroup's head, f32 [64]
constexpr int S_KN = S_QS + 256;              // new token's k (rope'd, bf16-rounded) f32 [64]
constexpr int S_VN = S_KN + 256;              // new token's v (bf16-rounded) f32 [64]
constexpr int S_PART = S_VN + 256;            // [8 waves][72]: o[64], l
constexpr int S_WMAX = S_PART + 8 * 72 * 4;   // [8] wave maxima + [1] block max
constexpr int S_PAGES = S_WMAX + 64;          // page ids of this workgroup's token slice, int [160]
constexpr int S_XB = S_PAGES + 160 * 4;       // S4 on the matrix cores: x . g split into three bf16 arrays [3][1024] + one 16-byte zero slot per wave
constexpr int S_END = S_XB + 3 * 2048 + 8 * 16;
constexpr int PS_LDS = 96 * 1024;             // requested size: > half of the CU's LDS, so the 256 workgroups sit on 256 different CUs
static_assert(S_END <= PS_LDS, "LDS budget");

// up to 8 units (16 B = 2 granules each) per lane, all in flight; units >= n are not loaded
__device__ __forceinline__ void pf_sweep8u(const u64* base, const int (&unit)[8], int n, unsigned tag, u32x4 (&v)[8], bool& dead, uint32_t* ctl) {
    const u64* p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = base + 2 * (size_t)unit[i < n ? i : 0];
    for (unsigned spins = 0;; ++spins) {
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
            "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
            : "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok &= pf_tags_ok(v[i], tag);
        if (ok || dead) return;
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }
    }
}

// A 16-byte load the compiler's s_waitcnt bookkeeping does not see: dst is valid after the next sweep (every sweep statement ends in
// s_waitcnt vmcnt(0)).  Used where the ISSUE POINT of a prefetch depends on the workgroup's role: as ordinary loads in two branches they
// made every later wait for an older load a vmcnt(0) (the pass cannot count loads of a branch not taken) and cost what the early issue gained.
__device__ __forceinline__ void ps_load16_unseen(u32x4& dst, const unsigned char* base, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base));
}

}  // namespace

// ------------------------------------------------------------------------------------------------ weight images
// w13_frag: the W13 region holds the workgroup's 32 rows as MFMA A fragments (v_mfma_f32_16x16x32_bf16) -- chunk u: row tile u / 4 (rows
// 32 b + 16 (u / 4) + (lane & 15)), k-step u % 4 of the wave's K slice: elements 128 wave + 32 (u % 4) + 8 (lane >> 4) .. + 8 (the layout of
// the fast decoder's resident W13, lm_persist.hip) -- instead of row pairs [chunk c: rows 4 c .. 4 c + 3][lane t: elements 2 t, 2 t + 1]
__global__ __launch_bounds__(PF_THREADS) void k_ps_pack_layer(LayerW w, unsigned char* __restrict__ image /*[PF_BLOCKS][PS_LAYER_IMAGE]*/, int w13_frag) {
    const int b = blockIdx.x, t = threadIdx.x;
    unsigned char* im = image + (size_t)b * PS_LAYER_IMAGE;
    const uint32_t* Wq = reinterpret_cast<const uint32_t*>(w.wqkv);
    const uint32_t* Wo = reinterpret_cast<const uint32_t*>(w.wo);
    const uint32_t* W13 = reinterpret_cast<const uint32_t*>(w.w13);
    const uint32_t* W2 = reinterpret_cast<const uint32_t*>(w.w2);
    u32x4 v;
    for (int r = 0; r < 4; ++r) v[r] = Wq[(size_t)(5 * b + r) * 512 + t];
    reinterpret_cast<u32x4*>(im + IM_QKV4)[t] = v;
    reinterpret_cast<uint32_t*>(im + IM_QKV1)[t] = Wq[(size_t)(5 * b + 4) * 512 + t];
    for (int r = 0; r < 4; ++r) v[r] = Wo[(size_t)(4 * b + r) * 512 + t];
    reinterpret_cast<u32x4*>(im + IM_WO)[t] = v;
    for (int c = 0; c < 8; ++c) {
        if (w13_frag) {
            const int rt = c >> 2, j = c & 3, wv = t >> 6, ln = t & 63;
            const size_t row = (size_t)(32 * b + 16 * rt + (ln & 15));
            for (int k = 0; k < 4; ++k) v[k] = W13[row * 512 + (size_t)(64 * wv + 16 * j + 4 * (ln >> 4) + k)];
        } else {
            for (int r = 0; r < 4; ++r) v[r] = W13[(size_t)(32 * b + 4 * c + r) * 512 + t];
        }
        reinterpret_cast<u32x4*>(im + IM_W13)[c * PF_THREADS + t] = v;
    }
    for (int q = 0; q < 4; ++q) {
        for (int r = 0; r < 4; ++r) v[r] = W2[(size_t)(4 * b + r) * 2048 + 512 * q + t];
        reinterpret_cast<u32x4*>(im + IM_W2)[q * PF_THREADS + t] = v;
    }
}
__global__ __launch_bounds__(PF_THREADS) void k_ps_pack_head(const uint32_t* __restrict__ W, int n_rows, unsigned char* __restrict__ image) {
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = 0; c < 2; ++c) {
        u32x4 v;
        for (int r = 0; r < 4; ++r) { const int row = 8 * b + 4 * c + r; v[r] = row < n_rows ? W[(size_t)row * 512 + t] : 0u; }
        reinterpret_cast<u32x4*>(image + (size_t)b * PS_HEAD_IMAGE)[c * PF_THREADS + t] = v;
    }
}
// FS_FP8 images (layout above).  W: row-major e4m3 bytes; a lane's pair of a row = one 16-bit load
__global__ __launch_bounds__(PF_THREADS) void k_ps_pack_layer_fp8(LayerW w, unsigned char* __restrict__ image /*[PF_BLOCKS][PS_LAYER_IMAGE_FP8]*/,
                                                                  float* __restrict__ scales /*[PF_BLOCKS][PS_SC]*/) {
    const int b = blockIdx.x, t = threadIdx.x;
    unsigned char* im = image + (size_t)b * PS_LAYER_IMAGE_FP8;
    const uint16_t* Wq = reinterpret_cast<const uint16_t*>(w.wqkv);  // [rows][512] pairs
    const uint16_t* Wo = reinterpret_cast<const uint16_t*>(w.wo);
    const uint16_t* W13 = reinterpret_cast<const uint16_t*>(w.w13);
    const uint16_t* W2 = reinterpret_cast<const uint16_t*>(w.w2);    // [rows][2048] pairs
    auto two = [](const uint16_t* W, size_t r0, size_t ld, int col) { return (uint32_t)W[r0 * ld + col] | ((uint32_t)W[(r0 + 1) * ld + col] << 16); };
    reinterpret_cast<u32x2*>(im + I8_QKV4)[t] = u32x2{two(Wq, 5 * b, 512, t), two(Wq, 5 * b + 2, 512, t)};
    reinterpret_cast<uint32_t*>(im + I8_QKV1)[t] = (uint32_t)Wq[(size_t)(5 * b + 4) * 512 + t];
    reinterpret_cast<u32x2*>(im + I8_WO)[t] = u32x2{two(Wo, 4 * b, 512, t), two(Wo, 4 * b + 2, 512, t)};
    for (int c = 0; c < 4; ++c) {
        u32x4 v;
        for (int k = 0; k < 4; ++k) v[k] = two(W13, 32 * b + 8 * c + 2 * k, 512, t);
        reinterpret_cast<u32x4*>(im + I8_W13)[c * PF_THREADS + t] = v;
    }
    for (int h = 0; h < 2; ++h) {
        u32x4 v;
        for (int qq = 0; qq < 2; ++qq) {
            const int col = 512 * (2 * h + qq) + t;
            v[2 * qq] = two(W2, 4 * b, 2048, col);
            v[2 * qq + 1] = two(W2, 4 * b + 2, 2048, col);
        }
        reinterpret_cast<u32x4*>(im + I8_W2)[h * PF_THREADS + t] = v;
    }
    if (t < PS_SC) {
        float sc = 0.f;
        if (t < 5) sc = w.s_qkv[5 * b + t];
        else if (t >= SC_WO && t < SC_WO + 4) sc = w.s_o[4 * b + t - SC_WO];
        else if (t >= SC_W13 && t < SC_W13 + 32) sc = w.s_13[32 * b + t - SC_W13];
        else if (t >= SC_W2) sc = w.s_2[4 * b + t - SC_W2];
        scales[(size_t)b * PS_SC + t] = sc;
    }
}
__global__ __launch_bounds__(PF_THREADS) void k_ps_pack_head_fp8(const uint16_t* __restrict__ W, const float* __restrict__ ws, int n_rows,
                                                                 unsigned char* __restrict__ image /*[PF_BLOCKS][PS_HEAD_IMAGE_FP8]*/,
                                                                 float* __restrict__ scales /*[PF_BLOCKS][8]*/) {
    const int b = blockIdx.x, t = threadIdx.x;
    u32x4 v;
    for (int k = 0; k < 4; ++k) {
        const int r0 = 8 * b + 2 * k;
        const uint32_t lo = r0 < n_rows ? W[(size_t)r0 * 512 + t] : 0u, hi = r0 + 1 < n_rows ? W[(size_t)(r0 + 1) * 512 + t] : 0u;
        v[k] = lo | (hi << 16);
    }
    reinterpret_cast<u32x4*>(image + (size_t)b * PS_HEAD_IMAGE_FP8)[t] = v;
    if (t < 8) scales[b * 8 + t] = 8 * b + t < n_rows ? ws[8 * b + t] : 0.f;
}
__global__ void k_ps_copy_norm(const float* __restrict__ src, float* __restrict__ dst) { dst[blockIdx.x * 256 + threadIdx.x] = src[blockIdx.x * 256 + threadIdx.x]; }

// ------------------------------------------------------------------------------------------------ the step kernel
template <bool FP8>
__global__ __launch_bounds__(PF_THREADS) void k_slow_persist(SlowPersistArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long t_entry = A.prof ? wall_clock64() : 0;  // launch-boundary stamps: see k_fast_persist
    float* xs = reinterpret_cast<float*>(smem + S_XS);
    float* red = reinterpret_cast<float*>(smem + S_RED);
    float* qs = reinterpret_cast<float*>(smem + S_QS);
    float* knew = reinterpret_cast<float*>(smem + S_KN);
    float* vnew = reinterpret_cast<float*>(smem + S_VN);
    float* part = reinterpret_cast<float*>(smem + S_PART);
    float* wmax = reinterpret_cast<float*>(smem + S_WMAX);
    int* s_pages = reinterpret_cast<int*>(smem + S_PAGES);

    const int tid_k = threadIdx.x, b = blockIdx.x;
    int tid = tid_k, lane = tid & 63, wave = tid >> 6;
    const int rep = b & (PF_REPL - 1);
    u64* const edges = A.edges;
    const size_t ering = (size_t)PF_REPL * PS_EDGE_CAP;
    const u64* const my_edges = A.edges + (size_t)rep * PS_EDGE_CAP;
    auto pub = [&](unsigned e, int rr, int index, unsigned tag, float value) {
        gu64* g = (gu64*)(edges + (size_t)(e & (PF_RING - 1)) * ering + (size_t)rr * PS_EDGE_CAP + index);
        __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(value), PF_RLX_AGENT);
    };

    if (A.state->done != 0) return;  // generator terminated: the slow sampler ignores the logits (k_sample_slow), replays leave the cache alone
    const int pos = A.state->pos;             // cached tokens; the new token sits at index pos
    const int rpos = pos + A.state->rope_off;
    const unsigned epoch = A.ctl[0];
    const unsigned tag0 = epoch * 256u;
    // attention role
    const int n_sl = A.n_sl;
    const bool att = b < 16 * n_sl;
    const int ah = att ? b / n_sl : 0, as = att ? b % n_sl : 0, ag = ah >> 3;
    const int chunk_t = (pos + n_sl - 1) / n_sl;
    const int t0 = min(pos, as * chunk_t), t1 = min(pos, (as + 1) * chunk_t);  // cached tokens [t0, t1) of this slice
    const bool last_slice = att && as == n_sl - 1;                                // also covers the new token
    const int n_tok = t1 - t0;
    const int n_tiles = att ? max((n_tok + 127) / 128, last_slice ? 1 : 0) : 0;
    if (att) {
        const int p0 = t0 >> 6;
        for (int i = tid; i < 160; i += PF_THREADS) s_pages[i] = (n_tok > 0 && p0 + i <= ((t1 - 1) >> 6)) ? A.page_table[p0 + i] : 0;
    }
    const int page_new = A.page_table[pos >> 6];
    float2 x2 = *reinterpret_cast<const float2*>(A.x + 2 * tid);
    float x0 = x2.x, x1 = x2.y;
    // this token's rotary pair of lane j = tid & 31: the same in all layers.  (Loaded behind S2's sweep in every layer it was a memory round
    // trip in the attention workgroups' critical chain, 24 times per frame.)
    float rope_c = A.cos_t[(size_t)rpos * 32 + (tid & 31)], rope_s = A.sin_t[(size_t)rpos * 32 + (tid & 31)];
    __syncthreads();
    asm volatile("" : "+v"(rope_c), "+v"(rope_s));

    constexpr size_t LIMG = FP8 ? PS_LAYER_IMAGE_FP8 : PS_LAYER_IMAGE;
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(A.wpack) + (size_t)b * LIMG;
    const size_t layer_img = (size_t)PF_BLOCKS * LIMG;
    // weight registers, each filled one stage ahead (bf16: a dword = one row's pair; fp8: two rows' pairs)
    u32x4 wq4 = u32x4{0, 0, 0, 0}, wo4 = u32x4{0, 0, 0, 0}, w13[8], w2r[4];
    u32x2 wq4f = u32x2{0, 0}, wo4f = u32x2{0, 0};
    u32x4 w13f[4], w2f[2];
    uint32_t wq1;
    if constexpr (FP8) {
        wq4f = reinterpret_cast<const u32x2*>(wimg + I8_QKV4)[tid];
        wq1 = reinterpret_cast<const uint32_t*>(wimg + I8_QKV1)[tid];
    } else {
        wq4 = reinterpret_cast<const u32x4*>(wimg + IM_QKV4)[tid];
        wq1 = reinterpret_cast<const uint32_t*>(wimg + IM_QKV1)[tid];
    }
    // fp8: this workgroup's row scales of the current layer (read by the publishing lanes, requested at the top of each stage)
    const float* scl = FP8 ? A.scales + (size_t)b * PS_SC : nullptr;
    const size_t scl_layer = (size_t)PF_BLOCKS * PS_SC;
    u32x4 kreg[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, vreg[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};  // zero unless this slice has cached tokens
    bool dead = false;
    unsigned e = 0;
    int par = 0;
    // W13 (needed by S4) is 56 % of a layer's bytes.  Requested one stage ahead by all 256 workgroups it is a 16.8 MB burst with ~2.4 us to
    // arrive (7 TB/s): S4's sweep, behind it in a wave's return order, completed when the burst did -- S4 waited 2.2 us against 0.9 us for
    // S1 (profiles/r05_stage_profile.txt).  The workgroups WITHOUT an attention item have no sweep and no K/V tile between S1's publish and
    // S3, so they ask for their W13 slice right behind S1's publish, three stages ahead, while HBM is idle; the attention workgroups
    // (64 or 128 of 256) keep asking in S3 and share a burst a quarter to a half the size.  (FISHRT_SLOW_NO_EARLY13=1: everybody in S3.)
    // Rule for every unseen request below: the compiler counts only the loads it knows, so its wait for a KNOWN load that is still in flight
    // (Wo, the norm vector, the next Wqkv rows, the K/V tile) must be made to land BEFORE the unseen loads are issued -- each such value is
    // pinned (empty asm) right behind the sweep that has already completed it; a wait left behind the unseen loads becomes a wait for them.
    // bit 0 (default): workgroups without an attention item request W13 behind S1's publish
    // bit 1: they request W2 (needed by S5, 28 % of the bytes) there as well (measured: 549.5 vs 549.0 us -- no gain, off)
    // Measured and removed: the attention workgroups whose slice is one K/V tile requesting W13 behind S1's publish as well (waves 2..7; waves
    // 0 / 1 behind their q / k / v sweep in S2; the K/V tile through unseen loads with an explicit s_waitcnt vmcnt(9) in S2): 576-582 us against
    // 549, and the greedy tokens stopped being reproducible run to run -- an attention workgroup's S2 is the critical path of the layer, and
    // anything queued in front of its K/V tile costs more than the burst it removes from S3.
    const int early_mode = A.l2_touch;
    const bool s4_mfma = (early_mode & 4) != 0;  // (bf16 images only; wave-uniform)
    const bool early13 = (early_mode & 1) && !att, early2 = (early_mode & 2) && !att;
    auto request_w13 = [&](const unsigned char* wl_) {  // the W13 slice of this layer -> registers, valid after the next sweep
        if constexpr (FP8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) ps_load16_unseen(w13f[c], wl_ + I8_W13 + (size_t)c * PF_THREADS * 16, (unsigned)tid_k * 16u);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) ps_load16_unseen(w13[c], wl_ + IM_W13 + (size_t)c * PF_THREADS * 16, (unsigned)tid_k * 16u);
        }
    };
    auto request_w2 = [&](const unsigned char* wl_) {
        if constexpr (FP8) {
            ps_load16_unseen(w2f[0], wl_ + I8_W2, (unsigned)tid_k * 16u); ps_load16_unseen(w2f[1], wl_ + I8_W2 + PF_THREADS * 16, (unsigned)tid_k * 16u);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) ps_load16_unseen(w2r[q], wl_ + IM_W2 + (size_t)q * PF_THREADS * 16, (unsigned)tid_k * 16u);
        }
    };
    unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = A.prof ? wall_clock64() : 0;  // [1..6] work of S1..S5 / head, [9..14] the wait (nap + sweep) in front of it
    const unsigned long long t_timers = t_last;
#define PS_TICK(k) do { if (A.prof) { const unsigned long long n_ = wall_clock64(); tk[k] += n_ - t_last; t_last = n_; } } while (0)

    // tile 0 of this workgroup's slice is the same rows of every layer's pool: its element offsets are computed ONCE (the per-layer request
    // used to redo two LDS page lookups + 64-bit address arithmetic per lane inside S1, on the critical path of the qkv edge: an attention
    // workgroup published 0.65 us later than the others, profiles/r05_stage_profile.txt)
    uint32_t kv_off0[2] = {0u, 0u};
    if (att && n_tok > 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + PF_THREADS * u;
            const int t = min(t0 + (i >> 3), max(t1 - 1, t0));
            const int pg = s_pages[(t >> 6) - (t0 >> 6)];
            kv_off0[u] = (uint32_t)(((pg * 2 + ag) * KV_PAGE + (t & 63)) * 64 + (i & 7) * 8);
        }
    }
    auto load_kv_tile0 = [&](int l) {
        const uint16_t* kpool = reinterpret_cast<const uint16_t*>(A.kv_pool) + (size_t)l * 2 * A.layer_half;
        const uint16_t* vpool = kpool + A.layer_half;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kreg[u] = *reinterpret_cast<const u32x4*>(kpool + kv_off0[u]);
            vreg[u] = *reinterpret_cast<const u32x4*>(vpool + kv_off0[u]);
        }
    };
    auto load_kv_tile = [&](int l, int tile) {  // 128 tokens x 64 dims of K and V of kv head ag: lane unit i = tid + 512 u -> token i >> 3, 16-B slice i & 7
        const uint16_t* kpool = reinterpret_cast<const uint16_t*>(A.kv_pool) + (size_t)l * 2 * A.layer_half;
        const uint16_t* vpool = kpool + A.layer_half;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + PF_THREADS * u;
            const int t = min(t0 + tile * 128 + (i >> 3), max(t1 - 1, t0));  // clamped: rows past the slice are masked at use
            const int pg = s_pages[(t >> 6) - (t0 >> 6)];
            const size_t off = ((size_t)(pg * 2 + ag) * KV_PAGE + (t & 63)) * 64 + (size_t)(i & 7) * 8;
            kreg[u] = *reinterpret_cast<const u32x4*>(kpool + off);
            vreg[u] = *reinterpret_cast<const u32x4*>(vpool + off);
        }
    };

#pragma unroll 1
    for (int l = 0; l < A.n_layer; ++l) {
        const unsigned char* wl = wimg + (size_t)l * layer_img;
        // ================= S1
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * l) * 1024 + 2 * tid);
            if (l > 0) {
                u32x4 v;
                pf_nap_before_sweep(A.naps[0]);
                pf_sweep1(my_edges + (size_t)(e & 3) * ering, tid, tag0 + e + 1, v, dead, A.ctl);
                x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
                ++e;
                PS_TICK(9);
            }
            const bool kv_late = (early_mode & 8) != 0;  // (bit 3: the tile request BEHIND the publish, so that the publishing stores do not queue behind 32 KB of loads in the CU's vector-memory pipeline)
            if (att && n_tok > 0 && !kv_late) load_kv_tile0(l);  // this layer's first K/V tile, under the qkv stage
            *reinterpret_cast<float2*>(xs + 2 * tid) = make_float2(x0, x1);
            const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float rsc = 1.f;
            if constexpr (FP8) {
                if (tid < 64) rsc = scl[(size_t)l * scl_layer + SC_QKV + min(tid & 15, 4)];
                pf_dot2x2_fp8(wq4f.x, xn0, xn1, a8[0], a8[1]); pf_dot2x2_fp8(wq4f.y, xn0, xn1, a8[2], a8[3]);
                pf_dot2x2_fp8(wq1, xn0, xn1, a8[4], a8[6]);  // (the odd half of this dword is zero)
                a8[6] = 0.f;
            } else {
                a8[0] = pf_dot2(wq4.x, xn0, xn1, 0.f); a8[1] = pf_dot2(wq4.y, xn0, xn1, 0.f);
                a8[2] = pf_dot2(wq4.z, xn0, xn1, 0.f); a8[3] = pf_dot2(wq4.w, xn0, xn1, 0.f);
                a8[4] = pf_dot2(wq1, xn0, xn1, 0.f);
            }
            a8[5] = fmaf(x1, x1, fmaf(x0, x0, 0.f));
            const float r8 = pf_reduce<8>(a8, lane);
            if ((lane & 7) == 0) red[(par * 8 + wave) * PS_RED + (lane >> 3)] = r8;
            __syncthreads();
            if (wave == 0) {  // publishing wave (pf_sum_rows): lane (row r, lane row k) -> replicas k, k + 4
                const int r = min(lane & 15, 5), k = lane >> 4;
                const float* rp = red + (par * 8 + k) * PS_RED;
                float t = pf_sum_rows(rp[r] + rp[4 * PS_RED + r]);
                const float tot = pf_sum_rows(rp[5] + rp[4 * PS_RED + 5]);
                if constexpr (FP8) t *= rsc;
                if ((lane & 15) < 5) {
                    const float val = t * pf_rms_inv(tot, A.eps);
                    pub(e, k, 5 * b + r, tag0 + e + 1, val);
                    pub(e, k + 4, 5 * b + r, tag0 + e + 1, val);
                }
            }
            if (att && n_tok > 0 && kv_late) load_kv_tile0(l);
            if (early13) request_w13(wl);  // (nothing the compiler knows of is in flight here: the stage's own weights and norm vector have been consumed)
            if (early2) request_w2(wl);
            par ^= 1;
            PS_TICK(1);
        }
        // ================= S2: attention of (head ah, slice as)
        if constexpr (FP8) wo4f = reinterpret_cast<const u32x2*>(wl + I8_WO)[tid];  // next stage's weights
        else wo4 = reinterpret_cast<const u32x4*>(wl + IM_WO)[tid];
        if (att) {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const u64* eb = my_edges + (size_t)(e & 3) * ering;
            if (tid < 96) {  // q of head ah (32 units), k / v of kv head ag (32 units each)
                const int unit = tid < 32 ? 32 * ah + tid : (tid < 64 ? 512 + 32 * ag + (tid - 32) : 576 + 32 * ag + (tid - 64));
                u32x4 v;
                pf_nap_before_sweep(A.naps[1]);
                pf_sweep1(eb, unit, tag0 + e + 1, v, dead, A.ctl);
                const float a0 = __uint_as_float(v.x), a1 = __uint_as_float(v.z);
                const int j = tid & 31;
                const float c = rope_c, s = rope_s;
                if (tid < 32) {  // rope_i, then the 1/sqrt(64) of the scores folded into q (a power of two: exact)
                    *reinterpret_cast<float2*>(qs + 2 * j) = make_float2((a0 * c - a1 * s) * 0.125f, (a0 * s + a1 * c) * 0.125f);
                } else if (tid < 64) {
                    const uint32_t k0 = f32_to_bf16_rne(a0 * c - a1 * s), k1 = f32_to_bf16_rne(a0 * s + a1 * c);
                    *reinterpret_cast<float2*>(knew + 2 * j) = make_float2(bf_lo(k0), bf_lo(k1));
                    if (last_slice && (ah & 7) == 0)  // append to the cache (dual_ar.rs:316-324) -- one writer per kv head
                        reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(A.kv_pool) + (size_t)l * 2 * A.layer_half +
                                                    ((size_t)(page_new * 2 + ag) * KV_PAGE + (pos & 63)) * 64)[j] = k0 | (k1 << 16);
                } else {
                    const uint32_t v0 = f32_to_bf16_rne(a0), v1 = f32_to_bf16_rne(a1);
                    *reinterpret_cast<float2*>(vnew + 2 * j) = make_float2(bf_lo(v0), bf_lo(v1));
                    if (last_slice && (ah & 7) == 0)
                        reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(A.kv_pool) + (size_t)l * 2 * A.layer_half + A.layer_half +
                                                    ((size_t)(page_new * 2 + ag) * KV_PAGE + (pos & 63)) * 64)[j] = v0 | (v1 << 16);
                }
            }
            ++e;
            __syncthreads();
            PS_TICK(10);
            // Every WAVE keeps its own running {m, l, o} over the tiles (flash-decoding inside the workgroup): the wave maximum is uniform by DPP /
            // readlane, a lane accumulates p * v for its own tokens and 8-dim slice and p for its token (lanes with du == 0), and nothing crosses
            // lanes or waves until the tiles are done -- one cross-lane sum, ONE block barrier, a merge of the 8 wave partials by the 66 publishing
            // threads.  (Before: a block-wide maximum, a partial table and a running-state merge PER TILE: three barriers and two LDS round
            // trips per tile + two more around the publish staging; the stage was barriers, not arithmetic.)
            float run_m = -1e30f, run_l = 0.f, ro[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int du = tid & 7;
            float qv[8];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(qs + du * 8), q1 = *reinterpret_cast<const float4*>(qs + du * 8 + 4);
                qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
            }
            for (int tile = 0; tile < n_tiles; ++tile) {
                if (tile > 0) load_kv_tile(l, tile);
                // scores of this lane's two tokens (8 lanes per token, 8 dims each)
                float sc[3];
                bool valid[3];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + tile * 128 + ((tid + PF_THREADS * u) >> 3);
                    valid[u] = t < t1;
                    const u32x4 kk = kreg[u];
                    float a = 0.f;
                    a = fmaf(qv[0], bf_lo(kk.x), a); a = fmaf(qv[1], bf_hi(kk.x), a); a = fmaf(qv[2], bf_lo(kk.y), a); a = fmaf(qv[3], bf_hi(kk.y), a);
                    a = fmaf(qv[4], bf_lo(kk.z), a); a = fmaf(qv[5], bf_hi(kk.z), a); a = fmaf(qv[6], bf_lo(kk.w), a); a = fmaf(qv[7], bf_hi(kk.w), a);
                    a += pf_dpp<PF_XOR1>(a); a += pf_dpp<PF_XOR2>(a); a += pf_dpp<PF_HALF_MIRROR>(a);
                    sc[u] = valid[u] ? a : -1e30f;
                }
                // the new token rides with the last tile of the last slice on lanes 0..7 of wave 0
                const bool has_new = last_slice && tile == n_tiles - 1 && tid < 8;
                {
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) a = fmaf(qv[i], knew[du * 8 + i], a);
                    a += pf_dpp<PF_XOR1>(a); a += pf_dpp<PF_XOR2>(a); a += pf_dpp<PF_HALF_MIRROR>(a);
                    valid[2] = has_new;
                    sc[2] = has_new ? a : -1e30f;
                }
                float m = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
                m = fmaxf(m, pf_dpp<PF_XOR1>(m)); m = fmaxf(m, pf_dpp<PF_XOR2>(m)); m = fmaxf(m, pf_dpp<PF_HALF_MIRROR>(m)); m = fmaxf(m, pf_dpp<PF_MIRROR>(m));
                m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31))),
                          fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 47)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63))));
                const float mn = fmaxf(run_m, m), ca = __expf(run_m - mn);  // (wave-uniform)
                run_l *= ca;
#pragma unroll
                for (int i = 0; i < 8; ++i) ro[i] *= ca;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float p = valid[u] ? __expf(sc[u] - mn) : 0.f;
                    if (du == 0) run_l += p;
                    if (u < 2) {
                        const u32x4 vv = vreg[u];
                        ro[0] = fmaf(p, bf_lo(vv.x), ro[0]); ro[1] = fmaf(p, bf_hi(vv.x), ro[1]); ro[2] = fmaf(p, bf_lo(vv.y), ro[2]); ro[3] = fmaf(p, bf_hi(vv.y), ro[3]);
                        ro[4] = fmaf(p, bf_lo(vv.z), ro[4]); ro[5] = fmaf(p, bf_hi(vv.z), ro[5]); ro[6] = fmaf(p, bf_lo(vv.w), ro[6]); ro[7] = fmaf(p, bf_hi(vv.w), ro[7]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) ro[i] = fmaf(p, vnew[du * 8 + i], ro[i]);
                    }
                }
                run_m = mn;
            }
            {   // the wave's partial: sum over its 8 token lanes that share a dim slice (lane ^ 8, ^ 16, ^ 32)
                float o9[9];
#pragma unroll
                for (int i = 0; i < 8; ++i) o9[i] = ro[i];
                o9[8] = run_l;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    float t = o9[i];
                    t += pf_dpp<PS_DROR8>(t);
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
                    t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
                    o9[i] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
                }
                if (lane < 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) part[wave * 72 + lane * 8 + i] = o9[i];
                    if (lane == 0) { part[wave * 72 + 64] = o9[8]; wmax[wave] = run_m; }
                }
            }
            __syncthreads();
            // merge the 8 wave partials (flash-decoding rescale) and publish {o[64], m, l}: 66 threads x 8 replicas (every store instruction
            // writes 64 consecutive granules of one replica; one granule per lane over 8 replicas per value measured +1.3 us on S3's wait)
            if (tid < 66) {
                float M = wmax[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) M = fmaxf(M, wmax[w]);
                float val = 0.f;
                if (tid == 64) val = M;
                else {
#pragma unroll
                    for (int w = 0; w < 8; ++w) val = fmaf(part[w * 72 + (tid < 64 ? tid : 64)], __expf(wmax[w] - M), val);
                }
                const int base = (ah * n_sl + as) * 66;
#pragma unroll
                for (int rr = 0; rr < PF_REPL; ++rr) pub(e, rr, base + tid, tag0 + e + 1, val);
            }
            __syncthreads();
            PS_TICK(2);
        } else {
            ++e;
        }
        // ================= S3: merge the slices of every head -> Wo rows + residual
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const int h = tid >> 5, j = tid & 31;
            const u64* eb = my_edges + (size_t)(e & 3) * ering;
            float rsc = 1.f;  // fp8 row scale of the publishing lanes: requested in FRONT of the sweep and pinned behind it (see the unseen-request rule)
            if constexpr (FP8) { if (tid < 64) rsc = scl[(size_t)l * scl_layer + SC_WO + min(tid & 15, 3)]; }
            float mn = -1e30f, L = 0.f, at0 = 0.f, at1 = 0.f;
            // two passes over the slices would need the maxima first: keep {m, l, o} of up to 16 slices in registers (n_sl <= 16)
            float sm[16], sl_[16], so0[16], so1[16];
            pf_nap_before_sweep(A.naps[2]);
            if (n_sl > 4) {
                // 8 or 16 slices per head: 8 slices = 16 units per lane in ONE sweep (round 4; two dependent 8-unit sweeps cost a second memory-side
                // round trip per layer: 0.8 us x 24 on every frame beyond 512 cached tokens)
                const unsigned off_o = (unsigned)((h * n_sl * 66 + 2 * j) * 8), off_ml = (unsigned)((h * n_sl * 66 + 64) * 8);
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    if (8 * rd < n_sl) {
                        const u64* sb[8];
                        u32x4 vo[8], vm[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) sb[k] = eb + (size_t)(8 * rd + k) * 66;
                        pr_sweep_att8(sb, off_o, off_ml, tag0 + e + 1, vo, vm, dead, A.ctl);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            so0[8 * rd + k] = __uint_as_float(vo[k].x); so1[8 * rd + k] = __uint_as_float(vo[k].z);
                            sm[8 * rd + k] = __uint_as_float(vm[k].x); sl_[8 * rd + k] = __uint_as_float(vm[k].z);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { so0[8 * rd + k] = 0.f; so1[8 * rd + k] = 0.f; sm[8 * rd + k] = -1e30f; sl_[8 * rd + k] = 0.f; }
                    }
                }
            } else
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int s0 = 4 * g4;  // compile-time: the slice arrays stay in registers
                if (s0 < n_sl) {
                    int unit[8];
                    u32x4 v[8];
                    const int ns = min(4, n_sl - s0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int gbase = (h * n_sl + s0 + (k < ns ? k : 0)) * 66;
                        unit[2 * k] = (gbase + 2 * j) >> 1;
                        unit[2 * k + 1] = (gbase + 64) >> 1;
                    }
                    pf_sweep8u(eb, unit, 2 * ns, tag0 + e + 1, v, dead, A.ctl);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        so0[s0 + k] = __uint_as_float(v[2 * k].x); so1[s0 + k] = __uint_as_float(v[2 * k].z);
                        sm[s0 + k] = __uint_as_float(v[2 * k + 1].x); sl_[s0 + k] = __uint_as_float(v[2 * k + 1].z);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { so0[s0 + k] = 0.f; so1[s0 + k] = 0.f; sm[s0 + k] = -1e30f; sl_[s0 + k] = 0.f; }
                }
            }
            ++e;
            PS_TICK(11);
            if constexpr (FP8) {
                asm volatile("" : "+v"(rsc));
                asm volatile("" : "+v"(wo4f));  // (in flight since S2: the compiler's wait for it lands HERE, where the sweep's vmcnt(0) has already satisfied it, not behind the unseen loads)
                if (!early13) request_w13(wl);  // next stage's weights (32 KB per CU)
            } else {
                asm volatile("" : "+v"(wo4));  // (see the fp8 branch)
                if (!early13) request_w13(wl);  // next stage's weights (64 KB per CU), behind the sweep (workgroups without an attention item asked for them in S1)
            }
            // flash-decoding combine of the head's slices, one instantiation per slice count (a runtime bound kept all 16 slots alive: 32
            // predicated v_exp per lane whatever n_sl was)
            auto merge = [&](auto NC) {
                constexpr int NS = decltype(NC)::value;
#pragma unroll
                for (int s = 0; s < NS; ++s) mn = fmaxf(mn, sm[s]);
                float ex[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) { ex[s] = __expf(sm[s] - mn); L += sl_[s] * ex[s]; }
                const float inv = 1.f / L;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float wj = ex[s] * inv;
                    at0 = fmaf(wj, so0[s], at0);
                    at1 = fmaf(wj, so1[s], at1);
                }
            };
            switch (n_sl) {
                case 1: merge(std::integral_constant<int, 1>{}); break;
                case 2: merge(std::integral_constant<int, 2>{}); break;
                case 4: merge(std::integral_constant<int, 4>{}); break;
                case 8: merge(std::integral_constant<int, 8>{}); break;
                default: merge(std::integral_constant<int, 16>{}); break;
            }
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FP8) {
                pf_dot2x2_fp8(wo4f.x, at0, at1, a4[0], a4[1]); pf_dot2x2_fp8(wo4f.y, at0, at1, a4[2], a4[3]);
            } else {
                a4[0] = pf_dot2(wo4.x, at0, at1, 0.f); a4[1] = pf_dot2(wo4.y, at0, at1, 0.f);
                a4[2] = pf_dot2(wo4.z, at0, at1, 0.f); a4[3] = pf_dot2(wo4.w, at0, at1, 0.f);
            }
            const float r4 = pf_reduce<4>(a4, lane);
            if ((lane & 15) == 0) red[(par * 8 + wave) * PS_RED + (lane >> 4)] = r4;
            float xres = 0.f;
            if (tid < 64) xres = xs[4 * b + min(tid & 15, 3)];
            __syncthreads();
            if (wave == 0) {
                const int r = min(lane & 15, 3), k = lane >> 4;
                const float* rp = red + (par * 8 + k) * PS_RED;
                float t = pf_sum_rows(rp[r] + rp[4 * PS_RED + r]);
                if constexpr (FP8) t *= rsc;
                if ((lane & 15) < 4) {
                    pub(e, k, 4 * b + r, tag0 + e + 1, xres + t);
                    pub(e, k + 4, 4 * b + r, tag0 + e + 1, xres + t);
                }
            }
            par ^= 1;
            PS_TICK(3);
        }
        // ================= S4: gather h -> RMSNorm folded -> 16 SwiGLU pairs
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * l + 1) * 1024 + 2 * tid);
            float rsa = 1.f, rsb = 1.f;
            if constexpr (FP8) {
                if (tid < 64) { rsa = scl[(size_t)l * scl_layer + SC_W13 + 2 * (tid & 15)]; rsb = scl[(size_t)l * scl_layer + SC_W13 + 2 * (tid & 15) + 1]; }
            }
            u32x4 v;
            pf_nap_before_sweep(A.naps[3]);
            pf_sweep1(my_edges + (size_t)(e & 3) * ering, tid, tag0 + e + 1, v, dead, A.ctl);
            ++e;
            PS_TICK(12);
            float nwx = nw.x, nwy = nw.y;
            asm volatile("" : "+v"(nwx), "+v"(nwy));  // (in flight since the top of the stage: its wait lands here, behind the sweep that completed it)
            if constexpr (FP8) asm volatile("" : "+v"(rsa), "+v"(rsb));
            // The W13 slice was requested through unseen loads (S1 or S3) and has landed with this sweep's vmcnt(0).  Re-define the registers
            // HERE for the compiler: whatever it derives from them is ordered behind this point, and a value it might have carried
            // in another register across the request is dead.  (What this cannot rule out -- a copy or spill of the destination registers
            // BETWEEN request and landing -- is what fs_lm_selftest("persist") runs for after a toolchain change.)
            if constexpr (FP8) {
#pragma unroll
                for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(w13f[c]));
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(w13[c]));
            }
            if (!early2) request_w2(wl);  // next stage's weights (workgroups without an attention item asked for them in S1)
            x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
            *reinterpret_cast<float2*>(xs + 2 * tid) = make_float2(x0, x1);
            const float xn0 = x0 * nwx, xn1 = x1 * nwy;
            const float ssw = pf_wave_sum(fmaf(x1, x1, fmaf(x0, x0, 0.f)));
            if (lane == 0) red[(par * 8 + wave) * PS_RED + 32] = ssw;
            if (!FP8 && s4_mfma) {
                // The 32 x 1024 GEMV on the matrix cores (as S3 of k_fast_persist, lm_persist.hip): x . g is split into three bf16 terms
                // (truncation: hi + mid + lo == the f32 value exactly) that become columns 0 / 1 / 2 of the B operand; a wave multiplies its
                // 128-deep K slice of the two 16-row tiles (8 x v_mfma_f32_16x16x32_bf16, A = the streamed fragments) and the K reduction
                // happens inside the instruction: 24 VALU operations per lane instead of 128 unpack / FMA + two 16-value halving trees.
                uint32_t* xb = reinterpret_cast<uint32_t*>(smem + S_XB);
                uint32_t pk[3];
                {
                    const uint32_t h0 = __float_as_uint(xn0) & 0xFFFF0000u, h1 = __float_as_uint(xn1) & 0xFFFF0000u;
                    const float r0 = xn0 - __uint_as_float(h0), r1 = xn1 - __uint_as_float(h1);
                    const uint32_t m0 = __float_as_uint(r0) & 0xFFFF0000u, m1 = __float_as_uint(r1) & 0xFFFF0000u;
                    const uint32_t l0 = __float_as_uint(r0 - __uint_as_float(m0)), l1 = __float_as_uint(r1 - __uint_as_float(m1));
                    pk[0] = (h0 >> 16) | h1; pk[1] = (m0 >> 16) | m1; pk[2] = (l0 >> 16) | (l1 & 0xFFFF0000u);
                }
                // a wave's K slice [128 wave, +128) is exactly what its own 64 lanes swept: the LDS round trip is a transpose inside the wave
                xb[tid] = pk[0]; xb[512 + tid] = pk[1]; xb[1024 + tid] = pk[2];
                if (lane < 4) xb[1536 + 4 * wave + lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                const int n = lane & 15, q4 = lane >> 4;
                const u32x4* xbv = reinterpret_cast<const u32x4*>(smem + S_XB);
                const int zslot = 384 + wave;
                const int src = n < 3 ? n * 128 + 16 * wave + q4 : zslot;
                ps_f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const ps_bf16x8_t bv = __builtin_bit_cast(ps_bf16x8_t, xbv[n < 3 ? src + 4 * j : zslot]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ps_bf16x8_t, w13[j]), bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ps_bf16x8_t, w13[4 + j]), bv, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // columns 0 + 1 + 2 (row_shl: lane i reads lane i + 1 / i + 2 of its row of 16)
                    acc0[r] += pf_dpp<0x101>(acc0[r]) + pf_dpp<0x102>(acc0[r]);
                    acc1[r] += pf_dpp<0x101>(acc1[r]) + pf_dpp<0x102>(acc1[r]);
                }
                if (n == 0) {  // D[row 4 q4 + r][column 0]
                    *reinterpret_cast<float4*>(red + (par * 8 + wave) * PS_RED + 4 * q4) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
                    *reinterpret_cast<float4*>(red + (par * 8 + wave) * PS_RED + 16 + 4 * q4) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
                }
            } else {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float a16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if constexpr (FP8) {
    #pragma unroll
                        for (int c = 0; c < 2; ++c) {  // rows 16 half + 8 c ..
                            const u32x4 w = w13f[half * 2 + c];
                            pf_dot2x2_fp8(w.x, xn0, xn1, a16[8 * c], a16[8 * c + 1]); pf_dot2x2_fp8(w.y, xn0, xn1, a16[8 * c + 2], a16[8 * c + 3]);
                            pf_dot2x2_fp8(w.z, xn0, xn1, a16[8 * c + 4], a16[8 * c + 5]); pf_dot2x2_fp8(w.w, xn0, xn1, a16[8 * c + 6], a16[8 * c + 7]);
                        }
                    } else {
    #pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 w = w13[half * 4 + c];
                            a16[4 * c] = pf_dot2(w.x, xn0, xn1, 0.f); a16[4 * c + 1] = pf_dot2(w.y, xn0, xn1, 0.f);
                            a16[4 * c + 2] = pf_dot2(w.z, xn0, xn1, 0.f); a16[4 * c + 3] = pf_dot2(w.w, xn0, xn1, 0.f);
                        }
                    }
                    const float r16 = pf_reduce<16>(a16, lane);
                    if ((lane & 3) == 0) red[(par * 8 + wave) * PS_RED + half * 16 + (lane >> 2)] = r16;
                }
            }
            __syncthreads();
            if (wave == 0) {
                const int jj = lane & 15, k = lane >> 4;
                const float* rp = red + (par * 8 + k) * PS_RED;
                const float2 g2 = *reinterpret_cast<const float2*>(rp + 2 * jj), h2 = *reinterpret_cast<const float2*>(rp + 4 * PS_RED + 2 * jj);
                float ga = pf_sum_rows(g2.x + h2.x), gb = pf_sum_rows(g2.y + h2.y);
                const float tot = pf_sum_rows(rp[32] + rp[4 * PS_RED + 32]);
                const float dni = pf_rms_inv(tot, A.eps);
                if constexpr (FP8) { ga *= rsa; gb *= rsb; }
                ga *= dni; gb *= dni;
                const float act = pf_silu(ga) * gb;
                pub(e, k, 16 * b + jj, tag0 + e + 1, act);
                pub(e, k + 4, 16 * b + jj, tag0 + e + 1, act);
            }
            par ^= 1;
            PS_TICK(4);
        }
        // ================= S5: gather the activations -> W2 rows + residual
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            u32x4 v[4];
            pf_nap_before_sweep(A.naps[4]);
            pf_sweep4(my_edges + (size_t)(e & 3) * ering, tid, tag0 + e + 1, v, dead, A.ctl);
            ++e;
            PS_TICK(13);
            if constexpr (FP8) { asm volatile("" : "+v"(w2f[0]), "+v"(w2f[1])); }  // (W2: requested through unseen loads, landed with this sweep -- see S4)
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(w2r[q]));
            }
            float rsc = 1.f;
            if constexpr (FP8) { if (tid < 64) rsc = scl[(size_t)l * scl_layer + SC_W2 + min(tid & 15, 3)]; }
            if (l + 1 < A.n_layer) {  // next layer's Wqkv rows
                if constexpr (FP8) {
                    wq4f = reinterpret_cast<const u32x2*>(wl + layer_img + I8_QKV4)[tid];
                    wq1 = reinterpret_cast<const uint32_t*>(wl + layer_img + I8_QKV1)[tid];
                } else {
                    wq4 = reinterpret_cast<const u32x4*>(wl + layer_img + IM_QKV4)[tid];
                    wq1 = reinterpret_cast<const uint32_t*>(wl + layer_img + IM_QKV1)[tid];
                }
            }
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FP8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 w = w2f[q >> 1];
                    const float c0 = __uint_as_float(v[q].x), c1 = __uint_as_float(v[q].z);
                    pf_dot2x2_fp8((q & 1) ? w.z : w.x, c0, c1, a4[0], a4[1]); pf_dot2x2_fp8((q & 1) ? w.w : w.y, c0, c1, a4[2], a4[3]);
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 w = w2r[q];
                const float c0 = __uint_as_float(v[q].x), c1 = __uint_as_float(v[q].z);
                a4[0] = pf_dot2(w.x, c0, c1, a4[0]); a4[1] = pf_dot2(w.y, c0, c1, a4[1]);
                a4[2] = pf_dot2(w.z, c0, c1, a4[2]); a4[3] = pf_dot2(w.w, c0, c1, a4[3]);
            }
            }
            const float r4 = pf_reduce<4>(a4, lane);
            if ((lane & 15) == 0) red[(par * 8 + wave) * PS_RED + (lane >> 4)] = r4;
            float xres = 0.f;
            if (tid < 64) xres = xs[4 * b + min(tid & 15, 3)];
            __syncthreads();
            if (wave == 0) {
                const int r = min(lane & 15, 3), k = lane >> 4;
                const float* rp = red + (par * 8 + k) * PS_RED;
                float t = pf_sum_rows(rp[r] + rp[4 * PS_RED + r]);
                if constexpr (FP8) t *= rsc;
                if ((lane & 15) < 4) {
                    pub(e, k, 4 * b + r, tag0 + e + 1, xres + t);
                    pub(e, k + 4, 4 * b + r, tag0 + e + 1, xres + t);
                }
            }
            par ^= 1;
            PS_TICK(5);
        }
    }
    // ================= head: gather the final x (= the hidden state the generator hands to the fast decoder) -> norm folded -> 8 rows
    {
        tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
        const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * A.n_layer) * 1024 + 2 * tid);
        const u32x4* hp = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(A.hpack) + (size_t)b * (FP8 ? PS_HEAD_IMAGE_FP8 : PS_HEAD_IMAGE));
        const u32x4 h0 = hp[tid], h1 = FP8 ? u32x4{0, 0, 0, 0} : hp[PF_THREADS + tid];
        float rsc = 1.f;
        if constexpr (FP8) { if (tid < 64) rsc = A.hscales[8 * b + min(tid & 15, 7)]; }
        u32x4 v;
        pf_nap_before_sweep(A.naps[5]);
        pf_sweep1(my_edges + (size_t)(e & 3) * ering, tid, tag0 + e + 1, v, dead, A.ctl);
        ++e;
        PS_TICK(14);
        x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
        if (b == 0) *reinterpret_cast<float2*>(A.x + 2 * tid) = make_float2(x0, x1);
        const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (FP8) {
            pf_dot2x2_fp8(h0.x, xn0, xn1, a8[0], a8[1]); pf_dot2x2_fp8(h0.y, xn0, xn1, a8[2], a8[3]);
            pf_dot2x2_fp8(h0.z, xn0, xn1, a8[4], a8[5]); pf_dot2x2_fp8(h0.w, xn0, xn1, a8[6], a8[7]);
        } else {
            a8[0] = pf_dot2(h0.x, xn0, xn1, 0.f); a8[1] = pf_dot2(h0.y, xn0, xn1, 0.f); a8[2] = pf_dot2(h0.z, xn0, xn1, 0.f); a8[3] = pf_dot2(h0.w, xn0, xn1, 0.f);
            a8[4] = pf_dot2(h1.x, xn0, xn1, 0.f); a8[5] = pf_dot2(h1.y, xn0, xn1, 0.f); a8[6] = pf_dot2(h1.z, xn0, xn1, 0.f); a8[7] = pf_dot2(h1.w, xn0, xn1, 0.f);
        }
        const float ssw = pf_wave_sum(fmaf(x1, x1, fmaf(x0, x0, 0.f)));
        const float r8 = pf_reduce<8>(a8, lane);
        if ((lane & 7) == 0) red[(par * 8 + wave) * PS_RED + (lane >> 3)] = r8;
        if (lane == 0) red[(par * 8 + wave) * PS_RED + 8] = ssw;
        __syncthreads();
        if (wave == 0) {
            const int r = min(lane & 15, 7), k = lane >> 4;
            const float* rp = red + (par * 8 + k) * PS_RED;
            float t = pf_sum_rows(rp[r] + rp[4 * PS_RED + r]);
            const float tot = pf_sum_rows(rp[8] + rp[4 * PS_RED + 8]);
            if constexpr (FP8) t *= rsc;
            if (lane < 8 && 8 * b + r < A.n_head_rows) A.logits[8 * b + r] = t * pf_rms_inv(tot, A.eps);
        }
        PS_TICK(6);
    }
    if (b == 0 && tid == 0) A.ctl[0] = epoch + 1;
    if (A.prof && b == A.prof_wg && tid == 0) {
        for (int k = 0; k < 16; ++k) A.prof[k] += tk[k];
        const unsigned long long t_end = wall_clock64(), peer_end = A.peer_stamps ? A.peer_stamps[0] : 0;
        if (peer_end && t_entry > peer_end && t_entry - peer_end < 20000) { A.prof[17] += t_entry - peer_end; A.prof[20] += 1; }  // (the first frame of a request follows the prefill)
        A.prof[18] += t_timers - t_entry;
        A.prof[19] += t_end - t_last;
        A.prof[16] = t_end;
    }
#undef PS_TICK
}

// ------------------------------------------------------------------------------------------------ host side
size_t slow_persist_pack_bytes(int n_layer, bool fp8) { return (size_t)n_layer * PF_BLOCKS * (fp8 ? PS_LAYER_IMAGE_FP8 : PS_LAYER_IMAGE); }
size_t slow_persist_scale_floats(int n_layer) { return (size_t)n_layer * PF_BLOCKS * PS_SC + (size_t)PF_BLOCKS * 8; }
size_t slow_persist_edge_bytes() { return (size_t)PF_RING * PF_REPL * PS_EDGE_CAP * 8; }

void launch_slow_persist_pack(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, const float* const* norm_ptrs,
                              void* wpack, void* hpack, float* norms_flat, hipStream_t st) {
    for (int l = 0; l < n_layer; ++l)
        hipLaunchKernelGGL(k_ps_pack_layer, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, layers[l],
                           reinterpret_cast<unsigned char*>(wpack) + (size_t)l * PF_BLOCKS * PS_LAYER_IMAGE, slow_s4_mfma() ? 1 : 0);
    hipLaunchKernelGGL(k_ps_pack_head, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, reinterpret_cast<const uint32_t*>(head_w), n_head_rows,
                       reinterpret_cast<unsigned char*>(hpack));
    for (int i = 0; i < 2 * n_layer + 1; ++i)
        hipLaunchKernelGGL(k_ps_copy_norm, dim3(4), dim3(256), 0, st, norm_ptrs[i], norms_flat + (size_t)i * 1024);
    FS_HIP(hipGetLastError());
}

// FS_FP8 images + scales (scales: [n_layer][PF_BLOCKS][PS_SC] then [PF_BLOCKS][8] head scales)
void launch_slow_persist_pack_fp8(const LayerW* layers, int n_layer, const void* head_w, const float* head_s, int n_head_rows,
                                  const float* const* norm_ptrs, void* wpack, void* hpack, float* scales, float* norms_flat, hipStream_t st) {
    for (int l = 0; l < n_layer; ++l)
        hipLaunchKernelGGL(k_ps_pack_layer_fp8, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, layers[l],
                           reinterpret_cast<unsigned char*>(wpack) + (size_t)l * PF_BLOCKS * PS_LAYER_IMAGE_FP8, scales + (size_t)l * PF_BLOCKS * PS_SC);
    hipLaunchKernelGGL(k_ps_pack_head_fp8, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, reinterpret_cast<const uint16_t*>(head_w), head_s, n_head_rows,
                       reinterpret_cast<unsigned char*>(hpack), scales + (size_t)n_layer * PF_BLOCKS * PS_SC);
    for (int i = 0; i < 2 * n_layer + 1; ++i)
        hipLaunchKernelGGL(k_ps_copy_norm, dim3(4), dim3(256), 0, st, norm_ptrs[i], norms_flat + (size_t)i * 1024);
    FS_HIP(hipGetLastError());
}

void launch_slow_persist(const SlowPersistArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slow_persist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS));
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slow_persist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS));
        attr_set = true;
    }
    if (a.scales) hipLaunchKernelGGL(k_slow_persist<true>, dim3(PF_BLOCKS), dim3(PF_THREADS), PS_LDS, st, a);
    else {
        SlowPersistArgs b2 = a;
        if (slow_s4_mfma()) b2.l2_touch |= 4;  // bit 2: the bf16 image carries W13 as MFMA fragments (launch_slow_persist_pack)
        hipLaunchKernelGGL(k_slow_persist<false>, dim3(PF_BLOCKS), dim3(PF_THREADS), PS_LDS, st, b2);
    }
    FS_HIP(hipGetLastError());
}

}  // namespace fs
