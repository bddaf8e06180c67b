// Persistent slow-transformer kernel for R concurrent batch-1 requests (see lm_persist_rows.h): one launch = forward_generate for ONE new
// token of each of R independent sequences (24 blocks over each row's own paged KV + final norm + audio-range head).
//
// Reference semantics (per row identical to lm_persist_slow.hip / the per-node kernels, other summation order):
//   forward_generate   fish_speech_core/lib/lm/dual_ar.rs:574-635 (L == 1: no mask, :360)
//   Attention::forward dual_ar.rs:281-384;  FeedForward :160-165;  TransformerBlock :429-440;  constrain_probs_to_audio generate/utils.rs:13-16
//   multi-request counterpart in the reference: the lock-step static batch, generate/static_batch.rs:117-274
//
// Structure = lm_persist_slow.hip (256 co-resident workgroups x 512 threads, all-gather edges of 8-byte {value, tag} granules, every
// stage's weight slice streamed one stage ahead), with two changes that make a stage serve R rows for little more than one:
//   * every GEMV runs on the matrix cores: the workgroup's weight rows are MFMA A fragments (streamed ONCE per stage for all rows), the R
//     activation vectors -- split into three bf16 terms, exactly -- are the B columns (3 per row; 4 rows per 16-column tile); the K range
//     is split over the 8 waves, each wave transposing its own swept K slice through 4 KB of LDS without a workgroup barrier;
//   * the attention stage has R x 16 heads x n_sl slices = at most 256 work items, one per workgroup, each over its row's own page table.
// Edge buffers carry a row dimension: [ring][replica][row][granules].
#include "lm_persist_rows.h"

#include <hip/hip_runtime.h>

#include "fs_common.h"

namespace fs {

namespace {

#include "lm_persist_dev.h"
#include "lm_persist_rows_dev.h"
#include "lm_bsample_dev.h"

constexpr int PS_DROR8 = 0x128;  // DPP row_ror:8

// byte offsets inside a (layer, workgroup) A-fragment image (same total as the VALU image: no padding rows are stored)
//   Wqkv: [8 waves][4 k-steps][4 q4][5 rows] x 16 B      rows 5b .., k = 128 wave + 32 j + 8 q4 .. + 8
//   Wo  : [8][4][4][4 rows] x 16 B                       rows 4b ..
//   W13 : [2 tiles][8][4][64 lanes] x 16 B               rows 32b + 16 tile + (lane & 15) (interleaved w1 / w3)
//   W2  : [8][16][4][4 rows] x 16 B                      rows 4b .., k-step j: k = 1024 (j >> 2) + 128 wave + 32 (j & 3) + 8 q4 .. + 8
constexpr size_t IR_QKV = 0, IR_WO = 10240, IR_W13 = 18432, IR_W2 = 83968;
static_assert(IR_W2 + 32768 == PS_LAYER_IMAGE, "image layout");

constexpr int RW = 36;  // row partials per (wave, request row): up to 32 weight rows + the sum of squares
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// FS_FP8 row scales: the tables of the slow / fast persistent kernels (lm_persist_slow.hip PS_SC layout: per (layer, workgroup) 48 floats,
// Wqkv 5 at [0], Wo 4 at [8], W13 32 at [12], W2 4 at [44]; lm_persist.h PF_SCL: 48 per fast layer + the 4 head rows at [192])
constexpr int PRS = 48, PRS_QKV = 0, PRS_WO = 8, PRS_W13 = 12, PRS_W2 = 44;

template <int R>
struct SlowLds {
    static constexpr int RPC = R < 4 ? R : 4, NCT = R / RPC;
    static constexpr int XB = 0;                               // [8 waves][NCT][256] x 16 B staging tiles
    static constexpr int XR = XB + 8 * NCT * 4096;             // [R][4] this workgroup's own residual elements
    static constexpr int RED = XR + R * 16;                    // [2][8][R][RW]
    static constexpr int QS = RED + 2 * 8 * R * RW * 4;        // attention item: rope'd q [64]
    static constexpr int KN = QS + 256;
    static constexpr int VN = KN + 256;
    static constexpr int PART = VN + 256;                      // [8][72]
    static constexpr int WMAX = PART + 8 * 72 * 4;
    static constexpr int PAGES = WMAX + 64;                    // int [160]
    static constexpr int END = PAGES + 160 * 4;
    static constexpr int BYTES = END < 96 * 1024 ? 96 * 1024 : END;  // > half of the CU's LDS: one workgroup per CU
};

}  // namespace

// ------------------------------------------------------------------------------------------------ weight images
__global__ __launch_bounds__(PF_THREADS) void k_pr_pack_layer(LayerW w, unsigned char* __restrict__ image /*[PF_BLOCKS][PS_LAYER_IMAGE]*/) {
    const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q4 = lane >> 4;
    unsigned char* im = image + (size_t)b * PS_LAYER_IMAGE;
    const u32x4* Wq = reinterpret_cast<const u32x4*>(w.wqkv);   // row-major bf16: 128 x 16 B per 1024-wide row
    const u32x4* Wo = reinterpret_cast<const u32x4*>(w.wo);
    const u32x4* W13 = reinterpret_cast<const u32x4*>(w.w13);
    const u32x4* W2 = reinterpret_cast<const u32x4*>(w.w2);     // 512 x 16 B per 4096-wide row
    for (int j = 0; j < 4; ++j) {
        const int ku = 16 * wave + 4 * j + q4;  // 16-byte unit of the row: elements 8 ku ..
        if (m < 5) reinterpret_cast<u32x4*>(im + IR_QKV)[((wave * 4 + j) * 4 + q4) * 5 + m] = Wq[(size_t)(5 * b + m) * 128 + ku];
        if (m < 4) reinterpret_cast<u32x4*>(im + IR_WO)[((wave * 4 + j) * 4 + q4) * 4 + m] = Wo[(size_t)(4 * b + m) * 128 + ku];
        for (int tile = 0; tile < 2; ++tile)
            reinterpret_cast<u32x4*>(im + IR_W13)[((tile * 8 + wave) * 4 + j) * 64 + lane] = W13[(size_t)(32 * b + 16 * tile + m) * 128 + ku];
    }
    for (int j = 0; j < 16; ++j) {
        const int ku = 128 * (j >> 2) + 16 * wave + 4 * (j & 3) + q4;
        if (m < 4) reinterpret_cast<u32x4*>(im + IR_W2)[((wave * 16 + j) * 4 + q4) * 4 + m] = W2[(size_t)(4 * b + m) * 512 + ku];
    }
}
// head rows [8b, 8b+8): [8 waves][4 k-steps][4 q4][8 rows] x 16 B (zero beyond n_rows)
__global__ __launch_bounds__(PF_THREADS) void k_pr_pack_head(const u32x4* __restrict__ W, int n_rows, unsigned char* __restrict__ image) {
    const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q4 = lane >> 4;
    if (m >= 8) return;
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * b + m, ku = 16 * wave + 4 * j + q4;
        reinterpret_cast<u32x4*>(image + (size_t)b * PS_HEAD_IMAGE)[((wave * 4 + j) * 4 + q4) * 8 + m] = row < n_rows ? W[(size_t)row * 128 + ku] : u32x4{0, 0, 0, 0};
    }
}

// FS_FP8 handles: the same A-fragment images from the e4m3 byte matrices, WIDENED to bf16 (exact: e4m3 is a subset of bf16) -- the row
// kernels are bound by their stage chain, not by bytes, so the images keep the bf16 layout and the kernels stay one
// instantiation; the per-row scales of the quantiser (the slow persistent kernel's own scale tables) multiply the K-summed row results in
// the publishing lanes, as in k_slow_persist<true>.
__device__ __forceinline__ u32x4 pr_widen8(uint2 q) {  // 8 e4m3 bytes -> 8 bf16
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(q.x, false), b2 = __builtin_amdgcn_cvt_pk_f32_fp8(q.x, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(q.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(q.y, true);
    auto pk = [](float lo, float hi) { return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u); };
    return u32x4{pk(a.x, a.y), pk(b2.x, b2.y), pk(c.x, c.y), pk(d.x, d.y)};
}
__global__ __launch_bounds__(PF_THREADS) void k_pr_pack_layer_fp8(LayerW w, unsigned char* __restrict__ image /*[PF_BLOCKS][PS_LAYER_IMAGE]*/) {
    const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q4 = lane >> 4;
    unsigned char* im = image + (size_t)b * PS_LAYER_IMAGE;
    const uint2* Wq = reinterpret_cast<const uint2*>(w.wqkv);   // row-major e4m3: 128 x 8 B per 1024-wide row
    const uint2* Wo = reinterpret_cast<const uint2*>(w.wo);
    const uint2* W13 = reinterpret_cast<const uint2*>(w.w13);
    const uint2* W2 = reinterpret_cast<const uint2*>(w.w2);     // 512 x 8 B per 4096-wide row
    for (int j = 0; j < 4; ++j) {
        const int ku = 16 * wave + 4 * j + q4;
        if (m < 5) reinterpret_cast<u32x4*>(im + IR_QKV)[((wave * 4 + j) * 4 + q4) * 5 + m] = pr_widen8(Wq[(size_t)(5 * b + m) * 128 + ku]);
        if (m < 4) reinterpret_cast<u32x4*>(im + IR_WO)[((wave * 4 + j) * 4 + q4) * 4 + m] = pr_widen8(Wo[(size_t)(4 * b + m) * 128 + ku]);
        for (int tile = 0; tile < 2; ++tile)
            reinterpret_cast<u32x4*>(im + IR_W13)[((tile * 8 + wave) * 4 + j) * 64 + lane] = pr_widen8(W13[(size_t)(32 * b + 16 * tile + m) * 128 + ku]);
    }
    for (int j = 0; j < 16; ++j) {
        const int ku = 128 * (j >> 2) + 16 * wave + 4 * (j & 3) + q4;
        if (m < 4) reinterpret_cast<u32x4*>(im + IR_W2)[((wave * 16 + j) * 4 + q4) * 4 + m] = pr_widen8(W2[(size_t)(4 * b + m) * 512 + ku]);
    }
}
__global__ __launch_bounds__(PF_THREADS) void k_pr_pack_head_fp8(const uint2* __restrict__ W, int n_rows, unsigned char* __restrict__ image) {
    const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q4 = lane >> 4;
    if (m >= 8) return;
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * b + m, ku = 16 * wave + 4 * j + q4;
        reinterpret_cast<u32x4*>(image + (size_t)b * PS_HEAD_IMAGE)[((wave * 4 + j) * 4 + q4) * 8 + m] = row < n_rows ? pr_widen8(W[(size_t)row * 128 + ku]) : u32x4{0, 0, 0, 0};
    }
}

// ------------------------------------------------------------------------------------------------ the step kernel
template <int R>
__global__ __launch_bounds__(PF_THREADS) void k_slow_rows(RowsSlowArgs A) {
    using L = SlowLds<R>;
    constexpr int RPC = L::RPC, NCT = L::NCT, NSLM = 16 / R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xr = reinterpret_cast<float*>(smem + L::XR);
    float* red = reinterpret_cast<float*>(smem + L::RED);
    float* qs = reinterpret_cast<float*>(smem + L::QS);
    float* knew = reinterpret_cast<float*>(smem + L::KN);
    float* vnew = reinterpret_cast<float*>(smem + L::VN);
    float* part = reinterpret_cast<float*>(smem + L::PART);
    float* wmax = reinterpret_cast<float*>(smem + L::WMAX);
    int* s_pages = reinterpret_cast<int*>(smem + L::PAGES);

    const int tid_k = threadIdx.x, b = blockIdx.x;
    int tid = tid_k, lane = tid & 63, wave = tid >> 6;
    const int rep = b & (PF_REPL - 1);
    constexpr size_t ering = (size_t)PF_REPL * R * PS_EDGE_CAP;
    auto ebase = [&](unsigned e, int rr, int r) -> u64* { return A.edges + (size_t)(e & (PF_RING - 1)) * ering + ((size_t)rr * R + r) * PS_EDGE_CAP; };
    auto pub = [&](unsigned e, int rr, int r, int index, unsigned tag, float value) {
        gu64* g = (gu64*)(ebase(e, rr, r) + index);
        __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(value), PF_RLX_AGENT);
    };

    // ---- rows of this launch (uniform): a row whose generator has terminated is skipped by every workgroup
    unsigned act = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) if (A.state[r].done == 0) act |= 1u << r;
    if (!act) return;
    const int first = __builtin_ctz(act);
    const unsigned epoch = A.ctl[0];
    const unsigned tag0 = epoch * 256u;
    // ---- attention item of this workgroup: (row ar, query head ah, token slice as)
    const int n_sl = A.n_sl;
    const bool item = b < R * 16 * n_sl;
    const int ar = item ? b / (16 * n_sl) : first, ah = item ? (b / n_sl) & 15 : 0, as = item ? b % n_sl : 0, ag = ah >> 3;
    const bool att = item && ((act >> ar) & 1u);
    const int pos = A.state[ar].pos;             // cached tokens of row ar; its new token sits at index pos
    const int rpos = pos + A.state[ar].rope_off;
    const int chunk_t = (pos + n_sl - 1) / n_sl;
    const int t0 = min(pos, as * chunk_t), t1 = min(pos, (as + 1) * chunk_t);
    const bool last_slice = att && as == n_sl - 1;
    const int n_tok = t1 - t0;
    // a tile = 2 or 3 tokens per lane (128 / 192 tokens): with R rows a head has only 16 / R slices, so at R = 4 a 513..768-token cache would
    // need a second, nearly empty tile round (three block barriers) per layer
    const bool g3 = chunk_t > 128;
    const int TS = g3 ? 192 : 128;
    const int n_tiles = att ? max((n_tok + TS - 1) / TS, last_slice ? 1 : 0) : 0;
    const int* ptab = A.page_table + (size_t)ar * A.pt_stride;
    if (att) {
        const int p0 = t0 >> 6;
        for (int i = tid; i < 160; i += PF_THREADS) s_pages[i] = (n_tok > 0 && p0 + i <= ((t1 - 1) >> 6)) ? ptab[p0 + i] : 0;
    }
    const int page_new = ptab[pos >> 6];
    float x0[R], x1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float2 x2 = *reinterpret_cast<const float2*>(A.x + (size_t)r * 1024 + 2 * tid);
        x0[r] = x2.x; x1[r] = x2.y;
    }
    // row ar's rotary pair of lane j = tid & 31, the same in all layers (see k_slow_persist)
    float rope_c = A.cos_t[(size_t)rpos * 32 + (tid & 31)], rope_s = A.sin_t[(size_t)rpos * 32 + (tid & 31)];
    __syncthreads();
    asm volatile("" : "+v"(rope_c), "+v"(rope_s));

    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(A.wimg) + (size_t)b * PS_LAYER_IMAGE;
    const size_t layer_img = (size_t)PF_BLOCKS * PS_LAYER_IMAGE;
    const float* const scl0 = A.scales + (size_t)b * PRS;  // this workgroup's row scales (FS_FP8: the quantiser's; bf16 handles: a table of ones -- an unconditional load the compiler can count, where a load behind `if (scales)` made later waits vmcnt(0): +65 us per 4-row frame), layer l at + l * PF_BLOCKS * PRS
    const u32x4 zero4 = u32x4{0, 0, 0, 0};
    u32x4 wq[4], wo[4], w13[8], w2[16];
    {
        // (lanes m >= rows load row (rows - 1) again instead of being predicated off: their products land in D rows nobody reads, and with
        // unconditional loads the compiler can COUNT the loads in flight -- behind a predicated prefetch it waits vmcnt(0) at the next use of
        // anything loaded earlier, i.e. for the whole prefetch)
        const int m = min(lane & 15, 4), q4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = reinterpret_cast<const u32x4*>(wimg + IR_QKV)[((wave * 4 + j) * 4 + q4) * 5 + m];
    }
    u32x4 kreg[3] = {zero4, zero4, zero4}, vreg[3] = {zero4, zero4, zero4};
    bool dead = false;
    unsigned e = 0;
    int par = 0;
    unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = A.prof ? wall_clock64() : 0;
#define PS_TICK(k) do { if (A.prof) { const unsigned long long n_ = wall_clock64(); tk[k] += n_ - t_last; t_last = n_; } } while (0)

    auto load_kv_tile = [&](int l, int tile) {
        const uint16_t* kpool = reinterpret_cast<const uint16_t*>(A.kv_pool) + (size_t)l * 2 * A.layer_half;
        const uint16_t* vpool = kpool + A.layer_half;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u == 2 && !g3) break;
            const int i = tid + PF_THREADS * u;
            const int t = min(t0 + tile * TS + (i >> 3), max(t1 - 1, t0));
            const int pg = s_pages[(t >> 6) - (t0 >> 6)];
            const size_t off = ((size_t)(pg * 2 + ag) * KV_PAGE + (t & 63)) * 64 + (size_t)(i & 7) * 8;
            kreg[u] = *reinterpret_cast<const u32x4*>(kpool + off);
            vreg[u] = *reinterpret_cast<const u32x4*>(vpool + off);
        }
    };
    // sweep of one 1024-wide vector per row: unit tid of every active row's region (an inactive row's pointer aliases an active one)
    auto sweep_x = [&](unsigned ee, u32x4 (&v)[R]) {
        const u64* bs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) bs[r] = ebase(ee, rep, ((act >> r) & 1u) ? r : first);
        pr_sweep_rows<R>(bs, (unsigned)tid * 16u, tag0 + ee + 1, v, dead, A.ctl);
    };

#pragma unroll 1
    for (int l = 0; l < A.n_layer; ++l) {
        const unsigned char* wl = wimg + (size_t)l * layer_img;
        const float* const scl = scl0 + (size_t)l * PF_BLOCKS * PRS;
        // ================= S1: (gather x) -> RMSNorm folded -> Wqkv rows [5b, 5b+5) of every row
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const int n = lane & 15, q4 = lane >> 4;
            uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * NCT * 4096);
            float* redw = red + (par * 8 + wave) * R * RW;
            const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * l) * 1024 + 2 * tid);
            // (row scale of this lane's publish: requested IN FRONT of the stage's sweep and pinned behind it.  vmcnt retires in order, so a
            // scale requested later waits for every prefetch in front of it -- the K/V tile here, the next stage's weights elsewhere: +85 us
            // per 4-row frame when it was loaded next to its use)
            float psc = scl[PRS_QKV + tid % 5];
            if (l > 0) {
                u32x4 v[R];
                pf_nap_before_sweep(A.naps[0]);
                sweep_x(e, v);
#pragma unroll
                for (int r = 0; r < R; ++r) { x0[r] = __uint_as_float(v[r].x); x1[r] = __uint_as_float(v[r].z); }
                ++e;
            }
            asm volatile("" : "+v"(psc));
            if (att && n_tok > 0) load_kv_tile(l, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) {  // (unguarded: an inactive row's inputs alias an active row's, nothing of it is published, and
                                           // per-row branches would keep the scheduler from interleaving the rows' chains)
                if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + r * 4 + 2 * tid - 4 * b) = make_float2(x0[r], x1[r]);
                pr_stage_pair(xt + (r / RPC) * 1024, 3 * (r % RPC), lane, x0[r] * nw.x, x1[r] * nw.y);
                const float ss = pf_wave_sum(fmaf(x1[r], x1[r], x0[r] * x0[r]));
                if (lane == 0) redw[r * RW + 32] = ss;
            }
            __builtin_amdgcn_wave_barrier();
            f32x4_t acc[1][NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[0][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            pr_mfma_seg<1, NCT>(wq, 4, 0, reinterpret_cast<const u32x4*>(xt), n, q4, acc);
            pr_extract<1, NCT, RPC, 5, RW>(acc, redw, n, q4);
            __syncthreads();
            for (int idx = tid; idx < 5 * R * PF_REPL; idx += PF_THREADS) {
                const int m = idx % 5, r = (idx / 5) % R, rr = idx / (5 * R);
                if ((act >> r) & 1u) {
                    const float* rp = red + (par * 8) * R * RW + r * RW;
                    float t = rp[m], tot = rp[32];
#pragma unroll
                    for (int w = 1; w < 8; ++w) { t += rp[w * R * RW + m]; tot += rp[w * R * RW + 32]; }
                    t *= idx == tid ? psc : scl[PRS_QKV + m];  // (5 R PF_REPL <= 512 for R <= 12: one iteration)
                    pub(e, rr, r, 5 * b + m, tag0 + e + 1, t * pf_rms_inv(tot, A.eps));
                }
            }
            par ^= 1;
            PS_TICK(1);
        }
        // ================= S2: attention item (row ar, head ah, slice as)
        {
            const int m = min(lane & 15, 3), q4 = lane >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) wo[j] = reinterpret_cast<const u32x4*>(wl + IR_WO)[((wave * 4 + j) * 4 + q4) * 4 + m];
        }
        if (att) {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const u64* eb = ebase(e, rep, ar);
            if (tid < 96) {
                const int unit = tid < 32 ? 32 * ah + tid : (tid < 64 ? 512 + 32 * ag + (tid - 32) : 576 + 32 * ag + (tid - 64));
                u32x4 v;
                pf_nap_before_sweep(A.naps[1]);
                pf_sweep1(eb, unit, tag0 + e + 1, v, dead, A.ctl);
                const float a0 = __uint_as_float(v.x), a1 = __uint_as_float(v.z);
                const int j = tid & 31;
                const float c = rope_c, s = rope_s;
                if (tid < 32) {
                    *reinterpret_cast<float2*>(qs + 2 * j) = make_float2((a0 * c - a1 * s) * 0.125f, (a0 * s + a1 * c) * 0.125f);
                } else if (tid < 64) {
                    const uint32_t k0 = f32_to_bf16_rne(a0 * c - a1 * s), k1 = f32_to_bf16_rne(a0 * s + a1 * c);
                    *reinterpret_cast<float2*>(knew + 2 * j) = make_float2(bf_lo(k0), bf_lo(k1));
                    if (last_slice && (ah & 7) == 0)
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
            // every WAVE keeps its own running {m, l, o} over the tiles; one cross-lane sum and ONE block barrier at the end (see k_slow_persist S2)
            float run_m = -1e30f, run_l = 0.f, ro[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int du = tid & 7;
            float qv[8];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(qs + du * 8), q1 = *reinterpret_cast<const float4*>(qs + du * 8 + 4);
                qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
            }
            for (int tile = 0; tile < n_tiles; ++tile) {
                if (tile > 0) load_kv_tile(l, tile);
                float sc[4];
                bool valid[4];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int t = t0 + tile * TS + ((tid + PF_THREADS * u) >> 3);
                    valid[u] = t < t1 && (u < 2 || g3);
                    sc[u] = -1e30f;
                    if (u == 2 && !g3) continue;
                    const u32x4 kk = kreg[u];
                    float a = 0.f;
                    a = fmaf(qv[0], bf_lo(kk.x), a); a = fmaf(qv[1], bf_hi(kk.x), a); a = fmaf(qv[2], bf_lo(kk.y), a); a = fmaf(qv[3], bf_hi(kk.y), a);
                    a = fmaf(qv[4], bf_lo(kk.z), a); a = fmaf(qv[5], bf_hi(kk.z), a); a = fmaf(qv[6], bf_lo(kk.w), a); a = fmaf(qv[7], bf_hi(kk.w), a);
                    a += pf_dpp<PF_XOR1>(a); a += pf_dpp<PF_XOR2>(a); a += pf_dpp<PF_HALF_MIRROR>(a);
                    sc[u] = valid[u] ? a : -1e30f;
                }
                const bool has_new = last_slice && tile == n_tiles - 1 && tid < 8;
                {
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) a = fmaf(qv[i], knew[du * 8 + i], a);
                    a += pf_dpp<PF_XOR1>(a); a += pf_dpp<PF_XOR2>(a); a += pf_dpp<PF_HALF_MIRROR>(a);
                    valid[3] = has_new;
                    sc[3] = has_new ? a : -1e30f;
                }
                float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
                m = fmaxf(m, pf_dpp<PF_XOR1>(m)); m = fmaxf(m, pf_dpp<PF_XOR2>(m)); m = fmaxf(m, pf_dpp<PF_HALF_MIRROR>(m)); m = fmaxf(m, pf_dpp<PF_MIRROR>(m));
                m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31))),
                          fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 47)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63))));
                const float mn = fmaxf(run_m, m), ca = __expf(run_m - mn);  // (wave-uniform)
                run_l *= ca;
#pragma unroll
                for (int i = 0; i < 8; ++i) ro[i] *= ca;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u == 2 && !g3) continue;
                    const float p = valid[u] ? __expf(sc[u] - mn) : 0.f;
                    if (du == 0) run_l += p;
                    if (u < 3) {
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
            {
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
            if (tid < 66) {  // merge the 8 wave partials (flash-decoding rescale) and publish {o[64], m, l}: 66 threads x 8 replicas
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
                for (int rr = 0; rr < PF_REPL; ++rr) pub(e, rr, ar, base + tid, tag0 + e + 1, val);
            }
            PS_TICK(2);
        } else {
            ++e;
        }
        // ================= S3: merge the slices of every (row, head) -> Wo rows [4b, 4b+4) + residual
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const int n = lane & 15, q4 = lane >> 4;
            uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * NCT * 4096);
            float* redw = red + (par * 8 + wave) * R * RW;
            const int h = tid >> 5, j = tid & 31;
            float mM[R], mL[R], mO0[R], mO1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { mM[r] = -1e30f; mL[r] = 0.f; mO0[r] = 0.f; mO1[r] = 0.f; }
            const unsigned off_o = (unsigned)((h * n_sl * 66 + 2 * j) * 8), off_ml = (unsigned)((h * n_sl * 66 + 64) * 8);
            float psc = scl[PRS_WO + (tid & 3)];
            pf_nap_before_sweep(A.naps[2]);
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                // slot = 8 round + k: row slot / NSLM, slice slot % NSLM (a slot past n_sl, or of an inactive row, aliases (first, 0))
                bool any = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = (8 * round + k) / NSLM, s = (8 * round + k) % NSLM;
                    any |= s < n_sl && ((act >> r) & 1u);
                }
                if (any) {
                    const u64* sb[8];
                    u32x4 vo[8], vm[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = (8 * round + k) / NSLM, s = (8 * round + k) % NSLM;
                        const bool ok = s < n_sl && ((act >> r) & 1u);
                        sb[k] = ebase(e, rep, ok ? r : first) + (ok ? s : 0) * 66;
                    }
                    pr_sweep_att8(sb, off_o, off_ml, tag0 + e + 1, vo, vm, dead, A.ctl);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = (8 * round + k) / NSLM, s = (8 * round + k) % NSLM;
                        if (s < n_sl && ((act >> r) & 1u)) {  // online flash-decoding merge, slices in ascending order
                            const float ms = __uint_as_float(vm[k].x), ls = __uint_as_float(vm[k].z);
                            const float mn = fmaxf(mM[r], ms);
                            const float ca = __expf(mM[r] - mn), cb2 = __expf(ms - mn);
                            mL[r] = mL[r] * ca + ls * cb2;
                            mO0[r] = mO0[r] * ca + __uint_as_float(vo[k].x) * cb2;
                            mO1[r] = mO1[r] * ca + __uint_as_float(vo[k].z) * cb2;
                            mM[r] = mn;
                        }
                    }
                }
            }
            ++e;
            PS_TICK(8);
            asm volatile("" : "+v"(psc));
            {   // next stage's weights (64 KB per CU), behind the sweep
                const u32x4* wp = reinterpret_cast<const u32x4*>(wl + IR_W13);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) w13[t * 4 + jj] = wp[((t * 8 + wave) * 4 + jj) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float inv = ((act >> r) & 1u) ? 1.f / mL[r] : 0.f;
                pr_stage_pair(xt + (r / RPC) * 1024, 3 * (r % RPC), lane, mO0[r] * inv, mO1[r] * inv);
            }
            __builtin_amdgcn_wave_barrier();
            f32x4_t acc[1][NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[0][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            pr_mfma_seg<1, NCT>(wo, 4, 0, reinterpret_cast<const u32x4*>(xt), n, q4, acc);
            pr_extract<1, NCT, RPC, 4, RW>(acc, redw, n, q4);
            __syncthreads();
            for (int idx = tid; idx < 4 * R * PF_REPL; idx += PF_THREADS) {
                const int m = idx & 3, r = (idx >> 2) % R, rr = idx / (4 * R);
                if ((act >> r) & 1u) {
                    const float* rp = red + (par * 8) * R * RW + r * RW;
                    float t = rp[m];
#pragma unroll
                    for (int w = 1; w < 8; ++w) t += rp[w * R * RW + m];
                    t *= psc;  // (m == tid & 3 in every iteration: the stride is a multiple of 4)
                    pub(e, rr, r, 4 * b + m, tag0 + e + 1, xr[r * 4 + m] + t);
                }
            }
            par ^= 1;
            PS_TICK(3);
        }
        // ================= S4: gather h -> RMSNorm folded -> 16 SwiGLU pairs of every row
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const int n = lane & 15, q4 = lane >> 4;
            uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * NCT * 4096);
            float* redw = red + (par * 8 + wave) * R * RW;
            const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * l + 1) * 1024 + 2 * tid);
            float psa = scl[PRS_W13 + 2 * (tid & 15)], psb = scl[PRS_W13 + 2 * (tid & 15) + 1];
            u32x4 v[R];
            pf_nap_before_sweep(A.naps[3]);
            sweep_x(e, v);
            ++e;
            PS_TICK(9);
            asm volatile("" : "+v"(psa), "+v"(psb));
            {   // next stage's weights
                const int m = min(lane & 15, 3);
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) w2[jj] = reinterpret_cast<const u32x4*>(wl + IR_W2)[((wave * 16 + jj) * 4 + q4) * 4 + m];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float a = __uint_as_float(v[r].x), c = __uint_as_float(v[r].z);
                if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + r * 4 + 2 * tid - 4 * b) = make_float2(a, c);
                pr_stage_pair(xt + (r / RPC) * 1024, 3 * (r % RPC), lane, a * nw.x, c * nw.y);
                const float ss = pf_wave_sum(fmaf(c, c, a * a));
                if (lane == 0) redw[r * RW + 32] = ss;
            }
            __builtin_amdgcn_wave_barrier();
            f32x4_t acc[2][NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) { acc[0][c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[1][c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
            pr_mfma_seg<2, NCT>(w13, 4, 0, reinterpret_cast<const u32x4*>(xt), n, q4, acc);
            pr_extract<2, NCT, RPC, 32, RW>(acc, redw, n, q4);
            PS_TICK(10);
            __syncthreads();
            PS_TICK(11);
            for (int idx = tid; idx < 16 * R * PF_REPL; idx += PF_THREADS) {
                const int jj = idx & 15, r = (idx >> 4) % R, rr = idx / (16 * R);
                if ((act >> r) & 1u) {
                    const float* rp = red + (par * 8) * R * RW + r * RW;
                    float ga = rp[2 * jj], gb = rp[2 * jj + 1], tot = rp[32];
#pragma unroll
                    for (int w = 1; w < 8; ++w) { ga += rp[w * R * RW + 2 * jj]; gb += rp[w * R * RW + 2 * jj + 1]; tot += rp[w * R * RW + 32]; }
                    const float dni = pf_rms_inv(tot, A.eps);
                    ga *= psa; gb *= psb;  // (jj == tid & 15 in every iteration)
                    ga *= dni; gb *= dni;
                    pub(e, rr, r, 16 * b + jj, tag0 + e + 1, pf_silu(ga) * gb);
                }
            }
            par ^= 1;
            PS_TICK(4);
        }
        // ================= S5: gather the activations -> W2 rows [4b, 4b+4) + residual
        {
            tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
            const int n = lane & 15, q4 = lane >> 4;
            uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * NCT * 4096);
            float* redw = red + (par * 8 + wave) * R * RW;
            f32x4_t acc[1][NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[0][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            unsigned offs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) offs[q] = (unsigned)(tid + PF_THREADS * q) * 16u;
            float psc = scl[PRS_W2 + (tid & 3)];
            pf_nap_before_sweep(A.naps[4]);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const u64* bs[RPC];
                u32x4 v[RPC][4];
#pragma unroll
                for (int i = 0; i < RPC; ++i) bs[i] = ebase(e, rep, ((act >> (c * RPC + i)) & 1u) ? c * RPC + i : first);
                pr_sweep_seg4<RPC>(bs, offs, tag0 + e + 1, v, dead, A.ctl);
                if (c == 0) asm volatile("" : "+v"(psc));
                if (c == NCT - 1 && l + 1 < A.n_layer) {  // next layer's Wqkv rows
                    const int m = min(lane & 15, 4);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) wq[jj] = reinterpret_cast<const u32x4*>(wl + layer_img + IR_QKV)[((wave * 4 + jj) * 4 + q4) * 5 + m];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // the wave's four 128-deep K segments go through the same staging tile, one after the other
#pragma unroll
                    for (int i = 0; i < RPC; ++i) pr_stage_pair(xt + c * 1024, 3 * i, lane, __uint_as_float(v[i][q].x), __uint_as_float(v[i][q].z));
                    __builtin_amdgcn_wave_barrier();
                    f32x4_t a1[1][1] = {{acc[0][c]}};
                    pr_mfma_seg<1, 1>(w2, 16, 4 * q, reinterpret_cast<const u32x4*>(xt + c * 1024), n, q4, a1);
                    acc[0][c] = a1[0][0];
                    __builtin_amdgcn_wave_barrier();
                }
            }
            ++e;
            PS_TICK(12);
            pr_extract<1, NCT, RPC, 4, RW>(acc, redw, n, q4);
            __syncthreads();
            PS_TICK(13);
            for (int idx = tid; idx < 4 * R * PF_REPL; idx += PF_THREADS) {
                const int m = idx & 3, r = (idx >> 2) % R, rr = idx / (4 * R);
                if ((act >> r) & 1u) {
                    const float* rp = red + (par * 8) * R * RW + r * RW;
                    float t = rp[m];
#pragma unroll
                    for (int w = 1; w < 8; ++w) t += rp[w * R * RW + m];
                    t *= psc;
                    pub(e, rr, r, 4 * b + m, tag0 + e + 1, xr[r * 4 + m] + t);
                }
            }
            par ^= 1;
            PS_TICK(5);
        }
    }
    // ================= head: gather the final x (the hidden state handed to the fast decoder) -> norm folded -> 8 rows
    {
        tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
        const int n = lane & 15, q4 = lane >> 4;
        uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * NCT * 4096);
        float* redw = red + (par * 8 + wave) * R * RW;
        const float2 nw = *reinterpret_cast<const float2*>(A.norms + (size_t)(2 * A.n_layer) * 1024 + 2 * tid);
        u32x4 hd[4];
        {
            const int m = min(lane & 15, 7);
            const u32x4* hp = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(A.himg) + (size_t)b * PS_HEAD_IMAGE);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) hd[jj] = hp[((wave * 4 + jj) * 4 + q4) * 8 + m];
        }
        const float phs = A.hscales[8 * b + (tid & 7)];
        u32x4 v[R];
        pf_nap_before_sweep(A.naps[5]);
        sweep_x(e, v);
        ++e;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float a = __uint_as_float(v[r].x), c = __uint_as_float(v[r].z);
            if (b == 0 && ((act >> r) & 1u)) *reinterpret_cast<float2*>(A.x + (size_t)r * 1024 + 2 * tid) = make_float2(a, c);
            pr_stage_pair(xt + (r / RPC) * 1024, 3 * (r % RPC), lane, a * nw.x, c * nw.y);
            const float ss = pf_wave_sum(fmaf(c, c, a * a));
            if (lane == 0) redw[r * RW + 32] = ss;
        }
        __builtin_amdgcn_wave_barrier();
        f32x4_t acc[1][NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[0][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        pr_mfma_seg<1, NCT>(hd, 4, 0, reinterpret_cast<const u32x4*>(xt), n, q4, acc);
        pr_extract<1, NCT, RPC, 8, RW>(acc, redw, n, q4);
        __syncthreads();
        if (tid < 8 * R) {
            const int m = tid & 7, r = tid >> 3;
            if (((act >> r) & 1u) && 8 * b + m < A.n_head_rows) {
                const float* rp = red + (par * 8) * R * RW + r * RW;
                float t = rp[m], tot = rp[32];
#pragma unroll
                for (int w = 1; w < 8; ++w) { t += rp[w * R * RW + m]; tot += rp[w * R * RW + 32]; }
                t *= phs;
                A.logits[(size_t)r * PR_LD + 8 * b + m] = t * pf_rms_inv(tot, A.eps);
            }
        }
        PS_TICK(6);
    }
    if (b == 0 && tid == 0) {
        A.ctl[0] = epoch + 1;
        if (A.prof) for (int k = 0; k < 16; ++k) A.prof[k] += tk[k];
    }
#undef PS_TICK
}

// dword-major copy of the batch-1 image's 40 row-pair dwords per lane (lm_persist.hip k_pf_pack chunks 0..9): [PF_BLOCKS][40][512] u32, so
// that a stage's 4-5 row pairs are coalesced 4-byte loads -- the row kernel streams them one stage ahead instead of holding them
__global__ __launch_bounds__(PF_THREADS) void k_pr_pack_rowpairs(const u32x4* __restrict__ pack, uint32_t* __restrict__ out) {
    const int b = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    const u32x4 v = pack[((size_t)b * PF_CHUNKS + c) * PF_THREADS + t];
    uint32_t* o = out + ((size_t)b * 40 + 4 * c) * PF_THREADS + t;
    o[0] = v.x; o[PF_THREADS] = v.y; o[2 * PF_THREADS] = v.z; o[3 * PF_THREADS] = v.w;
}

// ================================================================================================ fast decoder, R request rows
// One launch = the slow-token decision + forward_generate_fast x 8 + the 8 codebook decisions of ONE audio frame of each of R requests
// (dual_ar.rs:638-673, single_batch.rs:102-210), greedy decoding.  Same residency as k_fast_persist (lm_persist.hip: row pairs of Wqkv /
// Wo / fast_output + the W13 MFMA fragments of all four layers in 168 VGPRs per lane), except that W2 is streamed one stage ahead from
// the same image (its 128 KB of LDS hold the R rows' fast-decoder K/V caches instead); every stage serves the R rows: the W13 stage as
// 3 R matrix-core columns, the 4-5-row stages on the VALU with ONE halving tree over all rows' partial sums.
namespace {

constexpr int FRW = 36;  // row partials per (wave, request row)

template <int R>
struct FastLds {
    static constexpr int KC = 0;                                 // [R][4 layers][8 pos][64] bf16 pairs
    static constexpr int VC = KC + R * PF_LAYERS * 8 * 64 * 4;
    static constexpr int QS = VC + R * PF_LAYERS * 8 * 64 * 4;   // [R][1024] rope'd q
    static constexpr int XB = QS + R * 4096;                     // [8 waves][256] x 16 B staging tiles (one column tile: R <= 4)
    static constexpr int XR = XB + 8 * 4096;                     // [R][4]
    static constexpr int RED = XR + R * 16;                      // [2][8][R][FRW]
    static constexpr int SC = RED + 2 * 8 * R * FRW * 4;         // [R][16][8]
    static constexpr int AMAX = SC + R * 128 * 4;                // [2][R][8][2]
    static constexpr int ROPE = AMAX + 2 * R * 8 * 2 * 4;        // cos [8][32], sin [8][32]
    static constexpr int RING = ROPE + 2 * 8 * 32 * 4;           // per row: ring [8][17], meta [8][2], prev [16], misc [16], cfg [16], StdRng words [16]
    static constexpr int RING_ROW = 8 * 17 + 16 + 16 + 16 + 16 + 16;
    static constexpr int MB = RING + R * RING_ROW * 4;           // [R][512] repetition-penalty mask bits of each lane's two candidates
    static constexpr int SCL = MB + R * 512 * 4;                 // float [PF_SCL] this workgroup's row scales (an LDS read in the publish chain: a global one
                                                                 // retires behind every prefetch in front of it)
    static constexpr int END = SCL + ((PF_SCL * 4 + 15) & ~15);
    static constexpr int BYTES = END < 96 * 1024 ? 96 * 1024 : END;
    static_assert(END <= 160 * 1024, "LDS budget");
    static_assert(sizeof(BSampLds) <= 8 * 4096, "the sampler's scratch aliases the staging tiles (dead during a decision)");
};

// R rows x {q unit, k/v unit}: unit a of row i = b[i] + off_a, unit b = b[i] + off_b
template <int N>
__device__ __forceinline__ void pr_sweep_rows2(const u64* const (&b)[N], unsigned off_a, unsigned off_b, unsigned tag, u32x4 (&va)[N], u32x4 (&vb)[N],
                                               bool& dead, uint32_t* ctl) {
    static_assert(N == 1 || N == 2 || N == 4, "row counts");
    for (unsigned spins = 0;; ++spins) {
        if constexpr (N == 1) {
            asm volatile("global_load_dwordx4 %0, %2, %4 sc1\n\tglobal_load_dwordx4 %1, %3, %4 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(va[0]), "=&v"(vb[0]) : "v"(off_a), "v"(off_b), "s"(b[0]) : "memory");
        } else if constexpr (N == 2) {
            asm volatile("global_load_dwordx4 %0, %4, %6 sc1\n\tglobal_load_dwordx4 %2, %5, %6 sc1\n\t"
                         "global_load_dwordx4 %1, %4, %7 sc1\n\tglobal_load_dwordx4 %3, %5, %7 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(va[0]), "=&v"(va[1]), "=&v"(vb[0]), "=&v"(vb[1]) : "v"(off_a), "v"(off_b), "s"(b[0]), "s"(b[1]) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %8, %10 sc1\n\tglobal_load_dwordx4 %4, %9, %10 sc1\n\t"
                         "global_load_dwordx4 %1, %8, %11 sc1\n\tglobal_load_dwordx4 %5, %9, %11 sc1\n\t"
                         "global_load_dwordx4 %2, %8, %12 sc1\n\tglobal_load_dwordx4 %6, %9, %12 sc1\n\t"
                         "global_load_dwordx4 %3, %8, %13 sc1\n\tglobal_load_dwordx4 %7, %9, %13 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3])
                         : "v"(off_a), "v"(off_b), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]) : "memory");
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok &= pr_ok(va[i], tag) && pr_ok(vb[i], tag);
        PR_SPIN_TAIL(ok)
    }
}

// host ArgMax rule (LAST maximal index) over one wave: candidate (bv, bi) per lane -> {max value, largest index holding it} in every lane.
// All-VALU butterflies (DPP + permlane swaps): the readlane form costs a VALU -> SALU hop (~32 cycles) per step
__device__ __forceinline__ float pr_wave_max_f(float v) {
    v = fmaxf(v, pf_dpp<PF_XOR1>(v)); v = fmaxf(v, pf_dpp<PF_XOR2>(v)); v = fmaxf(v, pf_dpp<PF_HALF_MIRROR>(v)); v = fmaxf(v, pf_dpp<PF_MIRROR>(v));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ int pr_wave_max_i(int v) {
    v = max(v, __builtin_amdgcn_mov_dpp(v, PF_XOR1, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_mov_dpp(v, PF_XOR2, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_mov_dpp(v, PF_HALF_MIRROR, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_mov_dpp(v, PF_MIRROR, 0xF, 0xF, false));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)r[0], (int)r[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)r2[0], (int)r2[1]);
}
__device__ __forceinline__ void pr_wave_argmax(float bv, int bi, float& wm, int& ci) {
    wm = pr_wave_max_f(bv);
    ci = pr_wave_max_i((bv == wm) ? bi : -1);
}

// sum of N = 8, 16 or 32 per-lane values over the wave; lane with (lane & (64 / N - 1)) == 0 holds the total of value lane / (64 / N)
template <int N>
__device__ __forceinline__ float pr_reduce(float (&v)[N], int lane) {
    if constexpr (N == 4 || N == 8 || N == 16 || N == 32) return pf_reduce<N>(v, lane);
}

}  // namespace

// SAMPLED: every row decides with the block-parallel top-k / top-p / WeightedIndex sampler of lm_bsample_dev.h (temp > 0, 0 < top_k <= 256:
// the server default) on its own StdRng stream -- one row after the other (the sampler is a block-wide routine); the greedy instantiation
// keeps its register allocation
template <int R, bool SAMPLED>
__global__ __launch_bounds__(PF_THREADS) void k_fast_rows(RowsFastArgs A) {
    using L = FastLds<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* kc = reinterpret_cast<uint32_t*>(smem + L::KC);
    uint32_t* vc = reinterpret_cast<uint32_t*>(smem + L::VC);
    float* qs = reinterpret_cast<float*>(smem + L::QS);   // S2: rope'd q of every row; between a decision and the next S1: the rows' next inputs
    float* xr = reinterpret_cast<float*>(smem + L::XR);
    float* red = reinterpret_cast<float*>(smem + L::RED);
    float* sc = reinterpret_cast<float*>(smem + L::SC);
    float* amax = reinterpret_cast<float*>(smem + L::AMAX);
    float* rope_c = reinterpret_cast<float*>(smem + L::ROPE);
    float* rope_s = rope_c + 8 * 32;
    int* s_ring = reinterpret_cast<int*>(smem + L::RING);  // per row: [0,136) ring, [136,152) meta, [152,168) prev, [168,184) misc, [184,192) cfg
    uint32_t* s_mb = reinterpret_cast<uint32_t*>(smem + L::MB);
    float* fscl = reinterpret_cast<float*>(smem + L::SCL);
    BSampLds& samp = *reinterpret_cast<BSampLds*>(smem + L::XB);
    constexpr int RR = L::RING_ROW;
    // misc: [0] slow token, [1] have_prev, [2] done, [3] epoch (row 0), [4..11] codes of this frame
    // cfg : [0] rep_pen bits, [1] ignore_eos, [2] im_end_id, [3] audio_base, [4] sem_lo, [5] sem_hi, [6] temp bits, [7] top_p bits, [8] top_k
    // words (SAMPLED): the StdRng output words of this frame's 9 draws

    const int tid_k = threadIdx.x, b = blockIdx.x;
    int tid = tid_k, lane = tid & 63, wave = tid >> 6;
    const int rep = b & (PF_REPL - 1);
    constexpr size_t ering = (size_t)PF_REPL * R * PF_EDGE_CAP;
    auto ebase = [&](unsigned e, int rr, int r) -> u64* { return A.edges + (size_t)(e & (PF_RING - 1)) * ering + ((size_t)rr * R + r) * PF_EDGE_CAP; };
    auto pub = [&](unsigned e, int rr, int r, int index, unsigned tag, float value) {
        gu64* g = (gu64*)(ebase(e, rr, r) + index);
        __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(value), PF_RLX_AGENT);
    };

    // ---- per-frame inputs (read by every workgroup before its first publish; written by workgroup 0 only, behind full edges)
    for (int i = tid; i < R * 168; i += PF_THREADS) {
        const int r = i / 168, k = i % 168;
        int v;
        if (k < 136) v = A.rp_ring[r * 136 + k];
        else if (k < 152) v = A.rp_meta[r * 16 + k - 136];
        else v = (int)A.state[r].prev[k - 152];
        s_ring[r * RR + k] = v;
    }
    if (tid < R) {
        int* misc = s_ring + tid * RR + 168;
        misc[1] = A.state[tid].have_prev;
        misc[2] = A.state[tid].done;
        const SampleCfg& c = A.cfg[tid];
        misc[16] = __float_as_int(c.rep_pen); misc[17] = c.ignore_eos; misc[18] = (int)c.im_end_id; misc[19] = (int)c.audio_base;
        misc[20] = (int)c.sem_lo; misc[21] = (int)c.sem_hi;
        misc[22] = __float_as_int(c.temp); misc[23] = __float_as_int(c.top_p); misc[24] = c.top_k;
        misc[25] = c.legacy; misc[26] = (int)c.pad_id;
        const bool ok = SAMPLED ? (c.temp > 0.f && c.top_k > 0 && c.top_k <= BS_MAXK) : c.temp == 0.f;
        if (!ok && b == 0 && A.state[tid].done == 0) atomicAdd(A.ctl + 2, 1u);  // the host picks the instantiation by the sampling configuration
    }
    if (SAMPLED && tid >= PF_THREADS - 64 && tid < PF_THREADS - 64 + 9 * R) {  // this frame's StdRng words: ~2.4 us of dependent integer work each
        const int k = tid - (PF_THREADS - 64), r = k / 9, i = k % 9;
        s_ring[r * RR + 168 + 32 + i] = (int)chacha12_word(A.rng[r].key, A.rng[r].consumed + (unsigned)i);
    }
    // Fish <= 1.4 (SampleCfg::legacy, one token layout per handle): the slow token is a 2-way {pad, <|im_end|>} draw whatever the temperature
    // (sampling/mod.rs:8-26; single_batch.rs:104-124) -- the greedy instantiation needs that one StdRng word per row too
    if (!SAMPLED && tid >= PF_THREADS - 64 && tid < PF_THREADS - 64 + R && A.cfg[tid - (PF_THREADS - 64)].legacy)
        s_ring[(tid - (PF_THREADS - 64)) * RR + 168 + 32] = (int)chacha12_word(A.rng[tid - (PF_THREADS - 64)].key, A.rng[tid - (PF_THREADS - 64)].consumed);
    if (tid == 64) s_ring[168 + 3] = (int)A.ctl[0];
    if (tid >= 256) { const int i = tid - 256; rope_c[i] = A.cos_t[i]; rope_s[i] = A.sin_t[i]; }
    __syncthreads();
    const unsigned epoch = __builtin_amdgcn_readfirstlane((unsigned)s_ring[168 + 3]);
    unsigned live = 0, hp = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) { if (s_ring[r * RR + 168 + 2] == 0) live |= 1u << r; if (s_ring[r * RR + 168 + 1] != 0) hp |= 1u << r; }
    live = __builtin_amdgcn_readfirstlane(live); hp = __builtin_amdgcn_readfirstlane(hp);  // (read from LDS: uniform by construction)
    if (!live) return;
    const bool legacy = __builtin_amdgcn_readfirstlane(s_ring[168 + 25]) != 0;

    // ---- the slow-token decision of every live row (constrain_probs_to_audio utils.rs:13-16, rescale_semantic_tokens :45-46,
    // single_batch.rs:102-144), redundantly on every workgroup
    bool dead = false;
    unsigned run = 0;  // rows whose fast decoder runs: live and not terminated by <|im_end|> this frame (single_batch.rs:153-156)
    int n_draws[R];
#pragma unroll
    for (int r = 0; r < R; ++r) n_draws[r] = 0;
    {
        const int n = A.n_slow;
        int leg_idx[R];  // legacy draw of row r: 1 = pad, 0 = <|im_end|>
#pragma unroll
        for (int r = 0; r < R; ++r) leg_idx[r] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool lv_r = (live >> r) & 1u;
            if (legacy) {
                // legacy_softmax_sample (sampling/mod.rs:8-26): P(pad) = softmax([pad, eos])[0]; u ~ U[0, 1) = (next_u32 >> 8) * 2^-24 from the row's own
                // seeded StdRng stream (the reference: unseeded thread_rng), temperature ignored -- redundantly on every workgroup, as k_fast_persist
                const float pad = A.slow_logits[(size_t)r * PR_LD], eosl = A.slow_logits[(size_t)r * PR_LD + 1], m = fmaxf(pad, eosl);
                const float e_pad = expf(pad - m), e_eos = expf(eosl - m);
                const float p_pad = e_pad / (e_pad + e_eos);
                const float u = (float)((uint32_t)s_ring[r * RR + 168 + 32] >> 8) * (1.0f / 16777216.0f);
                const bool is_pad = u < p_pad || s_ring[r * RR + 168 + 17] != 0;
                leg_idx[r] = is_pad ? 1 : 0;
                if (lv_r) n_draws[r] += 1;
                if (A.cap && b == 0 && tid == 0 && lv_r && A.state[r].frame < A.cap_frames) {  // (the record of k_fast_persist: the two logits, the uniform draw, the pick)
                    float* cp = A.cap + ((size_t)r * A.cap_frames + A.state[r].frame) * 9 * 2048;
                    cp[0] = pad; cp[1] = eosl; cp[2] = u; cp[2047] = is_pad ? 0.f : 1.f;
                }
                continue;
            }
            float lv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int i = 4 * tid + s;
                lv[s] = (i < n && lv_r) ? A.slow_logits[(size_t)r * PR_LD + i] : -INFINITY;
                if (i == 0 && s_ring[r * RR + 168 + 17]) lv[s] = -INFINITY;
            }
            if (A.cap && b == 0 && lv_r && A.state[r].frame < A.cap_frames)
                *reinterpret_cast<float4*>(A.cap + ((size_t)r * A.cap_frames + A.state[r].frame) * 9 * 2048 + 4 * tid) = make_float4(lv[0], lv[1], lv[2], lv[3]);
            if (SAMPLED) {
                // one draw costs ~7 us on a whole workgroup: row r's is taken by workgroups 8r .. 8r+7 only (one per edge replica) and
                // handed to everybody through a two-granule decision edge {index | consumed << 16} -- R draws side by side instead of
                // one after the other on every workgroup
                if (lv_r && (b >> 3) == r) {
                    int used = 0;
                    const int* cf = s_ring + r * RR + 168 + 16;
                    const int idx = bsample<PF_THREADS, 4, 8>(lv, n, cf[8], (float)(1.0 / (double)__int_as_float(cf[6])), __int_as_float(cf[7]),
                                                           (uint32_t)s_ring[r * RR + 168 + 32], &used, samp);
                    if (tid < 2) pub(0u, b & 7, r, tid, epoch * 256u + 1u, __uint_as_float((uint32_t)idx | ((uint32_t)used << 16)));
                }
            } else {
            float bv = lv[0];
            int bi = 4 * tid;
#pragma unroll
            for (int s = 1; s < 4; ++s) if (!(lv[s] < bv)) { bv = lv[s]; bi = 4 * tid + s; }
            if (bi >= n) bi = -1;
            float wm; int ci;
            pr_wave_argmax(bv, bi, wm, ci);
            if (lane == 0) { amax[(r * 8 + wave) * 2] = wm; amax[(r * 8 + wave) * 2 + 1] = __int_as_float(ci); }
            }
        }
        u32x4 dv[R];
        if (SAMPLED && !legacy) {
            const int first_l = __builtin_ctz(live);
            const u64* bs[R];
#pragma unroll
            for (int r = 0; r < R; ++r) bs[r] = ebase(0u, rep, ((live >> r) & 1u) ? r : first_l);
            pf_nap_before_sweep(A.nap_draw);
            pr_sweep_rows<R>(bs, 0u, epoch * 256u + 1u, dv, dead, A.ctl);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float gv = (SAMPLED || legacy) ? 0.f : amax[(r * 8) * 2];
            int idx = legacy ? 0 : (SAMPLED ? (int)(dv[r].x & 0xFFFFu) : __float_as_int(amax[(r * 8) * 2 + 1]));
            if (SAMPLED && !legacy && ((live >> r) & 1u)) n_draws[r] += (int)(dv[r].x >> 16);
            if (!SAMPLED && !legacy) {
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float v2 = amax[(r * 8 + w) * 2];
                const int i2 = __float_as_int(amax[(r * 8 + w) * 2 + 1]);
                if (v2 > gv || (v2 == gv && i2 > idx)) { gv = v2; idx = i2; }
            }
            }
            const bool lv_r = (live >> r) & 1u;
            if (!legacy && A.cap && b == 0 && tid == 0 && lv_r && A.state[r].frame < A.cap_frames)
                A.cap[((size_t)r * A.cap_frames + A.state[r].frame) * 9 * 2048 + 2047] = (float)idx;
            const int* cf = s_ring + r * RR + 168 + 16;
            const uint32_t c0 = legacy ? (leg_idx[r] ? (uint32_t)cf[10] : (uint32_t)cf[2])
                                       : (idx > 0 ? (uint32_t)cf[3] + (uint32_t)idx : (uint32_t)cf[2]);  // audio_tok()
            if (lv_r && c0 != (uint32_t)cf[2]) run |= 1u << r;
            if (tid == 0) s_ring[r * RR + 168 + 0] = (int)c0;
        }
        __syncthreads();
    }
    run = __builtin_amdgcn_readfirstlane(run);
    if (!run && b >= R) return;  // (the end-of-frame rows are kept by workgroups 0 .. R-1)

    unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = A.prof ? wall_clock64() : 0;
#define PF_TICK(k) do { if (A.prof) { const unsigned long long n_ = wall_clock64(); tk[k] += n_ - t_last; t_last = n_; } } while (0)
    if (run) {
        const int first = __builtin_ctz(run);
        // (uniform bases + 32-bit per-lane indices from the per-stage opaque thread id: 64-bit per-lane addresses of four unrolled layers
        // would be hoisted out of the pass loop into dozens of registers)
        const u32x4* wp = reinterpret_cast<const u32x4*>(A.wpack) + (size_t)b * PF_CHUNKS * PF_THREADS;
        // row pairs (Wqkv 5 + Wo 4 per layer, 4 head rows): streamed one stage ahead from the dword-major image (L2-resident, 80 KB per CU)
        const uint32_t* rpi = A.rowpairs + (size_t)b * 40 * PF_THREADS;
        if (tid < PF_SCL) fscl[tid] = A.scales[(size_t)b * PF_SCL + tid];  // row scales of the image (FS_FP8: the quantiser's; bf16: ones), 48 per layer + head at [192]
        uint32_t wq5[5], wo4[4], wh4[4];
        uint2 wo8[4];  // R >= 2: Wo row pairs of the lane's FOUR attention dims (half-block attention, see S2)
#pragma unroll
        for (int i = 0; i < 5; ++i) wq5[i] = rpi[(unsigned)(i * PF_THREADS + tid)];
        u32x4 w13v[PF_LAYERS][8];
#pragma unroll
        for (int c = PF_ROW_CHUNKS; c < PF_REG_CHUNKS; ++c) w13v[(c - PF_ROW_CHUNKS) / 8][(c - PF_ROW_CHUNKS) % 8] = wp[(unsigned)(c * PF_THREADS + tid)];
        // per row: repetition-penalty mask bits of this lane's two candidates of every codebook, and the first pass's input -> LDS
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            uint32_t mb = 0;
            float2 xin = make_float2(0.f, 0.f);
            if ((run >> r) & 1u) {
#pragma unroll
                for (int cbi = 0; cbi < 8; ++cbi) {
                    const float2 m2 = *reinterpret_cast<const float2*>(A.rp_mask + ((size_t)r * 8 + cbi) * 1024 + 2 * tid);
                    mb |= (m2.x != 1.0f ? 1u : 0u) << (2 * cbi);
                    mb |= (m2.y != 1.0f ? 1u : 0u) << (2 * cbi + 1);
                }
                xin = *reinterpret_cast<const float2*>(A.xf + (size_t)r * 1024 + 2 * tid);
            }
            s_mb[r * 512 + tid] = mb;
            *reinterpret_cast<float2*>(qs + r * 1024 + 2 * tid) = xin;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int l = 0; l < PF_LAYERS; ++l)
#pragma unroll
            for (int u2 = 0; u2 < 8; ++u2) asm volatile("" : "+v"(w13v[l][u2]));
        PF_TICK(0);
        unsigned e = SAMPLED ? 1u : 0u;  // (edge 0: the slow-token draws)
        const unsigned tag0 = epoch * 256u;
        int par = 0;
        auto sweep_x = [&](unsigned ee, u32x4 (&v)[R]) {
            const u64* bs[R];
#pragma unroll
            for (int r = 0; r < R; ++r) bs[r] = ebase(ee, rep, ((run >> r) & 1u) ? r : first);
            pr_sweep_rows<R>(bs, (unsigned)tid * 16u, tag0 + ee + 1, v, dead, A.ctl);
        };
        uint32_t pcode[R];  // the code every row picked in the previous pass (uniform)
#pragma unroll
        for (int r = 0; r < R; ++r) pcode[r] = 0u;
#pragma unroll 1
        for (int cb = 0; cb < 8; ++cb) {
            const int T = cb + 1;
#pragma unroll
            for (int l = 0; l < PF_LAYERS; ++l) {
                // ================= S1: (gather x) -> RMSNorm folded -> Wqkv rows of every row
                // (round 6: layer 0 of the passes 1..7 with the qkv table -- the input row is fast_embeddings[code], so what this stage would publish
                // is row `code` of the table built at load time (lm_persist.hip k_pf_qkv0_table); S2 reads it, the stage and its edge do not exist)
                // (not in the sampled 4-row instantiation: at 256 registers the extra live values cost it 40 spilled W13 fragment registers, +25 us per frame)
                constexpr bool TBL_OK = !(SAMPLED && R == 4);
                const bool from_tbl = TBL_OK && l == 0 && cb > 0 && A.qkv0_tbl != nullptr;
                if (from_tbl) {
                    tid = pf_opaque(tid_k);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // next stage's row pairs
                        if constexpr (R == 1) wo4[i] = rpi[(unsigned)((9 * l + 5 + i) * PF_THREADS + tid)];
                        else wo8[i] = *reinterpret_cast<const uint2*>(rpi + (unsigned)((9 * l + 5 + i) * PF_THREADS + 2 * (tid & 255)));
                    }
                } else {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    const float2 nw = reinterpret_cast<const float2*>(A.norms[2 * l])[(unsigned)tid];
                    u32x4 v[R];
                    if (l > 0) {
                        pf_nap_before_sweep(A.naps[0]);
                        sweep_x(e, v);
                        ++e;
                        PF_TICK(9);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // next stage's row pairs
                        if constexpr (R == 1) wo4[i] = rpi[(unsigned)((9 * l + 5 + i) * PF_THREADS + tid)];
                        else wo8[i] = *reinterpret_cast<const uint2*>(rpi + (unsigned)((9 * l + 5 + i) * PF_THREADS + 2 * (tid & 255)));
                    }
                    // all rows' partial sums through halving trees of at most 16 values (two rows each: 32 live accumulators cost registers
                    // the resident W13 fragments need)
                    constexpr int RG = R < 2 ? R : 2;
#pragma unroll
                    for (int r0 = 0; r0 < R; r0 += RG) {
                        float a[8 * RG];
#pragma unroll
                        for (int i2 = 0; i2 < RG; ++i2) {
                            const int r = r0 + i2;
                            float xa, xb2;
                            if (l > 0) { xa = __uint_as_float(v[r].x); xb2 = __uint_as_float(v[r].z); }
                            else { const float2 t2 = *reinterpret_cast<const float2*>(qs + r * 1024 + 2 * tid); xa = t2.x; xb2 = t2.y; }
                            if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + r * 4 + 2 * tid - 4 * b) = make_float2(xa, xb2);
                            const float xn0 = xa * nw.x, xn1 = xb2 * nw.y;
#pragma unroll
                            for (int i = 0; i < 5; ++i) a[8 * i2 + i] = pf_dot2(wq5[i], xn0, xn1, 0.f);
                            a[8 * i2 + 5] = fmaf(xb2, xb2, xa * xa);
                            a[8 * i2 + 6] = 0.f; a[8 * i2 + 7] = 0.f;
                        }
                        const float tot = pf_reduce<8 * RG>(a, lane);
                        constexpr int SH = RG == 2 ? 2 : 3;  // value index = lane >> SH
                        if ((lane & ((1 << SH) - 1)) == 0) red[(par * 8 + wave) * R * FRW + (r0 + ((lane >> SH) >> 3)) * FRW + ((lane >> SH) & 7)] = tot;
                    }
                    const float psc = fscl[PRS * l + PRS_QKV + tid % 5];
                    __syncthreads();
                    for (int idx = tid; idx < 5 * R * PF_REPL; idx += PF_THREADS) {
                        const int m = idx % 5, r = (idx / 5) % R, rr = idx / (5 * R);
                        if ((run >> r) & 1u) {
                            const float* rp = red + (par * 8) * R * FRW + r * FRW;
                            float t = rp[m], ss = rp[5];
#pragma unroll
                            for (int w = 1; w < 8; ++w) { t += rp[w * R * FRW + m]; ss += rp[w * R * FRW + 5]; }
                            t *= idx == tid ? psc : fscl[PRS * l + PRS_QKV + m];
                            pub(e, rr, r, 5 * b + m, tag0 + e + 1, t * pf_rms_inv(ss, A.eps));
                        }
                    }
                    par ^= 1;
                    PF_TICK(1);
                }
                // ================= S2: gather qkv -> RoPE, KV append, attention over T <= 8 tokens -> Wo rows + residual
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    {
                        u32x4 vq[R], vk[R];
                        const u64* bs[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) bs[r] = ebase(e, rep, ((run >> r) & 1u) ? r : first);
                        if (from_tbl) {
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                // (uniform base + 32-bit index, like every streamed image of this kernel)
                                const float2* t2 = reinterpret_cast<const float2*>(A.qkv0_tbl);
                                const unsigned ro = pcode[r] * 640u + (unsigned)tid;
                                const float2 q2 = t2[ro];
                                vq[r].x = __float_as_uint(q2.x); vq[r].z = __float_as_uint(q2.y);
                                if (tid < 128) {  // k pairs 1024 + 2 tid (tid < 64), v pairs 1152 + 2 (tid - 64): the units 512 + tid of the qkv edge
                                    const float2 k2 = t2[ro + 512u];
                                    vk[r].x = __float_as_uint(k2.x); vk[r].z = __float_as_uint(k2.y);
                                }
                            }
                        } else {
                            pf_nap_before_sweep(A.naps[1]);
                            if (tid < 128) pr_sweep_rows2<R>(bs, (unsigned)tid * 16u, (unsigned)(512 + tid) * 16u, tag0 + e + 1, vq, vk, dead, A.ctl);
                            else pr_sweep_rows<R>(bs, (unsigned)tid * 16u, tag0 + e + 1, vq, dead, A.ctl);
                            ++e;
                        }
                        PF_TICK(10);
                        const int j = tid & 31;
                        const float c = rope_c[cb * 32 + j], s = rope_s[cb * 32 + j];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float qa = __uint_as_float(vq[r].x), qb = __uint_as_float(vq[r].z);
                            *reinterpret_cast<float2*>(qs + r * 1024 + 2 * tid) = make_float2(qa * c - qb * s, qa * s + qb * c);
                            if (tid < 64) {
                                const float ka = __uint_as_float(vk[r].x), kb = __uint_as_float(vk[r].z);
                                kc[((r * PF_LAYERS + l) * 8 + cb) * 64 + tid] = f32_to_bf16_rne(ka * c - kb * s) | (f32_to_bf16_rne(ka * s + kb * c) << 16);
                            } else if (tid < 128) {
                                vc[((r * PF_LAYERS + l) * 8 + cb) * 64 + tid - 64] = f32_to_bf16_rne(__uint_as_float(vk[r].x)) | (f32_to_bf16_rne(__uint_as_float(vk[r].z)) << 16);
                            }
                        }
                    }
                    __syncthreads();
                    if (A.cap && b == 0 && tid < 128) {  // fs_lm_debug_capture: this pass's K / V rows behind its logits (see k_fast_persist)
#pragma unroll 1
                        for (int r = 0; r < R; ++r)
                            if (((run >> r) & 1u) && A.state[r].frame < A.cap_frames)
                                reinterpret_cast<uint32_t*>(A.cap + (((size_t)r * A.cap_frames + A.state[r].frame) * 9 + 1 + cb) * 2048 + 1025)[l * 128 + tid] =
                                    tid < 64 ? kc[((r * PF_LAYERS + l) * 8 + cb) * 64 + tid] : vc[((r * PF_LAYERS + l) * 8 + cb) * 64 + tid - 64];
                    }
                    if constexpr (R == 1) {
                        const int h = tid >> 5, g = h >> 3, p = (tid >> 2) & 7, qd = tid & 3, j = tid & 31;
                        // rows in a REAL loop, two per iteration: unrolled over all rows the scheduler issues every row's LDS reads up front (~24 registers
                        // per row), one at a time each row is a ~1 us chain of dependent LDS / transcendental / cross-lane latencies
                        constexpr int RI = R < 2 ? R : 2;
#pragma unroll 1
                        for (int r0 = 0; r0 < R; r0 += RI) {
                            float accs[RI];
#pragma unroll
                            for (int i2 = 0; i2 < RI; ++i2) {
                                const int r = r0 + i2;
                                const float4* qp = reinterpret_cast<const float4*>(qs + r * 1024 + h * 64 + qd * 16);
                                const u32x4* kp = reinterpret_cast<const u32x4*>(kc + ((r * PF_LAYERS + l) * 8 + p) * 64 + g * 32 + qd * 8);
                                float acc = 0.f;
#pragma unroll
                                for (int i = 0; i < 2; ++i) {
                                    const u32x4 kw = kp[i];
                                    const float4 q0 = qp[2 * i], q1 = qp[2 * i + 1];
                                    acc = fmaf(q0.x, bf_lo(kw.x), acc); acc = fmaf(q0.y, bf_hi(kw.x), acc);
                                    acc = fmaf(q0.z, bf_lo(kw.y), acc); acc = fmaf(q0.w, bf_hi(kw.y), acc);
                                    acc = fmaf(q1.x, bf_lo(kw.z), acc); acc = fmaf(q1.y, bf_hi(kw.z), acc);
                                    acc = fmaf(q1.z, bf_lo(kw.w), acc); acc = fmaf(q1.w, bf_hi(kw.w), acc);
                                }
                                acc += pf_dpp<PF_XOR1>(acc);
                                acc += pf_dpp<PF_XOR2>(acc);
                                accs[i2] = acc * 0.125f;  // 1 / sqrt(64) on K (dual_ar.rs:260): a power of two, exact wherever it is applied
                            }
#pragma unroll
                            for (int i2 = 0; i2 < RI; ++i2) if (qd == 0) sc[(r0 + i2) * 128 + h * 8 + p] = accs[i2];
                            __builtin_amdgcn_wave_barrier();
                            float a[4 * RI];
#pragma unroll
                            for (int i2 = 0; i2 < RI; ++i2) {
                                const int r = r0 + i2;
                                float mn = -1e30f;
#pragma unroll
                                for (int t = 0; t < 8; ++t) if (t < T) mn = fmaxf(mn, sc[r * 128 + h * 8 + t]);
                                float Ls = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
                                for (int t = 0; t < 8; ++t)
                                    if (t < T) {
                                        const float pr = __expf(sc[r * 128 + h * 8 + t] - mn);
                                        Ls += pr;
                                        const uint32_t vw = vc[((r * PF_LAYERS + l) * 8 + t) * 64 + g * 32 + j];
                                        O0 = fmaf(pr, bf_lo(vw), O0);
                                        O1 = fmaf(pr, bf_hi(vw), O1);
                                    }
                                const float inv = 1.f / Ls;
                                const float at0 = O0 * inv, at1 = O1 * inv;
#pragma unroll
                                for (int i = 0; i < 4; ++i) a[4 * i2 + i] = pf_dot2(wo4[i], at0, at1, 0.f);
                            }
                            const float tot = pf_reduce<4 * RI>(a, lane);
                            constexpr int SH = RI == 2 ? 3 : 4;
                            if ((lane & ((1 << SH) - 1)) == 0) red[(par * 8 + wave) * R * FRW + (r0 + ((lane >> SH) >> 2)) * FRW + ((lane >> SH) & 3)] = tot;
                        }
                    } else {
                        // R >= 2: the two HALVES of the block (4 waves each) attend for different rows at the same time -- rows {0, 1} / {2, 3} at
                        // R = 4, row 0 / row 1 at R = 2 -- instead of the whole block taking one row pair after the other: a row pair is a ~1.6 us
                        // chain of dependent LDS / transcendental / cross-lane latencies, not work.  A lane covers head h, positions pp and pp + 4,
                        // 16 of the head's 64 dims in the score phase, and FOUR output dims (two Wo row-pair dwords) in the value phase.
                        constexpr int RH = R / 2;
                        const int half = tid >> 8, tq = tid & 255;
                        const int h = tq >> 4, g = h >> 3, pp = (tq >> 2) & 3, qd = tq & 3, jj = tq & 15;
                        float sa[RH], sb[RH];
#pragma unroll
                        for (int i2 = 0; i2 < RH; ++i2) {
                            const int r = half * RH + i2;
                            const float4* qp = reinterpret_cast<const float4*>(qs + r * 1024 + h * 64 + qd * 16);
                            const u32x4* ka = reinterpret_cast<const u32x4*>(kc + ((r * PF_LAYERS + l) * 8 + pp) * 64 + g * 32 + qd * 8);
                            const u32x4* kb = ka + 64;  // position pp + 4
                            float acc = 0.f, acd = 0.f;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const u32x4 kw = ka[i], kx = kb[i];
                                const float4 q0 = qp[2 * i], q1 = qp[2 * i + 1];
                                acc = fmaf(q0.x, bf_lo(kw.x), acc); acc = fmaf(q0.y, bf_hi(kw.x), acc);
                                acc = fmaf(q0.z, bf_lo(kw.y), acc); acc = fmaf(q0.w, bf_hi(kw.y), acc);
                                acc = fmaf(q1.x, bf_lo(kw.z), acc); acc = fmaf(q1.y, bf_hi(kw.z), acc);
                                acc = fmaf(q1.z, bf_lo(kw.w), acc); acc = fmaf(q1.w, bf_hi(kw.w), acc);
                                acd = fmaf(q0.x, bf_lo(kx.x), acd); acd = fmaf(q0.y, bf_hi(kx.x), acd);
                                acd = fmaf(q0.z, bf_lo(kx.y), acd); acd = fmaf(q0.w, bf_hi(kx.y), acd);
                                acd = fmaf(q1.x, bf_lo(kx.z), acd); acd = fmaf(q1.y, bf_hi(kx.z), acd);
                                acd = fmaf(q1.z, bf_lo(kx.w), acd); acd = fmaf(q1.w, bf_hi(kx.w), acd);
                            }
                            acc += pf_dpp<PF_XOR1>(acc); acd += pf_dpp<PF_XOR1>(acd);
                            acc += pf_dpp<PF_XOR2>(acc); acd += pf_dpp<PF_XOR2>(acd);
                            sa[i2] = acc * 0.125f; sb[i2] = acd * 0.125f;  // 1 / sqrt(64) on K (dual_ar.rs:260): a power of two, exact wherever it is applied
                        }
#pragma unroll
                        for (int i2 = 0; i2 < RH; ++i2)
                            if (qd == 0) { sc[(half * RH + i2) * 128 + h * 8 + pp] = sa[i2]; sc[(half * RH + i2) * 128 + h * 8 + pp + 4] = sb[i2]; }
                        __builtin_amdgcn_wave_barrier();  // (a wave holds four whole heads in both phases)
                        float a[4 * RH];
#pragma unroll
                        for (int i2 = 0; i2 < RH; ++i2) {
                            const int r = half * RH + i2;
                            float mn = -1e30f;
#pragma unroll
                            for (int t = 0; t < 8; ++t) if (t < T) mn = fmaxf(mn, sc[r * 128 + h * 8 + t]);
                            float Ls = 0.f, O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
#pragma unroll
                            for (int t = 0; t < 8; ++t)
                                if (t < T) {
                                    const float pr = __expf(sc[r * 128 + h * 8 + t] - mn);
                                    Ls += pr;
                                    const uint2 vw = *reinterpret_cast<const uint2*>(vc + ((r * PF_LAYERS + l) * 8 + t) * 64 + g * 32 + 2 * jj);
                                    O0 = fmaf(pr, bf_lo(vw.x), O0); O1 = fmaf(pr, bf_hi(vw.x), O1);
                                    O2 = fmaf(pr, bf_lo(vw.y), O2); O3 = fmaf(pr, bf_hi(vw.y), O3);
                                }
                            const float inv = 1.f / Ls;
                            const float at0 = O0 * inv, at1 = O1 * inv, at2 = O2 * inv, at3 = O3 * inv;
#pragma unroll
                            for (int i = 0; i < 4; ++i) a[4 * i2 + i] = pf_dot2(wo8[i].y, at2, at3, pf_dot2(wo8[i].x, at0, at1, 0.f));
                        }
                        const float tot = pf_reduce<4 * RH>(a, lane);
                        constexpr int SH = RH == 2 ? 3 : 4;
                        if ((lane & ((1 << SH) - 1)) == 0) red[(par * 8 + wave) * R * FRW + (half * RH + ((lane >> SH) >> 2)) * FRW + ((lane >> SH) & 3)] = tot;
                    }
                    const float pso = fscl[PRS * l + PRS_WO + (tid & 3)];  // (row scales: requested in front of the barrier)
                    __syncthreads();
                    for (int idx = tid; idx < 4 * R * PF_REPL; idx += PF_THREADS) {
                        const int m = idx & 3, r = (idx >> 2) % R, rr = idx / (4 * R);
                        if ((run >> r) & 1u) {
                            // (R >= 2: the row's Wo partials sit in the four waves of its half of the block)
                            constexpr int NWR = R == 1 ? 8 : 4;
                            const int w0 = R == 1 ? 0 : 4 * (r / (R / 2 > 0 ? R / 2 : 1));
                            const float* rp = red + (par * 8 + w0) * R * FRW + r * FRW;
                            float t = rp[m];
#pragma unroll
                            for (int w = 1; w < NWR; ++w) t += rp[w * R * FRW + m];
                            t *= pso;
                            pub(e, rr, r, 4 * b + m, tag0 + e + 1, xr[r * 4 + m] + t);
                        }
                    }
                    par ^= 1;
                    PF_TICK(2);
                }
                // ================= S3: gather h -> RMSNorm folded -> 16 SwiGLU pairs of every row on the matrix cores
                u32x4 w2r[4];
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    const int n = lane & 15, q4 = lane >> 4;
                    uint32_t* xt = reinterpret_cast<uint32_t*>(smem + L::XB + wave * 4096);
                    float* redw = red + (par * 8 + wave) * R * FRW;
                    const float2 nw = reinterpret_cast<const float2*>(A.norms[2 * l + 1])[(unsigned)tid];
                    {
                        u32x4 v[R];
                        pf_nap_before_sweep(A.naps[2]);
                        sweep_x(e, v);
                        ++e;
                        PF_TICK(11);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float xa = __uint_as_float(v[r].x), xb2 = __uint_as_float(v[r].z);
                            if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + r * 4 + 2 * tid - 4 * b) = make_float2(xa, xb2);
                            pr_stage_pair(xt, 3 * r, lane, xa * nw.x, xb2 * nw.y);
                            const float ss = pf_wave_sum(fmaf(xb2, xb2, xa * xa));
                            if (lane == 0) redw[r * FRW + 32] = ss;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    f32x4_t acc[2][1] = {{f32x4_t{0.f, 0.f, 0.f, 0.f}}, {f32x4_t{0.f, 0.f, 0.f, 0.f}}};
                    pr_mfma_seg<2, 1>(&w13v[l][0], 4, 0, reinterpret_cast<const u32x4*>(xt), n, q4, acc);
                    pr_extract<2, 1, R, 32, FRW>(acc, redw, n, q4);
                    // next stage's weights (32 KB per CU, the same image the batch-1 kernel keeps in LDS), requested behind everything this stage
                    // needs from memory: a wave's loads return in order
#pragma unroll
                    for (int q = 0; q < 4; ++q) w2r[q] = wp[(unsigned)((PF_REG_CHUNKS + 4 * l + q) * PF_THREADS + tid)];
                    const float psa = fscl[PRS * l + PRS_W13 + 2 * (tid & 15)], psb = fscl[PRS * l + PRS_W13 + 2 * (tid & 15) + 1];
                    PF_TICK(8);
                    __syncthreads();
                    for (int idx = tid; idx < 16 * R * PF_REPL; idx += PF_THREADS) {
                        const int jj = idx & 15, r = (idx >> 4) % R, rr = idx / (16 * R);
                        if ((run >> r) & 1u) {
                            const float* rp = red + (par * 8) * R * FRW + r * FRW;
                            float ga = rp[2 * jj], gb = rp[2 * jj + 1], ss = rp[32];
#pragma unroll
                            for (int w = 1; w < 8; ++w) { ga += rp[w * R * FRW + 2 * jj]; gb += rp[w * R * FRW + 2 * jj + 1]; ss += rp[w * R * FRW + 32]; }
                            const float dni = pf_rms_inv(ss, A.eps);
                            ga *= psa; gb *= psb;
                            ga *= dni; gb *= dni;
                            pub(e, rr, r, 16 * b + jj, tag0 + e + 1, pf_silu(ga) * gb);
                        }
                    }
                    par ^= 1;
                    PF_TICK(3);
                }
                // ================= S4: gather the 4096 activations of every row -> W2 rows (streamed) + residual
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    constexpr int RH = R < 2 ? R : 2;  // rows per sweep (8 x 16 B in flight per lane; all four rows in one sweep measured slower: 1017 vs 995 us per frame)
                    unsigned offs[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) offs[q] = (unsigned)(tid + PF_THREADS * q) * 16u;
                    pf_nap_before_sweep(A.naps[3]);
                    auto rows_dot = [&](const u32x4 (&v)[RH][4], int r0) {  // rows r0 .. r0 + RH - 1: W2 row pairs . activations -> halving tree -> red
                        float a[4 * RH];
#pragma unroll
                        for (int i = 0; i < RH; ++i) {
                            float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const u32x4 w = w2r[q];
                                const float c0 = __uint_as_float(v[i][q].x), c1 = __uint_as_float(v[i][q].z);
                                a4[0] = pf_dot2(w.x, c0, c1, a4[0]); a4[1] = pf_dot2(w.y, c0, c1, a4[1]);
                                a4[2] = pf_dot2(w.z, c0, c1, a4[2]); a4[3] = pf_dot2(w.w, c0, c1, a4[3]);
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) a[4 * i + k] = a4[k];
                        }
                        const float tot = pf_reduce<4 * RH>(a, lane);
                        constexpr int SH = RH == 4 ? 2 : (RH == 2 ? 3 : 4);
                        if ((lane & ((1 << SH) - 1)) == 0) red[(par * 8 + wave) * R * FRW + (r0 + ((lane >> SH) >> 2)) * FRW + ((lane >> SH) & 3)] = tot;
                    };
                    if constexpr (R == 4) {
                        // rows 0, 1: blocking sweep; rows 2, 3: requested right behind it and waited for behind the arithmetic on rows 0, 1
                        // (two blocking sweeps one after the other cost a second memory-side round trip per stage: 0.8 us)
                        const u64 *bsA[2], *bsB[2];
                        u32x4 vA[2][4], vB[2][4];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bsA[i] = ebase(e, rep, ((run >> i) & 1u) ? i : first);
                            bsB[i] = ebase(e, rep, ((run >> (2 + i)) & 1u) ? 2 + i : first);
                        }
                        pr_sweep_seg4<2>(bsA, offs, tag0 + e + 1, vA, dead, A.ctl);
#pragma unroll
                        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(w2r[q]));  // (the compiler's wait for the prefetched W2 rows lands HERE)
                        pr_issue_seg4_2(bsB, offs, vB);
                        rows_dot(vA, 0);
                        if (!pr_finish_seg4_2(vB, tag0 + e + 1) && !dead) pr_sweep_seg4<2>(bsB, offs, tag0 + e + 1, vB, dead, A.ctl);
                        rows_dot(vB, 2);
                    } else {
                        const u64* bs[RH];
                        u32x4 v[RH][4];
#pragma unroll
                        for (int i = 0; i < RH; ++i) bs[i] = ebase(e, rep, ((run >> i) & 1u) ? i : first);
                        pr_sweep_seg4<RH>(bs, offs, tag0 + e + 1, v, dead, A.ctl);
                        rows_dot(v, 0);
                    }
                    ++e;
                    if (l + 1 < PF_LAYERS) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) wq5[i] = rpi[(unsigned)((9 * (l + 1) + i) * PF_THREADS + tid)];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) wh4[i] = rpi[(unsigned)((9 * PF_LAYERS + i) * PF_THREADS + tid)];
                    }
                    const float ps2 = fscl[PRS * l + PRS_W2 + (tid & 3)];
                    PF_TICK(12);
                    __syncthreads();
                    for (int idx = tid; idx < 4 * R * PF_REPL; idx += PF_THREADS) {
                        const int m = idx & 3, r = (idx >> 2) % R, rr = idx / (4 * R);
                        if ((run >> r) & 1u) {
                            const float* rp = red + (par * 8) * R * FRW + r * FRW;
                            float t = rp[m];
#pragma unroll
                            for (int w = 1; w < 8; ++w) t += rp[w * R * FRW + m];
                            t *= ps2;
                            pub(e, rr, r, 4 * b + m, tag0 + e + 1, xr[r * 4 + m] + t);
                        }
                    }
                    par ^= 1;
                    PF_TICK(4);
                }
            }
            // ================= head: gather x -> fast_norm folded -> 4 rows of fast_output per row
            {
                tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                const float2 nw = reinterpret_cast<const float2*>(A.norms[2 * PF_LAYERS])[(unsigned)tid];
                u32x4 v[R];
                pf_nap_before_sweep(A.naps[4]);
                sweep_x(e, v);
                ++e;
                PF_TICK(13);
#pragma unroll
                for (int i = 0; i < 5; ++i) wq5[i] = rpi[(unsigned)(i * PF_THREADS + tid)];  // the next pass's first stage
                constexpr int RG = R < 2 ? R : 2;
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += RG) {
                    float a[8 * RG];
#pragma unroll
                    for (int i2 = 0; i2 < RG; ++i2) {
                        const int r = r0 + i2;
                        const float xa = __uint_as_float(v[r].x), xb2 = __uint_as_float(v[r].z);
                        const float xn0 = xa * nw.x, xn1 = xb2 * nw.y;
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[8 * i2 + i] = pf_dot2(wh4[i], xn0, xn1, 0.f);
                        a[8 * i2 + 4] = fmaf(xb2, xb2, xa * xa);
                        a[8 * i2 + 5] = 0.f; a[8 * i2 + 6] = 0.f; a[8 * i2 + 7] = 0.f;
                    }
                    const float tot = pf_reduce<8 * RG>(a, lane);
                    constexpr int SH = RG == 2 ? 2 : 3;
                    if ((lane & ((1 << SH) - 1)) == 0) red[(par * 8 + wave) * R * FRW + (r0 + ((lane >> SH) >> 3)) * FRW + ((lane >> SH) & 7)] = tot;
                }
                const float psh = fscl[4 * PRS + (tid & 3)];
                __syncthreads();
                for (int idx = tid; idx < 4 * R * PF_REPL; idx += PF_THREADS) {
                    const int m = idx & 3, r = (idx >> 2) % R, rr = idx / (4 * R);
                    if ((run >> r) & 1u) {
                        const float* rp = red + (par * 8) * R * FRW + r * FRW;
                        float t = rp[m], ss = rp[4];
#pragma unroll
                        for (int w = 1; w < 8; ++w) { t += rp[w * R * FRW + m]; ss += rp[w * R * FRW + 4]; }
                        t *= psh;
                        pub(e, rr, r, 4 * b + m, tag0 + e + 1, t * pf_rms_inv(ss, A.eps));
                    }
                }
                par ^= 1;
                PF_TICK(5);
            }
            // ================= decisions: gather the 1024 logits of every row -> rep-pen -> argmax (LAST maximal index) -> next input
            {
                tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                {
                    u32x4 v[R];
                    pf_nap_before_sweep(A.naps[5]);
                    sweep_x(e, v);
                    ++e;
                    PF_TICK(14);
#pragma unroll
                    for (int r = 0; r < R; ++r) {  // (unguarded compute, guarded global writes: see k_slow_rows S1)
                        const bool on = (run >> r) & 1u;
                        float lv0 = __uint_as_float(v[r].x), lv1 = __uint_as_float(v[r].z);
                        const int* ring = s_ring + r * RR;
                        const int* meta = ring + 136;
                        const bool have_prev = (hp >> r) & 1u;
                        // SingleBatchedRepPenProcessor::apply (rep_pen.rs:37-65)
                        int last = -1, dropped = -1, head = 0, len = 0;
                        bool drop = false;
                        if (have_prev) {
                            last = ring[152 + cb + 1];
                            head = (meta[cb * 2] + 16) % 17;
                            len = meta[cb * 2 + 1] + 1;
                            drop = len > 16;
                            if (drop) dropped = ring[cb * 17 + (head + len - 1) % 17];
                        }
                        const float pen = __int_as_float(ring[168 + 16]);
                        const uint32_t mb = s_mb[r * 512 + tid];
                        float m0 = ((mb >> (2 * cb)) & 1u) ? pen : 1.0f, m1 = ((mb >> (2 * cb + 1)) & 1u) ? pen : 1.0f;
                        if (have_prev) {
                            const float o0 = m0, o1 = m1;
                            const int i0 = 2 * tid, i1 = 2 * tid + 1;
                            if (i0 == last) m0 = pen;
                            if (i0 == dropped && m0 == pen) m0 = 1.0f;
                            if (i1 == last) m1 = pen;
                            if (i1 == dropped && m1 == pen) m1 = 1.0f;
                            if (b == r && on) {  // (row r's window state is kept by workgroup r: four rows' guarded stores on one workgroup sat on every decision's critical path)
                                float* mk = A.rp_mask + ((size_t)r * 8 + cb) * 1024;
                                if (m0 != o0) mk[i0] = m0;
                                if (m1 != o1) mk[i1] = m1;
                                if (tid == 0) { A.rp_ring[r * 136 + cb * 17 + head] = last; A.rp_meta[r * 16 + cb * 2] = head; A.rp_meta[r * 16 + cb * 2 + 1] = drop ? 16 : len; }
                            }
                            lv0 = lv0 / m0; lv1 = lv1 / m1;
                        }
                        if (A.cap && b == 0 && on && A.state[r].frame < A.cap_frames)
                            *reinterpret_cast<float2*>(A.cap + (((size_t)r * A.cap_frames + A.state[r].frame) * 9 + 1 + cb) * 2048 + 2 * tid) = make_float2(lv0, lv1);
                        if (SAMPLED) {
                            if (on && (b >> 3) == r) {  // this workgroup draws for row r (see the slow-token draws)
                                int used = 0;
                                const int* cf = ring + 168 + 16;
                                const float lvv[2] = {lv0, lv1};
                                __syncthreads();  // (the sampler's scratch aliases the staging tiles: every wave is past this pass's last S3)
                                const int gi = bsample<PF_THREADS, 2, 8>(lvv, 1024, cf[8], (float)(1.0 / (double)__int_as_float(cf[6])), __int_as_float(cf[7]),
                                                                      (uint32_t)ring[168 + 32 + n_draws[r]], &used, samp);
                                if (tid < 2) pub(e, b & 7, r, tid, tag0 + e + 1, __uint_as_float((uint32_t)gi | ((uint32_t)used << 16)));
                            }
                        } else {
                        float bv = lv0;
                        int bi = 2 * tid;
                        if (!(lv1 < bv)) { bv = lv1; bi = 2 * tid + 1; }
                        float wm; int ci;
                        pr_wave_argmax(bv, bi, wm, ci);
                        if (lane == 0) { amax[((par * R + r) * 8 + wave) * 2] = wm; amax[((par * R + r) * 8 + wave) * 2 + 1] = __int_as_float(ci); }
                        }
                    }
                }
                PF_TICK(15);
                u32x4 dv[R];
                if (SAMPLED) {
                    const u64* bs[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) bs[r] = ebase(e, rep, ((run >> r) & 1u) ? r : first);
                    pf_nap_before_sweep(A.nap_draw);
                    pr_sweep_rows<R>(bs, 0u, tag0 + e + 1, dv, dead, A.ctl);
                    ++e;
                }
                __syncthreads();
                uint32_t codes[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool on = (run >> r) & 1u;
                    float gv = SAMPLED ? 0.f : amax[((par * R + r) * 8) * 2];
                    int gi = SAMPLED ? (int)(dv[r].x & 0xFFFFu) : __float_as_int(amax[((par * R + r) * 8) * 2 + 1]);
                    if (SAMPLED && on) n_draws[r] += (int)(dv[r].x >> 16);
                    if (!SAMPLED) {
#pragma unroll
                    for (int w = 1; w < 8; ++w) {
                        const float v2 = amax[((par * R + r) * 8 + w) * 2];
                        const int i2 = __float_as_int(amax[((par * R + r) * 8 + w) * 2 + 1]);
                        if (v2 > gv || (v2 == gv && i2 > gi)) { gv = v2; gi = i2; }
                    }
                    }
                    codes[r] = (uint32_t)max(gi, 0);
                }
                // hidden_states = fast_embeddings(code) (single_batch.rs:181-183): the next pass's S1 reads it back (same lane).  All rows' table
                // rows are requested before anything else touches memory: inside the row loop each request waited behind the previous row's
                // guarded stores (3.3 us per decision at 4 rows)
                uint32_t ew[R];
                const bool next_tbl = !(SAMPLED && R == 4) && A.qkv0_tbl != nullptr;
                if (cb != 7) {
                    if (next_tbl) {  // the next pass's first layer reads the qkv table; of the embedding row only this workgroup's 4 residual elements are needed
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            pcode[r] = (uint32_t)__builtin_amdgcn_readfirstlane((int)codes[r]);
                            ew[r] = (uint32_t)reinterpret_cast<const uint16_t*>(A.fast_emb)[codes[r] * 1024u + (unsigned)(4 * b + (tid & 3))];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) ew[r] = reinterpret_cast<const uint32_t*>(A.fast_emb)[codes[r] * 512u + (unsigned)tid];
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool on = (run >> r) & 1u;
                    if (A.cap && b == 0 && tid == 0 && on && A.state[r].frame < A.cap_frames)
                        A.cap[(((size_t)r * A.cap_frames + A.state[r].frame) * 9 + 1 + cb) * 2048 + 1024] = (float)codes[r];
                    if (tid == 0) s_ring[r * RR + 168 + 4 + cb] = (int)codes[r];
                }
                if (cb != 7) {
                    if (next_tbl) {  // (read behind S2's block barrier; the last reader of xr -- layer 3's S4 epilogue -- is two barriers back)
                        if (tid < 4) {
#pragma unroll
                            for (int r = 0; r < R; ++r) xr[r * 4 + tid] = __uint_as_float(ew[r] << 16);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) *reinterpret_cast<float2*>(qs + r * 1024 + 2 * tid) = make_float2(bf_lo(ew[r]), bf_hi(ew[r]));
                    }
                }
                par ^= 1;
                PF_TICK(6);
            }
        }
    }
    if (b >= R) return;
    int* n_draws_l = reinterpret_cast<int*>(amax);  // (dead: every decision is taken)
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) n_draws_l[r] = n_draws[r];
    }
    // ---- end of frame: row r on workgroup r (single_batch.rs:185-210 + generate_blocking :250,264-266); every workgroup holds the same
    // per-row decisions in LDS, so the rows' state updates and next-input gathers run side by side instead of row by row on workgroup 0
    __syncthreads();
    {
        const int r = b;
        if ((live >> r) & 1u) {
        const bool eos = !((run >> r) & 1u);
        SeqState* st = A.state + r;
        const int* misc = s_ring + r * RR + 168;
        const uint32_t cur0 = (uint32_t)misc[0];
        if (tid == 0) {
            const int fr = st->frame;
            uint32_t codes[8];
            for (int c = 0; c < 8; ++c) codes[c] = eos ? 0u : (uint32_t)misc[4 + c];
            st->cur[0] = cur0;
            for (int c = 0; c < 8; ++c) st->cur[c + 1] = codes[c];
            if (fr == 0 || !eos) {
                const int o = st->n_out;
                if (o < A.out_cap)
                    for (int c = 0; c < 8; ++c) A.out_codes[((size_t)r * 8 + c) * A.out_cap + o] = codes[c];
                st->n_out = o + 1;
            }
            st->prev[0] = cur0;
            for (int c = 0; c < 8; ++c) st->prev[c + 1] = codes[c];
            st->have_prev = 1;
            st->pos += 1;
            st->frame = fr + 1;
            if (eos || fr + 1 >= A.budget[r]) st->done = 2;
            if (SAMPLED || legacy) A.rng[r].consumed += (unsigned long long)n_draws_l[r];
        }
        // next slow input: embed([slow, c0..c7]) (dual_ar.rs:532-567)
        {
            const float mk = (cur0 >= (uint32_t)misc[20] && cur0 <= (uint32_t)misc[21]) ? 1.f : 0.f;
            const uint32_t* te = reinterpret_cast<const uint32_t*>(A.tok_emb);
            const uint32_t* ce = reinterpret_cast<const uint32_t*>(A.cb_emb);
            const uint32_t w0 = te[(size_t)cur0 * 512 + tid];
            float e0 = 0.f + bf_lo(w0), e1 = 0.f + bf_hi(w0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t code = eos ? 0u : (uint32_t)misc[4 + c];
                const uint32_t wv = ce[((size_t)c * 1024 + code) * 512 + tid];
                e0 += bf_lo(wv) * mk;
                e1 += bf_hi(wv) * mk;
            }
            *reinterpret_cast<float2*>(A.x + (size_t)r * 1024 + 2 * tid) = make_float2(e0, e1);
        }
        }
    }
    if (b != 0) return;
    if (tid == 0) A.ctl[0] = epoch + 1;
    PF_TICK(7);
    if (A.prof && tid == 0) for (int k = 0; k < 16; ++k) A.prof[k] += tk[k];
#undef PF_TICK
}

// ------------------------------------------------------------------------------------------------ host side
size_t rows_slow_edge_bytes(int R) { return (size_t)PF_RING * PF_REPL * R * PS_EDGE_CAP * 8; }

void launch_rows_pack(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, void* wimg, void* himg, hipStream_t st) {
    for (int l = 0; l < n_layer; ++l)
        hipLaunchKernelGGL(k_pr_pack_layer, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, layers[l],
                           reinterpret_cast<unsigned char*>(wimg) + (size_t)l * PF_BLOCKS * PS_LAYER_IMAGE);
    hipLaunchKernelGGL(k_pr_pack_head, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, reinterpret_cast<const u32x4*>(head_w), n_head_rows,
                       reinterpret_cast<unsigned char*>(himg));
    FS_HIP(hipGetLastError());
}

void launch_rows_pack_fp8(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, void* wimg, void* himg, hipStream_t st) {
    for (int l = 0; l < n_layer; ++l)
        hipLaunchKernelGGL(k_pr_pack_layer_fp8, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, layers[l],
                           reinterpret_cast<unsigned char*>(wimg) + (size_t)l * PF_BLOCKS * PS_LAYER_IMAGE);
    hipLaunchKernelGGL(k_pr_pack_head_fp8, dim3(PF_BLOCKS), dim3(PF_THREADS), 0, st, reinterpret_cast<const uint2*>(head_w), n_head_rows,
                       reinterpret_cast<unsigned char*>(himg));
    FS_HIP(hipGetLastError());
}

template <int R>
static void launch_rows_slow_r(const RowsSlowArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slow_rows<R>), hipFuncAttributeMaxDynamicSharedMemorySize, SlowLds<R>::BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_slow_rows<R>, dim3(PF_BLOCKS), dim3(PF_THREADS), SlowLds<R>::BYTES, st, a);
    FS_HIP(hipGetLastError());
}
void launch_rows_pack_rowpairs(const void* fast_pack, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_pr_pack_rowpairs, dim3(PF_BLOCKS, PF_ROW_CHUNKS), dim3(PF_THREADS), 0, st, reinterpret_cast<const u32x4*>(fast_pack),
                       reinterpret_cast<uint32_t*>(out));
    FS_HIP(hipGetLastError());
}
size_t rows_fast_edge_bytes(int R) { return (size_t)PF_RING * PF_REPL * R * PF_EDGE_CAP * 8; }
template <int R, bool SAMPLED>
static void launch_rows_fast_r(const RowsFastArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_rows<R, SAMPLED>), hipFuncAttributeMaxDynamicSharedMemorySize, FastLds<R>::BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_fast_rows<R, SAMPLED>), dim3(PF_BLOCKS), dim3(PF_THREADS), FastLds<R>::BYTES, st, a);
    FS_HIP(hipGetLastError());
}
void launch_rows_fast(const RowsFastArgs& a, int R, bool sampled, hipStream_t st) {
    if (R == 1) { if (sampled) launch_rows_fast_r<1, true>(a, st); else launch_rows_fast_r<1, false>(a, st); }
    else if (R == 2) { if (sampled) launch_rows_fast_r<2, true>(a, st); else launch_rows_fast_r<2, false>(a, st); }
    else if (R == 4) { if (sampled) launch_rows_fast_r<4, true>(a, st); else launch_rows_fast_r<4, false>(a, st); }
    else throw Error("launch_rows_fast: R must be 1, 2 or 4");
}
void launch_rows_slow(const RowsSlowArgs& a, int R, hipStream_t st) {
    FS_REQUIRE(R * a.n_sl <= 16 && a.n_sl >= 1, "rows x attention slices exceed the 256 attention items of a launch");
    if (R == 2) launch_rows_slow_r<2>(a, st);
    else if (R == 4) launch_rows_slow_r<4>(a, st);
    else if (R == 8) launch_rows_slow_r<8>(a, st);
    else throw Error("launch_rows_slow: R must be 2, 4 or 8");
}

}  // namespace fs
