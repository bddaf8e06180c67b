// Persistent fast-decoder kernel (see lm_persist.h): one launch = the 8 codebook passes of one audio frame.
//
// Reference semantics implemented (same arithmetic as the per-node kernels of lm_kernels.hip, other summation order):
//   forward_generate_fast  fish_speech_core/lib/lm/dual_ar.rs:638-673  (4 blocks at RoPE position = codebook index, own KV
//                          cache cleared every frame, fast_norm, fast_output)
//   TransformerBlock / Attention / FeedForward  dual_ar.rs:160-165,281-384,429-440 (interleaved RoPE, scale on K, GQA by head index)
//   frame loop             generate/single_batch.rs:146-210 (8 codebooks, rep-pen keyed on the previous frame, EOS -> zeros)
//   rep-pen                sampling/rep_pen.rs:37-65 (window 16, never-incremented count)
//   embed                  dual_ar.rs:532-567
//
// Work split.  Workgroup b owns output rows [5b, 5b+5) of Wqkv, [4b, 4b+4) of Wo / W2 / fast_output and the 16 SwiGLU pairs
// [16b, 16b+16) of W13 of EVERY fast layer; inside the workgroup the reduction dimension is split over all 512 lanes (lane t owns
// input elements 2t, 2t+1; for W2: 1024q + 2t, +1, q = 0..3), so a lane's activation slice arrives straight from its own sweep
// loads and never passes through LDS.  Per-row partial sums are reduced with halving trees (v_permlane32/16_swap + DPP: 70
// instructions for 32 rows instead of 32 x 12), the 8 wave partials meet in LDS, `rows` lanes add them in wave order, apply the
// epilogue (residual / SwiGLU) and publish to the 8 replicas of the edge buffer.
//
// Edge protocol (MI355X guide, Guideline 16 R2): a granule = one aligned 8-byte {f32 value, tag} written by ONE relaxed
// agent-scope (sc1, write-through) store; tag = launch epoch * 256 + edge index + 1, unique over the life of the handle, so buffers
// are never re-armed; consumers sweep with 16-byte sc1 loads (2 granules) and retry only the units whose tags are not there yet.
// Edge e uses buffer e % 4: a producer can publish edge e only after it has consumed edge e-1, which every workgroup published
// after consuming edge e-2, so nobody still reads the buffer of edge e-4 (nor e-2).  Every spin is bounded; a timeout sets
// ctl[1] and lets the thread run on (garbage tokens, no hang), the host turns it into an error.
//
// Round 3 additions: the slow-token decision of the frame is taken in this kernel's prologue (A.slow_logits), so a
// frame is two launches; k_fast_persist<true> takes every decision with the block-parallel sampler of lm_bsample_dev.h (token-exact
// top-k / top-p / WeightedIndex chain); FS_FP8 handles bring a bf16-widened e4m3 image + row scales (A.scales); the W13 stage runs on the
// matrix cores (resident A fragments, the activation split into three bf16 terms as B columns); every stage sleeps A.naps[kind] x 64
// clocks before its first sweep (pf_nap_before_sweep: early polling floods the memory side in front of the granules everybody waits for).
//
// State ordering inside a launch: per-frame inputs (SeqState, rep-pen ring / mask, sampler config) are read by every workgroup
// BEFORE its first publish; only workgroup 0 writes them, and it cannot reach its first write before it has consumed a
// full edge, i.e. before every workgroup has completed those reads.
#include "lm_persist.h"

#include <hip/hip_runtime.h>

#include "fs_common.h"

namespace fs {

namespace {

#include "lm_persist_dev.h"
#include "lm_bsample_dev.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// LDS carve (bytes).  Everything lives in ONE dynamic array (guide: Guideline 17)
constexpr int L_W2 = 0;                                   // [16 chunks][512 lanes] x 16 B
constexpr int L_KC = L_W2 + PF_LDS_CHUNKS * PF_THREADS * 16;  // K cache: [4 layers][8 pos][64] bf16 pairs
constexpr int L_VC = L_KC + PF_LAYERS * 8 * 64 * 4;
constexpr int L_QS = L_VC + PF_LAYERS * 8 * 64 * 4;       // rope'd q, f32 [1024]
constexpr int L_XS = L_QS + 4096;                          // residual stream copy, f32 [1024]
constexpr int PF_RED = 40;                                 // row partials per wave: 32 rows + sum of squares
constexpr int L_RED = L_XS + 4096;                         // [2][8 waves][PF_RED]
constexpr int L_SC = L_RED + 2 * 8 * PF_RED * 4;                   // [16 heads][8 pos] attention scores
constexpr int L_AMAX = L_SC + 16 * 8 * 4;                  // [2][8 waves] {value, index}
constexpr int L_ROPE = L_AMAX + 2 * 8 * 8;                 // cos [8][32], sin [8][32]
constexpr int L_RING = L_ROPE + 2 * 8 * 32 * 4;            // rep-pen ring [8][17], meta [8][2], prev [16], misc [16]
constexpr int L_WORDS = L_RING + (8 * 17 + 8 * 2 + 16 + 16) * 4;  // StdRng output words of this frame's draws [8] (sampled requests)
constexpr int L_SCL = L_WORDS + 16 * 4;  // FS_FP8 handles: this workgroup's row scales [PF_SCL]
constexpr int L_XR = L_SCL + PF_SCL * 4;  // this workgroup's own 4 elements of the residual stream (all a W2 / Wo epilogue reads back)
constexpr int L_END = L_XR + 16;
// S3's MFMA B operands: x . g split into three bf16 arrays [3][1024] + one zero slot, over the q scratch and the old residual copy
constexpr int L_XB = L_QS;
static_assert(L_XB + 3 * 2048 + 8 * 16 <= L_RED, "the bf16 activation arrays (+ a zero slot per wave) must not reach the row partials");
static_assert(L_END <= 160 * 1024, "LDS budget");
// the sampler's scratch (lm_bsample_dev.h) aliases q / residual copy / row partials / scores / argmax slots: all dead during a decision
static_assert(L_QS % 16 == 0 && L_QS + (int)sizeof(BSampLds) <= L_ROPE, "sampler scratch must fit the stage scratch it aliases");

}  // namespace

// ------------------------------------------------------------------------------------------------ weight image
// chunk c < 10: dwords 4c .. 4c+3 of the lane's row-pair image; dword d < 36: layer l = d / 9, i = d % 9: i < 5: Wqkv row 5b + i, else Wo row
//   4b + i - 5; d >= 36: fast_output row 4b + d - 36.  Each dword = elements (2t, 2t+1) of that row.
// chunk 10 + 8l + u: one MFMA A fragment (v_mfma_f32_16x16x32_bf16) of layer l's W13: row tile rt = u / 4 (interleaved rows 32b + 16rt + (lane & 15)),
//   k-step j = u % 4 of the wave's K range: elements 128 wave + 32 j + 8 (lane >> 4) .. + 8 of that row.
// chunk 42 + 4l + q: {W2_l row 4b + r, elements (1024q + 2t, +1)}, r = 0..3
// FP8: the weights are e4m3 bytes; the image holds them widened to bf16 (exact: e4m3 is a subset of bf16), so the frame kernel is the same
// and the per-row f32 scales multiply the K-summed row results in its publishing lanes (k_pf_pack_scales).
template <bool FP8>
__global__ __launch_bounds__(PF_THREADS) void k_pf_pack(LayerW w0, LayerW w1, LayerW w2, LayerW w3, const void* __restrict__ head,
                                                        u32x4* __restrict__ pack) {
    const int b = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    const LayerW* ws[4] = {&w0, &w1, &w2, &w3};
    auto pair = [](const void* W, size_t idx) -> uint32_t {  // elements (2 idx, 2 idx + 1) of the flattened matrix as a bf16 pair
        if constexpr (FP8) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 f = __builtin_amdgcn_cvt_pk_f32_fp8((uint32_t)reinterpret_cast<const uint16_t*>(W)[idx], false);
            return (__float_as_uint(f.x) >> 16) | (__float_as_uint(f.y) & 0xFFFF0000u);
        } else {
            return reinterpret_cast<const uint32_t*>(W)[idx];
        }
    };
    u32x4 out;
    if (c < PF_ROW_CHUNKS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = 4 * c + k;
            uint32_t v;
            if (d >= PF_LAYERS * 9) v = pair(head, (size_t)(4 * b + d - PF_LAYERS * 9) * 512 + t);
            else {
                const LayerW& w = *ws[d / 9];
                const int i = d % 9;
                if (i < 5) v = pair(w.wqkv, (size_t)(5 * b + i) * 512 + t);
                else v = pair(w.wo, (size_t)(4 * b + i - 5) * 512 + t);
            }
            out[k] = v;
        }
    } else if (c < PF_REG_CHUNKS) {
        const int l = (c - PF_ROW_CHUNKS) / 8, u2 = (c - PF_ROW_CHUNKS) % 8, rt = u2 >> 2, j = u2 & 3, wv = t >> 6, ln = t & 63;
        const size_t row = (size_t)(32 * b + 16 * rt + (ln & 15));
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = pair(ws[l]->w13, row * 512 + (size_t)(64 * wv + 16 * j + 4 * (ln >> 4) + k));
    } else {
        const int l = (c - PF_REG_CHUNKS) / 4, q = (c - PF_REG_CHUNKS) % 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = pair(ws[l]->w2, (size_t)(4 * b + r) * 2048 + 512 * q + t);
    }
    pack[((size_t)b * PF_CHUNKS + c) * PF_THREADS + t] = out;
}
// row scales of an FS_FP8 handle, per workgroup: layer l at [48 l]: Wqkv 5 at +0, Wo 4 at +8, W13 32 at +12, W2 4 at +44; fast_output 4 at [192]
__global__ void k_pf_pack_scales(LayerW w0, LayerW w1, LayerW w2, LayerW w3, const float* __restrict__ head_s, float* __restrict__ out) {
    const int b = blockIdx.x, t = threadIdx.x;
    const LayerW* ws[4] = {&w0, &w1, &w2, &w3};
    if (t >= PF_SCL) return;
    float sc = 0.f;
    if (t >= 192) { if (t < 196) sc = head_s[4 * b + t - 192]; }
    else {
        const LayerW& w = *ws[t / 48];
        const int i = t % 48;
        if (i < 5) sc = w.s_qkv[5 * b + i];
        else if (i >= 8 && i < 12) sc = w.s_o[4 * b + i - 8];
        else if (i >= 12 && i < 44) sc = w.s_13[32 * b + i - 12];
        else if (i >= 44) sc = w.s_2[4 * b + i - 44];
    }
    out[(size_t)b * PF_SCL + t] = sc;
}

// ------------------------------------------------------------------------------------------------ layer-0 qkv table (round 6)
// Codebook passes 1..7 feed the fast decoder fast_embeddings[code] (single_batch.rs:181-183), so the first fast layer's
// attention_norm + Wqkv of those passes is a pure function of the 1024 code ids: tbl[code][1280] = what stage S1 of layer 0 would publish
// (pre-RoPE q | k | v; the position's rotation is applied by the consumer as before).  Built once per weight load FROM THE PACKED IMAGE
// with S1's own instruction sequence (same per-lane products, same halving tree, same wave-partial order, same v_rsq), so a table entry
// carries the bits the stage would have published.  The frame kernel then reads 3 float2 per lane behind the decision instead of
// computing, publishing and sweeping an edge: 7 of the 137 stages of a frame disappear.
__global__ __launch_bounds__(PF_THREADS) void k_pf_qkv0_table(const void* __restrict__ wpack, const float* __restrict__ scales,
                                                              const float* __restrict__ norm0, const void* __restrict__ fast_emb, float eps,
                                                              float* __restrict__ tbl) {
    __shared__ float red[2 * 8 * PF_RED];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack) + (size_t)b * PF_CHUNKS * PF_THREADS + tid;
    const u32x4 c0 = wp[0], c1 = wp[PF_THREADS];
    const uint32_t wl[5] = {c0.x, c0.y, c0.z, c0.w, c1.x};  // layer 0: dwords 0..4 = Wqkv rows 5b .. 5b+4, elements (2 tid, 2 tid + 1)
    const float2 nw = *reinterpret_cast<const float2*>(norm0 + 2 * tid);
    int par = 0;
    for (int code = blockIdx.y; code < 1024; code += gridDim.y) {
        const uint32_t ew = reinterpret_cast<const uint32_t*>(fast_emb)[(size_t)code * 512 + tid];
        const float x0 = bf_lo(ew), x1 = bf_hi(ew);
        const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;
        float a8[8];
#pragma unroll
        for (int r = 0; r < 5; ++r) a8[r] = pf_dot2(wl[r], xn0, xn1, 0.f);
        a8[5] = fmaf(x1, x1, fmaf(x0, x0, 0.f));
        a8[6] = 0.f; a8[7] = 0.f;
        const float r8 = pf_reduce<8>(a8, lane);
        if ((lane & 7) == 0) red[(par * 8 + wave) * PF_RED + (lane >> 3)] = r8;
        __syncthreads();
        if (wave == 0) {
            const int r = min(lane & 15, 5), k = lane >> 4;
            const float* rp = red + (par * 8 + k) * PF_RED;
            float t = pf_sum_rows(rp[r] + rp[4 * PF_RED + r]);
            const float tot = pf_sum_rows(rp[5] + rp[4 * PF_RED + 5]);
            if (scales) t *= scales[(size_t)b * PF_SCL + min(r, 4)];
            if (lane < 5) tbl[(size_t)code * 1280 + 5 * b + r] = t * pf_rms_inv(tot, eps);
        }
        par ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------ the frame kernel
// SAMPLED: temp > 0 with 0 < top_k <= 256 (the server default, server/lib/utils/load.rs:116-125): the decision is the block-parallel
// top-k / top-p / WeightedIndex sampler of lm_bsample_dev.h, run redundantly by every workgroup on the same logits and the same StdRng
// word, so every workgroup still knows the next input without another edge.  A separate instantiation: the greedy kernel keeps its
// register allocation.
template <bool SAMPLED>
__global__ __launch_bounds__(PF_THREADS) void k_fast_persist(FastPersistArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // launch-boundary stamps (FISHRT_PERSIST_PROF; s_memrealtime is one clock for all kernels): prof[16] = when this launch's workgroup 0 finished,
    // [17] += entry - the slow kernel's finish, [18] += entry -> first stage timer (per-frame state + the slow-token decision), [19] += last timer -> finish
    const unsigned long long t_entry = A.prof ? wall_clock64() : 0;
    u32x4* w2s = reinterpret_cast<u32x4*>(smem + L_W2);
    uint32_t* kc = reinterpret_cast<uint32_t*>(smem + L_KC);
    uint32_t* vc = reinterpret_cast<uint32_t*>(smem + L_VC);
    float* qs = reinterpret_cast<float*>(smem + L_QS);
    float* xr = reinterpret_cast<float*>(smem + L_XR);
    float* red = reinterpret_cast<float*>(smem + L_RED);
    float* sc = reinterpret_cast<float*>(smem + L_SC);
    float* amax = reinterpret_cast<float*>(smem + L_AMAX);
    float* rope_c = reinterpret_cast<float*>(smem + L_ROPE);
    float* rope_s = rope_c + 8 * 32;
    int* s_ring = reinterpret_cast<int*>(smem + L_RING);
    int* s_meta = s_ring + 8 * 17;
    uint32_t* s_prev = reinterpret_cast<uint32_t*>(s_meta + 8 * 2);
    uint32_t* s_misc = s_prev + 16;  // [0] cur0, [1] have_prev, [2] done, [3] epoch, [4..11] codes of this frame
    uint32_t* s_words = reinterpret_cast<uint32_t*>(smem + L_WORDS);
    float* s_scl = reinterpret_cast<float*>(smem + L_SCL);
    const bool fp8 = A.scales != nullptr;  // FS_FP8 handle: bf16-widened e4m3 image + row scales (wave-uniform)
    BSampLds& samp = *reinterpret_cast<BSampLds*>(smem + L_QS);

    const int tid_k = threadIdx.x, b = blockIdx.x;
    int tid = tid_k, lane = tid & 63, wave = tid >> 6;
    const int rep = b & (PF_REPL - 1);
    u64* const edges = A.edges;
    const u64* const my_edges = A.edges + (size_t)rep * PF_EDGE_CAP;  // + (e & 3) * PF_REPL * PF_EDGE_CAP per edge
    const SampleCfg cfg = *A.cfg;

    // ---- per-frame inputs (see "State ordering" above)
    if (tid < 8 * 17) s_ring[tid] = A.rp.ring[tid];
    else if (tid < 8 * 17 + 16) s_meta[tid - 8 * 17] = A.rp.ring_meta[tid - 8 * 17];
    else if (tid < 8 * 17 + 32) s_prev[tid - 8 * 17 - 16] = A.state->prev[tid - 8 * 17 - 16];
    else if (tid == 200) s_misc[0] = A.state->cur[0];
    else if (tid == 201) s_misc[1] = (uint32_t)A.state->have_prev;
    else if (tid == 202) s_misc[2] = (uint32_t)A.state->done;
    else if (tid == 203) s_misc[3] = A.ctl[0];
    if (fp8 && tid < PF_SCL) s_scl[tid] = A.scales[(size_t)b * PF_SCL + tid];
    if (tid >= 256) {  // RoPE rows 0..7
        const int i = tid - 256;
        rope_c[i] = A.cos_t[i];
        rope_s[i] = A.sin_t[i];
    }
    const unsigned long long consumed0 = (SAMPLED || cfg.legacy) ? A.rng->consumed : 0ull;  // (a per-frame input: read before the first publish)
    // this frame's StdRng words (slow draw + 8 codebook draws): ~2.4 us of dependent integer work, nine lanes of the last wave
    if (SAMPLED && tid >= PF_THREADS - 64 && tid < PF_THREADS - 64 + 9)
        s_words[tid - (PF_THREADS - 64)] = chacha12_word(A.rng->key, consumed0 + (unsigned)(tid - (PF_THREADS - 64)));
    // Fish <= 1.4: the slow token is a 2-way {pad, im_end} draw whatever the temperature (sampling/mod.rs:8-26): the greedy kernel needs that one word
    if (!SAMPLED && cfg.legacy && A.slow_logits && tid == PF_THREADS - 64) s_words[0] = chacha12_word(A.rng->key, A.rng->consumed);
    __syncthreads();
    uint32_t cur0 = s_misc[0];
    const bool have_prev = s_misc[1] != 0;
    int done_in = (int)s_misc[2];
    const unsigned epoch = s_misc[3];
    if (done_in == 2) return;  // generator already terminated: replays leave position / outputs untouched
    const bool cfg_ok = SAMPLED ? (cfg.temp > 0.f && cfg.top_k > 0 && cfg.top_k <= BS_MAXK) : cfg.temp == 0.f;
    if (!cfg_ok && tid == 0 && b == 0) atomicAdd(A.ctl + 2, 1u);  // the host picks the instantiation by the sampling configuration
    int n_draws = 0;
    // ---- the slow-token decision of this frame, folded in (A.slow_logits != null; otherwise k_sample_slow ran in front of this launch):
    // logits over [im_end, V) (constrain_probs_to_audio, utils.rs:13-16) -> sample -> token = index + im_end (rescale_semantic_tokens,
    // utils.rs:45-46), single_batch.rs:102-144.  Every workgroup decides redundantly on the same logits (and the same StdRng word), so
    // the token -- needed for the <|im_end|> test here and for the next input's embedding at the end -- needs no edge.
    if (A.slow_logits && cfg.legacy) {
        // legacy_softmax_sample (sampling/mod.rs:8-26; single_batch.rs:104-124): P(pad) = softmax([pad, eos])[0]; u ~ U[0,1) = (next_u32 >> 8) * 2^-24
        // (rand Standard<f32>), temperature ignored.  The reference draws from an unseeded thread_rng; the draw comes from the request's seeded
        // StdRng stream here (k_sample_slow does the same on the per-node path).  Every workgroup decides redundantly on the same two logits
        // and the same word.
        const float pad = A.slow_logits[0], eosl = A.slow_logits[1], m = fmaxf(pad, eosl);
        const float e_pad = expf(pad - m), e_eos = expf(eosl - m);
        const float p_pad = e_pad / (e_pad + e_eos);
        const float u = (float)(s_words[0] >> 8) * (1.0f / 16777216.0f);
        const bool is_pad = u < p_pad || cfg.ignore_eos;
        n_draws += 1;
        if (b == 0) {
            float* hid = A.hid_slot ? *A.hid_slot : nullptr;
            if (hid) *reinterpret_cast<float2*>(hid + (size_t)A.state->frame * 1024 + 2 * tid) = *reinterpret_cast<const float2*>(A.xf + 2 * tid);
            if (A.cap && A.state->frame < A.cap_frames && tid == 0) {  // (fs_lm_debug_capture: the two logits, the uniform draw, the pick)
                float* cap = A.cap + (size_t)A.state->frame * 9 * 2048;
                cap[0] = pad; cap[1] = eosl; cap[2] = u; cap[2047] = is_pad ? 0.f : 1.f;
            }
        }
        cur0 = is_pad ? cfg.pad_id : cfg.im_end_id;
        if (cur0 == cfg.im_end_id) done_in = 1;
    } else if (A.slow_logits) {
        const int n = A.n_slow;
        float lv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = 4 * tid + s;
            lv[s] = i < n ? A.slow_logits[i] : -INFINITY;
            if (i == 0 && cfg.ignore_eos) lv[s] = -INFINITY;
        }
        if (b == 0) {  // hidden_states of this iteration (generate_blocking_with_hidden, single_batch.rs:250,264-266)
            float* hid = A.hid_slot ? *A.hid_slot : nullptr;
            if (hid) *reinterpret_cast<float2*>(hid + (size_t)A.state->frame * 1024 + 2 * tid) = *reinterpret_cast<const float2*>(A.xf + 2 * tid);
        }
        float* cap = (A.cap && b == 0 && A.state->frame < A.cap_frames) ? A.cap + (size_t)A.state->frame * 9 * 2048 : nullptr;
        if (cap) *reinterpret_cast<float4*>(cap + 4 * tid) = make_float4(lv[0], lv[1], lv[2], lv[3]);  // (fs_lm_debug_capture)
        int idx;
        if (SAMPLED) {
            int used = 0;
            idx = bsample<PF_THREADS, 4, 8>(lv, n, cfg.top_k, (float)(1.0 / (double)cfg.temp), cfg.top_p, s_words[0], &used, samp);
            n_draws += used;
        } else {  // host ArgMax rule: the LAST maximal index (per thread ascending, then value / index maxima)
            float bv = lv[0];
            int bi = 4 * tid;
#pragma unroll
            for (int s = 1; s < 4; ++s) if (!(lv[s] < bv)) { bv = lv[s]; bi = 4 * tid + s; }
            if (bi >= n) bi = -1;
            float wm = bv;
            wm = fmaxf(wm, pf_dpp<PF_XOR1>(wm)); wm = fmaxf(wm, pf_dpp<PF_XOR2>(wm));
            wm = fmaxf(wm, pf_dpp<PF_HALF_MIRROR>(wm)); wm = fmaxf(wm, pf_dpp<PF_MIRROR>(wm));
            wm = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 31))),
                       fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 47)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 63))));
            int ci = (bv == wm) ? bi : -1;
            ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_XOR1, 0xF, 0xF, false)); ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_XOR2, 0xF, 0xF, false));
            ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_HALF_MIRROR, 0xF, 0xF, false)); ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_MIRROR, 0xF, 0xF, false));
            ci = max(max(__builtin_amdgcn_readlane(ci, 15), __builtin_amdgcn_readlane(ci, 31)), max(__builtin_amdgcn_readlane(ci, 47), __builtin_amdgcn_readlane(ci, 63)));
            if (lane == 0) { amax[wave * 2] = wm; amax[wave * 2 + 1] = __int_as_float(ci); }
            __syncthreads();
            float gv = amax[0];
            idx = __float_as_int(amax[1]);
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float v2 = amax[w * 2];
                const int i2 = __float_as_int(amax[w * 2 + 1]);
                if (v2 > gv || (v2 == gv && i2 > idx)) { gv = v2; idx = i2; }
            }
            __syncthreads();  // amax is written again by the first codebook decision
        }
        if (cap && tid == 0) cap[2047] = (float)idx;
        cur0 = audio_tok(cfg, max(idx, 0));
        if (cur0 == cfg.im_end_id) done_in = 1;  // terminated by THIS frame (workgroup 0 records token and flag at the end of the frame)
    }
    const bool eos = cur0 == cfg.im_end_id;  // single_batch.rs:153-156: push zeros, skip the fast decoder
    if (eos && b != 0) return;

    bool dead = false;
    float x0 = 0.f, x1 = 0.f;
    unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = A.prof ? wall_clock64() : 0;
    const unsigned long long t_timers = t_last;
#define PF_TICK(k) do { if (A.prof) { const unsigned long long n_ = wall_clock64(); tk[k] += n_ - t_last; t_last = n_; } } while (0)
    if (!eos) {
        // ---- resident weights: 42 register chunks + 16 LDS chunks per lane
        const u32x4* wp = reinterpret_cast<const u32x4*>(A.wpack) + (size_t)b * PF_CHUNKS * PF_THREADS + tid;
        uint32_t wr[4 * PF_ROW_CHUNKS];        // Wqkv / Wo / fast_output row pairs
        u32x4 w13v[PF_LAYERS][8];              // W13 as MFMA A fragments: [layer][row tile * 4 + k-step]
#pragma unroll
        for (int c = 0; c < PF_ROW_CHUNKS; ++c) {
            const u32x4 t4 = wp[(size_t)c * PF_THREADS];
            wr[4 * c] = t4.x; wr[4 * c + 1] = t4.y; wr[4 * c + 2] = t4.z; wr[4 * c + 3] = t4.w;
        }
#pragma unroll
        for (int c = PF_ROW_CHUNKS; c < PF_REG_CHUNKS; ++c) w13v[(c - PF_ROW_CHUNKS) / 8][(c - PF_ROW_CHUNKS) % 8] = wp[(size_t)c * PF_THREADS];
#pragma unroll
        for (int c = 0; c < PF_LDS_CHUNKS; ++c) w2s[c * PF_THREADS + tid] = wp[(size_t)(PF_REG_CHUNKS + c) * PF_THREADS];
        // this lane's slice of the nine RMSNorm weight vectors: resident in the greedy kernel; the sampled kernel needs those 18 registers
        // for the sampler's temporaries and re-reads the (L2-resident) pair in front of each stage's sweep instead
        float2 nwr[2 * PF_LAYERS + 1];
        if (!SAMPLED) {
#pragma unroll
            for (int i = 0; i < 2 * PF_LAYERS + 1; ++i) nwr[i] = *reinterpret_cast<const float2*>(A.norms[i] + 2 * tid);
        }
        auto norm_w = [&](int i, int t) { return SAMPLED ? *reinterpret_cast<const float2*>(A.norms[i] + 2 * t) : nwr[i]; };
        // repetition-penalty mask of this lane's two candidates of every codebook: bit 2 cb + k <=> mask[cb][2 tid + k] != 1
        uint32_t mbits = 0;
#pragma unroll
        for (int cbi = 0; cbi < 8; ++cbi) {
            const float2 m2 = *reinterpret_cast<const float2*>(A.rp.mask + (size_t)cbi * 1024 + 2 * tid);
            mbits |= (m2.x != 1.0f ? 1u : 0u) << (2 * cbi);
            mbits |= (m2.y != 1.0f ? 1u : 0u) << (2 * cbi + 1);
        }
        {
            const float2 xin = *reinterpret_cast<const float2*>(A.xf + 2 * tid);
            x0 = xin.x; x1 = xin.y;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every per-frame input has arrived before this workgroup's first publish
        // pin the weight image: keeps the optimiser from re-loading (sinking) any of it into the loop
#pragma unroll
        for (int d = 0; d < 4 * PF_ROW_CHUNKS; ++d) asm volatile("" : "+v"(wr[d]));
#pragma unroll
        for (int l = 0; l < PF_LAYERS; ++l)
#pragma unroll
            for (int u2 = 0; u2 < 8; ++u2) asm volatile("" : "+v"(w13v[l][u2]));

        PF_TICK(0);
        unsigned e = 0;                       // edge counter of this launch
        const unsigned tag0 = epoch * 256u;   // tag of edge e = tag0 + e + 1
        int par = 0;                          // parity of the double-buffered LDS partials
        float2 tq = make_float2(0.f, 0.f), tk2 = tq, tv2 = tq;  // qkv table entries of the code picked by the previous decision (this lane's S2 units)
        float xres_tbl = 0.f;
#pragma unroll 1
        for (int cb = 0; cb < 8; ++cb) {
            const int T = cb + 1;
#pragma unroll
            for (int l = 0; l < PF_LAYERS; ++l) {
                const uint32_t* wl = wr + 9 * l;
                // ================= S1: (gather x) -> RMSNorm -> Wqkv rows -> publish 5 values
                // (layer 0 of the passes 1..7 with the qkv table: the stage does not exist -- S2 takes q / k / v of the new token from the table row
                // of the code the previous decision picked)
                const bool from_tbl = l == 0 && cb > 0 && A.qkv0_tbl != nullptr;
                if (!from_tbl) {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    const float2 nw = norm_w(2 * l, tid);
                    if (l > 0) {
                        u32x4 v;
                        pf_nap_before_sweep(A.naps[0]);
                        pf_sweep1(my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP, tid, tag0 + e + 1, v, dead, A.ctl);
                        x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
                        ++e;
                        PF_TICK(9);
                    }
                    if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + 2 * tid - 4 * b) = make_float2(x0, x1);
                    // RMSNorm folded through the GEMV: W . ((x / d) * g) = (W . (x * g)) / d -- the row sums and sum(x^2) go through ONE
                    // reduction (no separate norm barrier); the publishing lanes divide by d
                    const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;
                    float a8[8];
#pragma unroll
                    for (int r = 0; r < 5; ++r) a8[r] = pf_dot2(wl[r], xn0, xn1, 0.f);
                    a8[5] = fmaf(x1, x1, fmaf(x0, x0, 0.f));
                    a8[6] = 0.f; a8[7] = 0.f;
                    const float r8 = pf_reduce<8>(a8, lane);
                    if ((lane & 7) == 0) red[(par * 8 + wave) * PF_RED + (lane >> 3)] = r8;
                    __syncthreads();
                    if (wave == 0) {  // publishing wave (pf_sum_rows): lane (row r, lane row k) -> replicas k, k + 4
                        const int r = min(lane & 15, 5), k = lane >> 4;
                        const float* rp = red + (par * 8 + k) * PF_RED;
                        float t = pf_sum_rows(rp[r] + rp[4 * PF_RED + r]);
                        const float tot = pf_sum_rows(rp[5] + rp[4 * PF_RED + 5]);
                        if (fp8) t *= s_scl[48 * l + min(r, 4)];
                        if ((lane & 15) < 5) {
                            const float val = t * pf_rms_inv(tot, A.eps);
                            pf_publish(edges, e, k, 5 * b + r, tag0 + e + 1, val);
                            pf_publish(edges, e, k + 4, 5 * b + r, tag0 + e + 1, val);
                        }
                    }
                    par ^= 1;
                    PF_TICK(1);
                }
                // ================= S2: gather qkv -> RoPE, KV append, attention over T <= 8 tokens -> Wo rows + residual -> publish 4
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    // Attention of one head on the 32 lanes of a half-wave, q / k / v of the NEW token straight from the edge (every lane loads its
                    // head's two q dims and the matching two dims of its kv head: three units per lane) and the earlier positions' K / V from LDS,
                    // requested before the sweep.  Scores are 32-lane sums left in every lane (permlane16_swap + DPP), softmax and the value sum
                    // run in registers: no q round trip through LDS, no block barrier in front of the attention, no score table -- the stage was a
                    // chain of LDS round trips and barriers, not arithmetic.  Heads 0 and 8 append the token's K / V rows for the later passes.
                    const int h = tid >> 5, g = h >> 3, j = tid & 31;
                    uint32_t kw[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) kw[t] = kc[(l * 8 + t) * 64 + g * 32 + j];  // (positions >= cb: stale words, not used)
                    u32x4 vq, vk, vv;
                    if (from_tbl) {
                        vq.x = __float_as_uint(tq.x); vq.z = __float_as_uint(tq.y);
                        vk.x = __float_as_uint(tk2.x); vk.z = __float_as_uint(tk2.y);
                        vv.x = __float_as_uint(tv2.x); vv.z = __float_as_uint(tv2.y);
                    } else {
                        const u64* eb = my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP;
                        pf_nap_before_sweep(A.naps[1]);
                        pf_sweep3(eb, tid, 512 + g * 32 + j, 576 + g * 32 + j, tag0 + e + 1, vq, vk, vv, dead, A.ctl);
                        ++e;
                    }
                    PF_TICK(10);
                    const float c = rope_c[cb * 32 + j], s = rope_s[cb * 32 + j];
                    // q pair (dual_ar.rs:246-247), 1 / sqrt(64) folded in (a power of two: exact; the reference scales K, dual_ar.rs:260)
                    const float qa = __uint_as_float(vq.x), qb = __uint_as_float(vq.z);
                    const float q0 = (qa * c - qb * s) * 0.125f, q1 = (qa * s + qb * c) * 0.125f;
                    const float ka = __uint_as_float(vk.x), kb = __uint_as_float(vk.z);
                    const uint32_t knew = f32_to_bf16_rne(ka * c - kb * s) | (f32_to_bf16_rne(ka * s + kb * c) << 16);  // bf16: the cache dtype
                    const uint32_t vnew = f32_to_bf16_rne(__uint_as_float(vv.x)) | (f32_to_bf16_rne(__uint_as_float(vv.z)) << 16);
                    if ((h & 7) == 0) {
                        kc[(l * 8 + cb) * 64 + g * 32 + j] = knew;
                        vc[(l * 8 + cb) * 64 + g * 32 + j] = vnew;
                        // fs_lm_debug_capture: the K / V rows this pass appended (raw bf16 pairs: K [64], V [64] per layer) behind the pass's logits,
                        // so that a test can make the oracle attend over exactly these rows (tests/test_kv_forced_gpu.py)
                        if (A.cap && b == 0 && A.slow_logits && A.state->frame < A.cap_frames) {
                            uint32_t* cw = reinterpret_cast<uint32_t*>(A.cap + ((size_t)A.state->frame * 9 + 1 + cb) * 2048 + 1025) + l * 128 + g * 32 + j;
                            cw[0] = knew; cw[64] = vnew;
                        }
                    }
                    float at0, at1;
                    {
                        float scr[8];
                        float mn = -1e30f;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (t < T) {
                                const uint32_t kk = (t == cb || t == 7) ? knew : kw[t < 7 ? t : 0];
                                scr[t] = pf_allsum32(fmaf(q1, bf_hi(kk), q0 * bf_lo(kk)));
                                mn = fmaxf(mn, scr[t]);
                            }
                        float L = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (t < T) {
                                const float pr = __expf(scr[t] - mn);
                                L += pr;
                                const uint32_t vw = (t == cb) ? vnew : vc[(l * 8 + t) * 64 + g * 32 + j];
                                O0 = fmaf(pr, bf_lo(vw), O0);
                                O1 = fmaf(pr, bf_hi(vw), O1);
                            }
                        const float inv = 1.f / L;
                        at0 = O0 * inv; at1 = O1 * inv;
                    }
                    float a4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a4[r] = pf_dot2(wl[5 + r], at0, at1, 0.f);
                    const float r4 = pf_reduce<4>(a4, lane);
                    if ((lane & 15) == 0) red[(par * 8 + wave) * PF_RED + (lane >> 4)] = r4;
                    float xres = 0.f;
                    if (from_tbl) xres = xres_tbl;  // (this workgroup's four residual elements = fast_embeddings[code][4b ..], requested behind the decision)
                    else if (tid < 64) xres = xr[min(tid & 15, 3)];
                    __syncthreads();
                    if (wave == 0) {
                        const int r = min(lane & 15, 3), k = lane >> 4;
                        const float* rp = red + (par * 8 + k) * PF_RED;
                        float t = pf_sum_rows(rp[r] + rp[4 * PF_RED + r]);
                        if (fp8) t *= s_scl[48 * l + 8 + r];
                        if ((lane & 15) < 4) {
                            pf_publish(edges, e, k, 4 * b + r, tag0 + e + 1, xres + t);
                            pf_publish(edges, e, k + 4, 4 * b + r, tag0 + e + 1, xres + t);
                        }
                    }
                    par ^= 1;
                    PF_TICK(2);
                }
                // ================= S3: gather h -> RMSNorm -> 16 SwiGLU pairs of W13 -> publish 16 activations
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    const float2 nw = norm_w(2 * l + 1, tid);
                    u32x4 v;
                    pf_nap_before_sweep(A.naps[2]);
                    pf_sweep1(my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP, tid, tag0 + e + 1, v, dead, A.ctl);
                    ++e;
                    PF_TICK(11);
                    x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
                    if ((unsigned)(2 * tid - 4 * b) < 4u) *reinterpret_cast<float2*>(xr + 2 * tid - 4 * b) = make_float2(x0, x1);
                    const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;  // RMSNorm folded through the GEMV (see S1)
                    const float ssw = pf_wave_sum(fmaf(x1, x1, fmaf(x0, x0, 0.f)));
                    if (lane == 0) red[(par * 8 + wave) * PF_RED + 32] = ssw;
                    // The 32 x 1024 GEMV on the matrix cores.  x . g is split into three bf16 terms (truncation: hi + mid + lo == the f32 value
                    // exactly) that become columns 0 / 1 / 2 of the B operand; a wave multiplies its 128-deep K slice of the two 16-row tiles
                    // (8 x v_mfma_f32_16x16x32_bf16, A = the resident weight fragments) and the K reduction happens inside the instruction:
                    // 24 VALU operations per lane instead of 128 unpack / FMA + a 70-instruction halving tree.
                    {
                        uint32_t* xb = reinterpret_cast<uint32_t*>(smem + L_XB);
                        uint32_t pk[3];
                        {
                            const uint32_t h0 = __float_as_uint(xn0) & 0xFFFF0000u, h1 = __float_as_uint(xn1) & 0xFFFF0000u;
                            const float r0 = xn0 - __uint_as_float(h0), r1 = xn1 - __uint_as_float(h1);
                            const uint32_t m0 = __float_as_uint(r0) & 0xFFFF0000u, m1 = __float_as_uint(r1) & 0xFFFF0000u;
                            const uint32_t l0 = __float_as_uint(r0 - __uint_as_float(m0)), l1 = __float_as_uint(r1 - __uint_as_float(m1));
                            pk[0] = (h0 >> 16) | h1; pk[1] = (m0 >> 16) | m1; pk[2] = (l0 >> 16) | (l1 & 0xFFFF0000u);
                        }
                        // a wave's K slice [128 wave, +128) is exactly what its own 64 lanes swept: the LDS round trip is a transpose inside the
                        // wave (its LDS operations execute in order), no workgroup barrier
                        xb[tid] = pk[0]; xb[512 + tid] = pk[1]; xb[1024 + tid] = pk[2];
                        if (lane < 4) xb[1536 + 4 * wave + lane] = 0u;
                        __builtin_amdgcn_wave_barrier();
                        const int n = lane & 15, q4 = lane >> 4;
                        const u32x4* xbv = reinterpret_cast<const u32x4*>(smem + L_XB);
                        const int zslot = 384 + wave;
                        const int src = n < 3 ? n * 128 + 16 * wave + q4 : zslot;  // (+ 4 j per k-step below)
                        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, xbv[n < 3 ? src + 4 * j : zslot]);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w13v[l][j]), bv, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w13v[l][4 + j]), bv, acc1, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {  // columns 0 + 1 + 2 (row_shl: lane i reads lane i + 1 / i + 2 of its row of 16)
                            acc0[r] += pf_dpp<0x101>(acc0[r]) + pf_dpp<0x102>(acc0[r]);
                            acc1[r] += pf_dpp<0x101>(acc1[r]) + pf_dpp<0x102>(acc1[r]);
                        }
                        if (n == 0) {  // D[row 4 q4 + r][column 0]
                            *reinterpret_cast<float4*>(red + (par * 8 + wave) * PF_RED + 4 * q4) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
                            *reinterpret_cast<float4*>(red + (par * 8 + wave) * PF_RED + 16 + 4 * q4) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
                        }
                    }
                    PF_TICK(8);
                    __syncthreads();
                    PF_TICK(15);
                    if (wave == 0) {
                        const int jj = lane & 15, k = lane >> 4;
                        const float* rp = red + (par * 8 + k) * PF_RED;
                        const float2 g2 = *reinterpret_cast<const float2*>(rp + 2 * jj), h2 = *reinterpret_cast<const float2*>(rp + 4 * PF_RED + 2 * jj);
                        float ga = pf_sum_rows(g2.x + h2.x), gb = pf_sum_rows(g2.y + h2.y);
                        const float tot = pf_sum_rows(rp[32] + rp[4 * PF_RED + 32]);
                        const float dni = pf_rms_inv(tot, A.eps);
                        if (fp8) { ga *= s_scl[48 * l + 12 + 2 * jj]; gb *= s_scl[48 * l + 12 + 2 * jj + 1]; }
                        ga *= dni; gb *= dni;
                        const float act = pf_silu(ga) * gb;  // candle silu = x / (1 + exp(-x))
                        pf_publish(edges, e, k, 16 * b + jj, tag0 + e + 1, act);
                        pf_publish(edges, e, k + 4, 16 * b + jj, tag0 + e + 1, act);
                    }
                    par ^= 1;
                    PF_TICK(3);
                }
                // ================= S4: gather the 4096 activations -> W2 rows (LDS-resident) + residual -> publish 4
                {
                    tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                    u32x4 v[4];
                    pf_nap_before_sweep(A.naps[3]);
                    pf_sweep4(my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP, tid, tag0 + e + 1, v, dead, A.ctl);
                    ++e;
                    PF_TICK(12);
                    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 w = w2s[(l * 4 + q) * PF_THREADS + tid];
                        const float c0 = __uint_as_float(v[q].x), c1 = __uint_as_float(v[q].z);
                        a4[0] = fmaf(bf_lo(w.x), c0, a4[0]); a4[0] = fmaf(bf_hi(w.x), c1, a4[0]);
                        a4[1] = fmaf(bf_lo(w.y), c0, a4[1]); a4[1] = fmaf(bf_hi(w.y), c1, a4[1]);
                        a4[2] = fmaf(bf_lo(w.z), c0, a4[2]); a4[2] = fmaf(bf_hi(w.z), c1, a4[2]);
                        a4[3] = fmaf(bf_lo(w.w), c0, a4[3]); a4[3] = fmaf(bf_hi(w.w), c1, a4[3]);
                    }
                    const float r4 = pf_reduce<4>(a4, lane);
                    if ((lane & 15) == 0) red[(par * 8 + wave) * PF_RED + (lane >> 4)] = r4;
                    float xres = 0.f;
                    if (tid < 64) xres = xr[min(tid & 15, 3)];
                    __syncthreads();
                    if (wave == 0) {
                        const int r = min(lane & 15, 3), k = lane >> 4;
                        const float* rp = red + (par * 8 + k) * PF_RED;
                        float t = pf_sum_rows(rp[r] + rp[4 * PF_RED + r]);
                        if (fp8) t *= s_scl[48 * l + 44 + r];
                        if ((lane & 15) < 4) {
                            pf_publish(edges, e, k, 4 * b + r, tag0 + e + 1, xres + t);
                            pf_publish(edges, e, k + 4, 4 * b + r, tag0 + e + 1, xres + t);
                        }
                    }
                    par ^= 1;
                    PF_TICK(4);
                }
            }
            // ================= head: gather x -> fast_norm -> 4 rows of fast_output -> publish 4 logits
            {
                tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                const float2 nw = norm_w(2 * PF_LAYERS, tid);
                u32x4 v;
                pf_nap_before_sweep(A.naps[4]);
                pf_sweep1(my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP, tid, tag0 + e + 1, v, dead, A.ctl);
                ++e;
                PF_TICK(13);
                x0 = __uint_as_float(v.x); x1 = __uint_as_float(v.z);
                const float xn0 = x0 * nw.x, xn1 = x1 * nw.y;  // RMSNorm folded through the GEMV (see S1)
                float a8[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) a8[r] = pf_dot2(wr[PF_LAYERS * 9 + r], xn0, xn1, 0.f);
                a8[4] = fmaf(x1, x1, fmaf(x0, x0, 0.f));
                a8[5] = 0.f; a8[6] = 0.f; a8[7] = 0.f;
                const float r8 = pf_reduce<8>(a8, lane);
                if ((lane & 7) == 0) red[(par * 8 + wave) * PF_RED + (lane >> 3)] = r8;
                __syncthreads();
                if (wave == 0) {
                    const int r = min(lane & 15, 3), k = lane >> 4;
                    const float* rp = red + (par * 8 + k) * PF_RED;
                    float t = pf_sum_rows(rp[r] + rp[4 * PF_RED + r]);
                    const float tot = pf_sum_rows(rp[4] + rp[4 * PF_RED + 4]);
                    if (fp8) t *= s_scl[192 + r];
                    if ((lane & 15) < 4) {
                        const float val = t * pf_rms_inv(tot, A.eps);
                        pf_publish(edges, e, k, 4 * b + r, tag0 + e + 1, val);
                        pf_publish(edges, e, k + 4, 4 * b + r, tag0 + e + 1, val);
                    }
                }
                par ^= 1;
                    PF_TICK(5);
            }
            // ================= decision: gather the 1024 logits -> rep-pen -> argmax (host ArgMax rule: LAST maximal index) -> next input
            {
                tid = pf_opaque(tid_k); lane = tid & 63; wave = tid >> 6;
                u32x4 v;
                pf_nap_before_sweep(A.naps[5]);
                pf_sweep1(my_edges + (size_t)(e & 3) * PF_REPL * PF_EDGE_CAP, tid, tag0 + e + 1, v, dead, A.ctl);
                ++e;
                PF_TICK(14);
                float lv0 = __uint_as_float(v.x), lv1 = __uint_as_float(v.z);
                // SingleBatchedRepPenProcessor::apply (rep_pen.rs:37-65); "token in tokens_seen" == "mask[token] == penalty"
                int last = -1, dropped = -1, head = 0, len = 0;
                bool drop = false;
                if (have_prev) {
                    last = (int)s_prev[cb + 1];
                    head = (s_meta[cb * 2] + 16) % 17;  // push_front
                    len = s_meta[cb * 2 + 1] + 1;
                    drop = len > 16;
                    if (drop) dropped = s_ring[cb * 17 + (head + len - 1) % 17];  // pop_back
                }
                float m0 = ((mbits >> (2 * cb)) & 1u) ? cfg.rep_pen : 1.0f, m1 = ((mbits >> (2 * cb + 1)) & 1u) ? cfg.rep_pen : 1.0f;
                if (have_prev) {
                    const float o0 = m0, o1 = m1;
                    const int i0 = 2 * tid, i1 = 2 * tid + 1;
                    if (i0 == last) m0 = cfg.rep_pen;
                    if (i0 == dropped && m0 == cfg.rep_pen) m0 = 1.0f;
                    if (i1 == last) m1 = cfg.rep_pen;
                    if (i1 == dropped && m1 == cfg.rep_pen) m1 = 1.0f;
                    if (b == 0) {
                        if (m0 != o0) A.rp.mask[(size_t)cb * 1024 + i0] = m0;
                        if (m1 != o1) A.rp.mask[(size_t)cb * 1024 + i1] = m1;
                        if (tid == 0) { A.rp.ring[cb * 17 + head] = last; A.rp.ring_meta[cb * 2] = head; A.rp.ring_meta[cb * 2 + 1] = drop ? 16 : len; }
                    }
                    lv0 = lv0 / m0; lv1 = lv1 / m1;
                }
                float* cap = (A.cap && b == 0 && A.slow_logits && A.state->frame < A.cap_frames) ? A.cap + ((size_t)A.state->frame * 9 + 1 + cb) * 2048 : nullptr;
                if (cap) *reinterpret_cast<float2*>(cap + 2 * tid) = make_float2(lv0, lv1);  // the penalised logits this decision sees
                int gi;
                if (SAMPLED) {
                    int used = 0;
                    __syncthreads();  // (the scratch aliases this stage's own LDS: nobody is still reading the previous decision's)
                    const float lvv[2] = {lv0, lv1};
                    gi = bsample<PF_THREADS, 2, 8>(lvv, 1024, cfg.top_k, (float)(1.0 / (double)cfg.temp), cfg.top_p, s_words[n_draws], &used, samp);
                    n_draws += used;
                    PF_TICK(6);
                } else {
                // per lane: ascending index, the later equal value wins
                float bv = lv0;
                int bi = 2 * tid;
                if (!(lv1 < bv)) { bv = lv1; bi = 2 * tid + 1; }
                float wm = bv;
                wm = fmaxf(wm, pf_dpp<PF_XOR1>(wm)); wm = fmaxf(wm, pf_dpp<PF_XOR2>(wm));
                wm = fmaxf(wm, pf_dpp<PF_HALF_MIRROR>(wm)); wm = fmaxf(wm, pf_dpp<PF_MIRROR>(wm));
                wm = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 31))),
                           fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 47)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 63))));
                int ci = (bv == wm) ? bi : -1;
                ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_XOR1, 0xF, 0xF, false)); ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_XOR2, 0xF, 0xF, false));
                ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_HALF_MIRROR, 0xF, 0xF, false)); ci = max(ci, __builtin_amdgcn_mov_dpp(ci, PF_MIRROR, 0xF, 0xF, false));
                ci = max(max(__builtin_amdgcn_readlane(ci, 15), __builtin_amdgcn_readlane(ci, 31)), max(__builtin_amdgcn_readlane(ci, 47), __builtin_amdgcn_readlane(ci, 63)));
                if (lane == 0) { amax[(par * 8 + wave) * 2] = wm; amax[(par * 8 + wave) * 2 + 1] = __int_as_float(ci); }
                __syncthreads();
                float gv = amax[(par * 8) * 2];
                gi = __float_as_int(amax[(par * 8) * 2 + 1]);
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const float v2 = amax[(par * 8 + w) * 2];
                    const int i2 = __float_as_int(amax[(par * 8 + w) * 2 + 1]);
                    if (v2 > gv || (v2 == gv && i2 > gi)) { gv = v2; gi = i2; }
                }
                par ^= 1;
                    PF_TICK(6);
                }
                const uint32_t code = (uint32_t)max(gi, 0);
                if (cap && tid == 0) cap[1024] = (float)code;
                if (tid == 0) {
                    s_misc[4 + cb] = code;
                    if (b == 0) A.state->cur[cb + 1] = code;
                }
                if (cb != 7) {  // hidden_states = fast_embeddings(code) (single_batch.rs:181-183)
                    if (A.qkv0_tbl) {
                        const int h2 = tid >> 5, g2 = h2 >> 3, j2 = tid & 31;  // (S2's lane geometry: head, kv group, dim pair)
                        const float* row = A.qkv0_tbl + (size_t)code * 1280;
                        tq = *reinterpret_cast<const float2*>(row + 2 * tid);
                        tk2 = *reinterpret_cast<const float2*>(row + 1024 + g2 * 64 + 2 * j2);
                        tv2 = *reinterpret_cast<const float2*>(row + 1152 + g2 * 64 + 2 * j2);
                        const uint32_t hw = reinterpret_cast<const uint16_t*>(A.fast_emb)[(size_t)code * 1024 + 4 * b + min(tid & 15, 3)];
                        xres_tbl = __uint_as_float(hw << 16);
                    } else {
                        const uint32_t ew = reinterpret_cast<const uint32_t*>(A.fast_emb)[(size_t)code * 512 + tid];
                        x0 = bf_lo(ew); x1 = bf_hi(ew);
                    }
                }
            }
        }
    } else if (tid < 8) {
        s_misc[4 + tid] = 0;
        A.state->cur[tid + 1] = 0;
    }
    if (b != 0) return;
    // ---- end of frame, workgroup 0 only (single_batch.rs:185-210 + generate_blocking :250,264-266)
    __syncthreads();
    uint32_t* cur = s_prev;  // reuse: [slow, c0..c7]
    if (tid == 0) cur[0] = cur0;
    else if (tid <= 8) cur[tid] = s_misc[3 + tid];
    __syncthreads();
    if (tid == 0) {
        SeqState* st = A.state;
        const int frame = st->frame;
        if (done_in == 1) st->done = 2;
        if (A.slow_logits) st->cur[0] = cur0;
        if (frame == 0 || cur0 != cfg.im_end_id) {
            const int o = st->n_out;
            if (o < A.out_cap)
                for (int cc = 0; cc < 8; ++cc) A.out_codes[(size_t)cc * A.out_cap + o] = cur[cc + 1];
            st->n_out = o + 1;
        }
        for (int i = 0; i <= 8; ++i) st->prev[i] = cur[i];
        st->have_prev = 1;
        st->pos += 1;
        st->frame = frame + 1;
        if (SAMPLED || (cfg.legacy && A.slow_logits)) A.rng->consumed = consumed0 + (unsigned long long)n_draws;
        A.ctl[0] = epoch + 1;
    }
    // next slow input: embed([slow, c0..c7]) (dual_ar.rs:532-567): token row first, then the codebook rows in order
    {
        const uint32_t sem = cur[0];
        const float m = (sem >= cfg.sem_lo && sem <= cfg.sem_hi) ? 1.f : 0.f;
        const uint32_t* te = reinterpret_cast<const uint32_t*>(A.tok_emb);
        const uint32_t* ce = reinterpret_cast<const uint32_t*>(A.cb_emb);
        const uint32_t w0 = te[(size_t)sem * 512 + tid];
        float e0 = 0.f + bf_lo(w0), e1 = 0.f + bf_hi(w0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t wv = ce[((size_t)c * 1024 + cur[c + 1]) * 512 + tid];
            e0 += bf_lo(wv) * m;
            e1 += bf_hi(wv) * m;
        }
        *reinterpret_cast<float2*>(A.x + 2 * tid) = make_float2(e0, e1);
    }
    PF_TICK(7);
    if (A.prof && tid == 0) {
        for (int k = 0; k < 16; ++k) A.prof[k] += tk[k];
        const unsigned long long t_end = wall_clock64(), peer_end = A.peer_stamps ? A.peer_stamps[0] : 0;
        if (peer_end && t_entry > peer_end && t_entry - peer_end < 20000) { A.prof[17] += t_entry - peer_end; A.prof[20] += 1; }
        A.prof[18] += t_timers - t_entry;
        A.prof[19] += t_end - t_last;
        A.prof[16] = t_end;
    }
#undef PF_TICK
}

// ------------------------------------------------------------------------------------------------ reduction self-test
__global__ __launch_bounds__(64) void k_pf_reduce_selftest(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float v32[32], v16[16], v8[8], v4[4];
#pragma unroll
    for (int i = 0; i < 32; ++i) v32[i] = in[lane * 32 + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = in[lane * 32 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = in[lane * 32 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = in[lane * 32 + i];
    const float r32 = pf_reduce<32>(v32, lane), r16 = pf_reduce<16>(v16, lane), r8 = pf_reduce<8>(v8, lane), r4 = pf_reduce<4>(v4, lane);
    // out: [0, 32) N = 32; [32, 48) N = 16; [48, 56) N = 8; [64, 68) N = 4; [68] wave sum.  The LAST lane of every group writes, so a
    // result that is wrong in part of the group is seen
    if ((lane & 1) == 1) out[lane >> 1] = r32;
    if ((lane & 3) == 3) out[32 + (lane >> 2)] = r16;
    if ((lane & 7) == 7) out[48 + (lane >> 3)] = r8;
    if ((lane & 15) == 15) out[64 + (lane >> 4)] = r4;
    const float ws = pf_wave_sum(in[lane * 32]);
    if (lane == 37) out[68] = ws;
}

// ------------------------------------------------------------------------------------------------ host side
bool fast_persist_supported(const ModelDims& d, int n_fast_layer, int n_cb, int cb_size) {
    return d.dim == 1024 && d.inter == 4096 && d.H == 16 && d.Hk == 2 && d.Dh == 64 && n_fast_layer == PF_LAYERS && n_cb == 8 &&
           cb_size == 1024;
}
size_t fast_persist_pack_bytes() { return (size_t)PF_BLOCKS * PF_CHUNKS * PF_THREADS * 16; }
size_t fast_persist_edge_bytes() { return (size_t)PF_RING * PF_REPL * PF_EDGE_CAP * 8; }

void launch_fast_persist_pack(const LayerW* fast, const void* head_w, void* pack, hipStream_t st, bool fp8, const float* head_s, float* scales) {
    if (fp8) {
        hipLaunchKernelGGL(k_pf_pack<true>, dim3(PF_BLOCKS, PF_CHUNKS), dim3(PF_THREADS), 0, st, fast[0], fast[1], fast[2], fast[3], head_w,
                           reinterpret_cast<u32x4*>(pack));
        hipLaunchKernelGGL(k_pf_pack_scales, dim3(PF_BLOCKS), dim3(256), 0, st, fast[0], fast[1], fast[2], fast[3], head_s, scales);
    } else {
        hipLaunchKernelGGL(k_pf_pack<false>, dim3(PF_BLOCKS, PF_CHUNKS), dim3(PF_THREADS), 0, st, fast[0], fast[1], fast[2], fast[3], head_w,
                           reinterpret_cast<u32x4*>(pack));
    }
    FS_HIP(hipGetLastError());
}

size_t fast_persist_qkv0_bytes() { return (size_t)1024 * 1280 * sizeof(float); }
void launch_fast_persist_qkv0_table(const void* pack, const float* scales, const float* norm0, const void* fast_emb, float eps, float* tbl,
                                    hipStream_t st) {
    hipLaunchKernelGGL(k_pf_qkv0_table, dim3(PF_BLOCKS, 32), dim3(PF_THREADS), 0, st, pack, scales, norm0, fast_emb, eps, tbl);
    FS_HIP(hipGetLastError());
}

void launch_fast_persist(const FastPersistArgs& a, bool sampled, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_persist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END));
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_persist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END));
        attr_set = true;
    }
    if (sampled) hipLaunchKernelGGL(k_fast_persist<true>, dim3(PF_BLOCKS), dim3(PF_THREADS), L_END, st, a);
    else hipLaunchKernelGGL(k_fast_persist<false>, dim3(PF_BLOCKS), dim3(PF_THREADS), L_END, st, a);
    FS_HIP(hipGetLastError());
}
bool fast_persist_samples(float temp, int top_k, int cb_size) { return temp > 0.f && top_k > 0 && top_k <= BS_MAXK && top_k < cb_size; }

void launch_pf_reduce_selftest(const float* in, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_pf_reduce_selftest, dim3(1), dim3(64), 0, st, in, out);
    FS_HIP(hipGetLastError());
}

}  // namespace fs
