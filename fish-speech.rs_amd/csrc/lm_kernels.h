// Launch interface of the dual-AR decode kernels (lm_kernels.hip).  gfx950 only.
//
// Data layout in HBM:
//  * weights: row-major [out, in] in WT (bf16, f32, or OCP e4m3fn bytes + one f32 scale per row), one 256-B aligned slab
//    per tensor; with fp8 weights the embedding tables and the KV cache stay bf16 (KVT<WT>); W1/W3 are stored
//    row-interleaved (row 2r = w1[r], row 2r+1 = w3[r]) so one wave streams a contiguous 2-row block and applies
//    SwiGLU in registers;
//  * KV cache: paged.  One pool per layer; page = 64 tokens; K element (t, g, d) lives at
//        pool_k + ((page_table[t >> 6] * Hk + g) * 64 + (t & 63)) * Dh + d
//    so that one wave-wide 16-B/lane load covers 8 (bf16) consecutive tokens of ONE kv head (1 KiB contiguous);
//  * activations: f32 vectors (residual stream x[dim], q[H*Dh], act[inter], logits) -- a few KB, L2-resident.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fs_common.h"

namespace fs {

// storage type of the KV cache and the embedding tables for a given weight type (fp8 weights keep bf16 KV / embeddings)
template <typename WT> struct KVOf { using type = WT; };
template <> struct KVOf<fp8_t> { using type = bf16_t; };
template <typename WT> using KVT = typename KVOf<WT>::type;

constexpr int KV_PAGE = 64;  // tokens per KV page

// Per-sequence device-resident generation state (read by the captured graph; never passed by value).
struct SeqState {
    int pos;        // current KV length T == index where the next token's K/V is written
    int rope_off;   // RoPE row = pos + rope_off (forward_generate's input_pos may differ from the KV length)
    int frame;      // decode iterations executed in this generate call
    int done;       // <|im_end|> sampled (audio_only termination, single_batch.rs:199-204)
    int n_out;      // frames recorded in out_codes
    int have_prev;  // previous_codes.is_some() (single_batch.rs:162-168)
    int step;       // prompt column consumed by the next prefill step
    int prompt_L;   // row stride of the staged prompt (tokens per row)
    uint32_t cur[16];   // [slow, c0..c7] of the frame being generated
    uint32_t prev[16];  // previous frame (rep-pen key)
};

struct KVView {
    void* k;                // pool base for this layer
    void* v;
    const int* page_table;  // [max_pages] for this sequence
};

struct LayerW {
    const void* wqkv;  // [(H+2Hk)*Dh, dim]
    const void* wo;    // [dim, dim]
    const void* w13;   // [2*inter, dim] row-interleaved
    const void* w2;    // [dim, inter]
    const float* attn_norm;
    const float* ffn_norm;
    // fp8 (e4m3fn) weights only: one f32 dequantisation scale per output row of each matrix (null otherwise)
    const float* s_qkv = nullptr;
    const float* s_o = nullptr;
    const float* s_13 = nullptr;  // interleaved like w13
    const float* s_2 = nullptr;
    // false: weights are streamed once per frame -> non-temporal loads (slow transformer, 717 MB / frame);
    // true: re-read 8x per frame and small enough for the 256 MB Infinity Cache (fast decoder, 122 MB) -> default policy
    bool cache_resident = false;
};

struct ModelDims {
    int dim, inter, H, Hk, Dh, n_rep;
    float eps;
};

struct SampleCfg;
size_t rows_xchg_bytes(int dim);

// Buffers + row mapping of the MFMA row path (prefill / batched decode).  All activation buffers hold Mcap rows (a multiple
// of 32: the GEMMs read whole 32-row panels, so rows >= M must exist and hold finite values).
struct RowsCtx {
    int Mcap;            // row capacity of every buffer below (except `part`)
    int part_rows;       // row capacity of `part` (chunked attention: static-batch steps, head_dim != 64 prefill passes)
    int down_split;      // K ranges (== slabs) of the down projection: inter / down_split must be 1024, 256 or 128
    float* X;            // [Mcap][dim]   residual stream (f32)
    float* Q;            // [Mcap][dim]   rope'd queries (f32)
    float* part;         // [part_rows][H][n_chunks_max][Dh + 2] attention partials
    float* P;            // [down_split][Mcap][dim] down-projection split-K slabs
    uint16_t* A;         // [Mcap][dim]   bf16 hi+lo GEMM input (normed x / attention output), fragment-major (lm_kernels.hip frag_off)
    uint16_t* C;         // [Mcap][inter] bf16 hi+lo SwiGLU activations, fragment-major
    uint16_t* A2 = nullptr;  // [Mcap][dim] second GEMM-input buffer: with `ss` set, the ffn RMSNorm is folded into the Wo / W13 GEMMs
    float* ss = nullptr;     // [Mcap][dim / 16] sum-of-squares partials of the Wo GEMM's blocks
    const float *cos_t, *sin_t;
    const SeqState* state;  // position source (row m sits at state->pos + m * pos_step)
    int n_chunks_max;    // stride of `part`
    int nc_launch;       // attention chunks launched (covers the longest row of this pass)
    int pos_step;        // 1: rows = consecutive tokens of one sequence (prefill); 0: rows = sequences (batched decode)
    int pt_stride;       // page-table stride between rows (0 for prefill) / between sequences (group prefill)
    int seq_rows = 0;    // > 0: group prefill -- rows = seq_rows consecutive tokens of each of M / seq_rows sequences, all starting at state->pos
    bool no_flash = false;       // micro-benchmark / test hook: keep the chunked row attention for prefill passes
    bool chunked_attn = false;   // micro-benchmark / test hook: keep k_attn_decode + k_attn_combine for rows-are-sequences passes
    bool small_attn = false;     // every row attends over <= 8 tokens of ONE page (fast decoder): fused attention node
    void* xchg = nullptr;        // fold: exchange units of the layer-closing down projection (rows_xchg_bytes(dim), zero-initialised, used by nothing else)
    uint32_t* epoch = nullptr;   // fold: [0] = step epoch (>= 1; bumped once per step by the slow-token sampler node), [1] = exchange timeouts
    uint32_t node_id = 0;        // fold: unique per rows_layer call of a step (< 4096)
    bool attn_t1 = false;        // fold + small_attn: every row sits at position 0 of an empty cache (first codebook pass): no attention node
    bool identity_pages = false; // small_attn: page_table[m * pt_stride] == m for every row (the batched fast decoder's one-page-per-row table)
    bool first_prepped = false;  // fold: the first layer's normalised input fragments are already in A (written by the sampler that produced the row)
    bool fold = false;           // decode step with the un-split down projection (see rows_layer's next_norm)
    const float* qkv0_tbl = nullptr;         // fold + small_attn, first layer of a codebook pass >= 1: [codebook_size][(H + 2 Hk) Dh] f32 qkv table (lm_persist.hip) -- no Wqkv node
    const SeqState* row_states = nullptr;    // ... the rows' generator states: row m's previous code = row_states[m].cur[code_slot]
    int code_slot = 0;
    unsigned stage_mask = 0xFFu;  // micro-benchmark hook: bit i enables stage i of rows_layer (prep, qkv, attn, combine, wo, prep, w13, w2)
};

// ---- launchers (all asynchronous on `st`) -------------------------------------------------------------------
template <typename WT>
struct LmKernels {
    // x -> rmsnorm -> Wqkv GEMV -> interleaved RoPE(q,k) -> q_out (f32), K/V appended at state->pos (or pos_static)
    // (state != null: KV index = state->pos, RoPE row = state->pos + state->rope_off; else the two static values)
    static void qkv(const ModelDims& d, const float* x, const LayerW& w, const float* cos_t, const float* sin_t,
                    const SeqState* state, int pos_static, int rope_static, float* q_out, KVView kv, hipStream_t st);
    // decode attention over the paged cache: un-normalised partial {o[Dh], m, l} per (q head, token chunk);
    // part: [H][n_chunks_max][Dh + 2]; chunk c covers tokens [c * attn_chunk(), (c + 1) * attn_chunk())
    static int attn_chunk();
    // nc_launch: chunks actually launched (>= ceil(T / attn_chunk()) for every T the launch will see; the host knows
    // the position of every frame, so graphs are captured per power-of-two bucket of nc_launch)
    static void attn_decode(const ModelDims& d, const float* q, KVView kv, const SeqState* state, float* part,
                            int n_chunks_max, int nc_launch, hipStream_t st);
    // combine the chunks of state->pos + 1 tokens (or, fused_T > 0: attend over fused_T <= 8 cached tokens in the
    // prologue) -> Wo GEMV -> x += .
    // nc_launch = the chunk count attn_decode was launched with (ignored when fused_T > 0)
    static void wo(const ModelDims& d, const float* part, int n_chunks_max, int nc_launch, const SeqState* state, const float* q,
                   KVView kv, int fused_T, const LayerW& w, float* x, hipStream_t st);
    static void ffn_up(const ModelDims& d, const float* x, const LayerW& w, float* act, hipStream_t st);
    static void ffn_down(const ModelDims& d, const float* act, const LayerW& w, float* x, hipStream_t st);
    // x -> rmsnorm(norm_w) -> rows [0, n_rows) of W (x wscale[row] for fp8 weights) -> logits f32
    static void head(const ModelDims& d, const float* x, const float* norm_w, const void* W, const float* wscale, int n_rows,
                     float* logits, hipStream_t st);
    // x = tok_emb[t0] + sum_c mask * cb_emb[c*cbsize + t_{c+1}]   (dual_ar.rs:532-567); tokens from prompt column
    // state->step (prompt != null) or from state->cur
    static void embed(const ModelDims& d, const void* tok_emb, const void* cb_emb, int n_cb, int cb_size,
                      const SampleCfg* cfg, const uint32_t* prompt, SeqState* state, float* x, hipStream_t st);
    // ---- MFMA row path (prefill passes, static-batch steps): bf16 weights, or fp8 weights widened to bf16 in registers
    static bool has_mfma_prefill();
    // X[m] = embed(prompt column state->step + m), m < M
    static void prefill_embed(const ModelDims& d, const void* tok_emb, const void* cb_emb, int n_cb, int cb_size,
                              const SampleCfg* cfg, const uint32_t* prompt, const SeqState* state, int M, float* X,
                              hipStream_t st, int seq_rows = 0, size_t prompt_stride = 0);
    // one transformer block over M <= Mcap activation rows (X updated in place up to the down-projection, whose split-K
    // slabs are folded in by the NEXT rows_layer / rows_finish):
    //   x += slabs | rmsnorm+Wqkv+rope+KV append | attention over the paged cache | Wo + residual | rmsnorm+W13+SwiGLU | W2 slabs
    // c.fold (decode steps of <= 32 rows, rows_fold_ok): next_norm = the norm weight of whoever consumes this layer's output (the next
    // layer's attention_norm, or the final norm in front of the head) -- the down projection runs un-split and closes the layer itself
    // (no slabs, no k_prep node; rows_finish is then not called and rows_head takes rms = true)
    static void rows_layer(const ModelDims& d, int M, const RowsCtx& c, const LayerW& w, KVView kv, bool first, hipStream_t st,
                           const float* next_norm = nullptr);
    static bool rows_fold_ok(const ModelDims& d, int M, const RowsCtx& c);
    static void rows_finish(const ModelDims& d, int M, const RowsCtx& c, const float* norm_w, hipStream_t st);
    static void rows_warmup();
    static void rows_head(const ModelDims& d, int M, const RowsCtx& c, const void* W, const float* wscale, int n_rows, float* logits, int ld,
                          hipStream_t st, bool rms = false);
    // fast_embeddings gather: out[i] = fast_emb[ids[i]]
    static void fast_embed(const ModelDims& d, const void* fast_emb, const uint32_t* ids, int n, float* out,
                           hipStream_t st);
};

struct SampleCfg {  // device-resident (the captured graphs read it through a pointer, so one graph serves every config)
    float temp, top_p;
    int top_k;
    float rep_pen;
    int ignore_eos;
    uint32_t im_end_id;
    uint32_t sem_lo, sem_hi;  // embed mask range (inclusive); Fish<=1.4: lo == hi == semantic id
    // slow-token candidates (constrain_probs_to_audio / rescale_semantic_tokens, generate/utils.rs:6-56): candidate 0 = <|im_end|>,
    // candidate i >= 1 = token audio_base + i, audio_base = semantic_start_id - 1.  Fish 1.5: audio_base == im_end_id (the adjacent
    // layout, one contiguous slice of the head); generic DualAR layout: the head image is [W[im_end]; W[semantic_start .. V)].
    uint32_t audio_base = 0;
    // Fish <= 1.4 slow token (single_batch.rs:104-124, sampling/mod.rs:8-26): 2-way softmax over {pad_id, im_end_id},
    // temperature ignored; logits[0] = pad logit, logits[1] = im_end logit
    int legacy;
    uint32_t pad_id;
    // batch_rows > 0: BatchedLogitsProcessor semantics on the single-sequence samplers (row batch_row of a static batch generated on its
    // own, lm_engine.hip generate_batch_sequential): temp <= 1e-7 -> FIRST-max argmax; temp > 0 -> child StdRng of (call, row)
    int batch_rows = 0, batch_row = 0, batch_calls = 0;  // batch_calls = sample() calls per frame (num_codebooks + 1)
    // continuous batching (fs_lm_session_*): the rows of the static-batch step are independent request slots -- a finished or empty slot
    // stops stepping (position, frame counter and outputs frozen) instead of following the live rows in lock-step
    int session = 0;
    // BatchedLogitsProcessor compares top_p (f64) with `sum_p as f64` (sampling/mod.rs:68); the single-sequence LogitsProcessor narrows
    // top_p to f32 first.  The batched samplers (k_sample_*_rows, batch_rows > 0) use this field for that one comparison.
    double top_p64 = 0.0;
};

__host__ __device__ inline uint32_t audio_tok(const SampleCfg& c, int idx) { return idx > 0 ? c.audio_base + (uint32_t)idx : c.im_end_id; }

struct RepPenState {   // rep_pen.rs:4-72, one per codebook
    float* mask;       // [n_cb][cb_size]
    uint8_t* seen;     // [n_cb][cb_size]
    int* ring;         // [n_cb][17]  (push_front / pop_back deque as a ring)
    int* ring_meta;    // [n_cb][2] head, len
};

struct RngState {      // rand 0.8.5 StdRng (ChaCha12) stream position
    uint32_t key[8];
    unsigned long long consumed;  // u32 words consumed so far
};

template <typename WT>
struct SampleKernels {
    // slow token: logits over [im_end, V) (utils.rs:13-16) -> token = idx + im_end; sets done; copies x -> xf; *hid_slot (device
    // pointer cell, may hold null): hidden-state rows [iteration][dim] of generate_blocking_with_hidden
    static void sample_slow(const ModelDims& d, const float* logits, int n, const SampleCfg* c, RngState* rng,
                            SeqState* state, const float* x, float* xf, hipStream_t st, float* const* hid_slot = nullptr);
    // codebook cb: rep-pen (if have_prev), sample, cur[cb+1]; xf = fast_emb[code]; cb == last: finish the frame
    // (record codes, prev = cur, x = embed(cur), pos++, frame++)
    static void sample_fast(const ModelDims& d, const float* logits, int cb, int n_cb, int cb_size, const SampleCfg* c,
                            RngState* rng, RepPenState rp, SeqState* state, const void* fast_emb, float* xf,
                            const void* tok_emb, const void* cb_emb, float* x, uint32_t* out_codes, int out_cap,
                            hipStream_t st);
    // static-batch generator (static_batch.rs): one block per batch row, child StdRng per (call, row)
    // words != null: the block-parallel sampler (512 threads per row) with this step's pre-derived StdRng words (rows_rng_words);
    // requires temp > 1e-7, 0 < top_k <= 256 < candidates (rows_par_sampler_ok)
    static void sample_slow_rows(const ModelDims& d, const float* logits, int ld, int n, const SampleCfg* c, const RngState* master, int B,
                                 int calls_per_frame, SeqState* states, const float* X, float* XF, hipStream_t st, const uint32_t* words = nullptr,
                                 const float* prep_g = nullptr, uint16_t* prep_A = nullptr, uint32_t* epoch = nullptr);
    // prep_g != null (both samplers): the block also leaves RMSNorm(XF row; prep_g) as fragment-major hi/lo GEMM input in prep_A -- what the
    // k_prep node in front of the fast decoder's first layer would produce (RowsCtx::first_prepped)
    static void sample_fast_rows(const ModelDims& d, const float* logits, int cb, int n_cb, int cb_size, const SampleCfg* c,
                                 const RngState* master, int B, SeqState* states, const void* fast_emb, float* XF, const void* tok_emb,
                                 const void* cb_emb, float* X, uint32_t* out_codes, int out_cap, hipStream_t st, const uint32_t* words = nullptr,
                                 const float* prep_g = nullptr, uint16_t* prep_A = nullptr);
    // words[b * 16 + call] = the StdRng word of sample() call `call` of this step for row b (child stream of master u64 number (frame * calls + call) * B + b)
    static void rows_rng_words(const RngState* master, int B, int calls_per_frame, const SeqState* states, uint32_t* words, hipStream_t st);
};
bool rows_par_sampler_ok(double temp, uint64_t top_k, int n_slow, int cb_size);

// dst[0] = src[r0], dst[1] = src[r1] (rows of `dim` elements): the 2-row head of the Fish <= 1.4 slow-token sampler
template <typename WT>
void launch_gather_rows(const void* src, int dim, uint32_t r0, uint32_t r1, void* dst, hipStream_t st);
void launch_advance(SeqState* state, hipStream_t st);  // pos++, step++ (sequential prefill step)
void launch_advance_n(SeqState* state, int n, hipStream_t st);
// decision capture on the row path (static batch / sessions): cap[row][cap_frames][9][2048]
void launch_cap_rows_logits(const float* logits, int ld, int n, const SeqState* states, const SampleCfg* cfg, int B, float* cap, int cap_frames, int decision,
                            hipStream_t st);
void launch_cap_rows_picks(const SeqState* states, const SampleCfg* cfg, int B, float* cap, int cap_frames, int n_cb, hipStream_t st);  // pos += n, step += n (chunked prefill)
void launch_reppen_reset(RepPenState rp, int n_cb, int cb_size, hipStream_t st);

// synthetic tensor fill (fs_synth.h): n_rows x n_cols, destination row = r * row_mul + row_off (W1/W3 interleave)
template <typename WT>
void launch_synth_fill(WT* dst, uint64_t key, int64_t n_rows, int64_t n_cols, int row_mul, int row_off, float mean,
                       float scale, int round_bf16, hipStream_t st);
// f32 -> WT conversion with the same row mapping (safetensors loader)
template <typename WT>
void launch_convert_rows(WT* dst, const float* src, int64_t n_rows, int64_t n_cols, int row_mul, int row_off,
                         hipStream_t st);
// fp8 quantisers: per-row absmax scale (amax / 448, 1 for an all-zero row), e4m3fn RNE bytes; same row mapping for the
// bytes and the scales.  synth: elements come from the generator (fs_synth.h); rows: from a staged f32 matrix.
void launch_synth_quant_fp8(uint8_t* dst, float* scales, uint64_t key, int64_t n_rows, int64_t n_cols, int row_mul, int row_off,
                            float mean, float scale, hipStream_t st);
void launch_quant_rows_fp8(uint8_t* dst, float* scales, const float* src, int64_t n_rows, int64_t n_cols, int row_mul,
                           int row_off, hipStream_t st);
// out[slot * 256 + b] = f32 value the fp8 GEMV unpack path gives byte b at lane-slot `slot` (16 x 256 floats)
void launch_fp8_decode_table(float* out, hipStream_t st);

// test hook behind fs_selftest_sample_rows (include/fishrt.h)
void debug_sample_rows(int device, const float* logits, int B, int n, double temp, double top_p, uint64_t top_k, uint64_t seed,
                       int call_index, uint32_t* out);

}  // namespace fs
