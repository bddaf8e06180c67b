// Persistent fast-decoder kernel of the batch-1 decode frame (lm_persist.hip).  gfx950 only, Fish geometry only
// (dim 1024, 16 x 64 heads over 2 kv heads, intermediate 4096, 4 fast layers, 8 codebooks of 1024 codes, bf16 weights).
//
// One launch = forward_generate_fast x 8 + the 8 codebook decisions of ONE audio frame (dual_ar.rs:638-673,
// single_batch.rs:146-210): 256 workgroups (one per CU) x 512 threads keep their slice of ALL fast-decoder weights
// on chip for the whole launch (168 VGPRs per lane + 128 KB of LDS per CU = 475 KB per CU = the 121.7 MB of the four fast
// blocks + fast_output), and the 17 dependent GEMV stages of a codebook pass hand their output vectors to every
// workgroup as 8-byte {value, tag} granules through HBM-side memory (relaxed agent-scope stores / loads, MI355X guide
// Guideline 16 R2) instead of through 18 kernel boundaries.  Measured (tools/ubench_engine.hip): an in-launch
// all-to-all edge + a resident-weight GEMV stage costs 1.6-1.8 us against 3.7 us per graph node.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lm_kernels.h"

namespace fs {

constexpr int PF_BLOCKS = 256;     // workgroups == CUs (all must be co-resident: grid-wide data dependencies)
constexpr int PF_THREADS = 512;    // 8 waves, 2 per SIMD -> 256 VGPRs per lane
constexpr int PF_REPL = 8;         // replicas of every edge buffer (workgroup b sweeps replica b % 8)
constexpr int PF_EDGE_CAP = 4096;  // granules per edge buffer (largest edge: the SwiGLU activations)
constexpr int PF_RING = 4;         // edge buffers in rotation
constexpr int PF_LAYERS = 4;
constexpr int PF_REG_DW = PF_LAYERS * (5 + 4 + 32) + 4;  // weight dwords (bf16 pairs) per lane held in VGPRs: wqkv, wo, w13 per layer + head
constexpr int PF_REG_CHUNKS = PF_REG_DW / 4;              // 42 x 16 B
constexpr int PF_ROW_CHUNKS = (PF_LAYERS * 9 + 4) / 4;    // of which 10 hold row pairs (Wqkv 5 + Wo 4 per layer, 4 head rows); 32 hold W13 MFMA fragments
constexpr int PF_LDS_CHUNKS = PF_LAYERS * 4;              // w2 of every layer lives in LDS: 16 x 16 B per lane
constexpr int PF_CHUNKS = PF_REG_CHUNKS + PF_LDS_CHUNKS;  // 58 x 16 B x 512 lanes = 475 KB per workgroup
constexpr int PF_SCL = 200;                               // FS_FP8 handles: row scales per workgroup (4 layers x 48 + 4 head rows, padded)

struct FastPersistArgs {
    const void* wpack;          // [PF_BLOCKS][PF_CHUNKS][PF_THREADS] x 16 B: per-lane weight image (launch_fast_persist_pack)
    const float* scales;        // null, or [PF_BLOCKS][PF_SCL]: FS_FP8 handle -- the image holds the e4m3 weights widened to bf16, these are their row scales
    const float* norms[2 * PF_LAYERS + 1];  // attention_norm l, ffn_norm l (l = 0..3), fast_norm: f32 [1024]
    const void* fast_emb;       // bf16 [1024][1024]
    const float* qkv0_tbl;      // null, or f32 [1024][1280]: what S1 of fast layer 0 publishes for input fast_embeddings[code] (launch_fast_persist_qkv0_table)
    const void* tok_emb;        // bf16 [V][1024]
    const void* cb_emb;         // bf16 [8 * 1024][1024]
    const float* cos_t;         // [max_seq_len][32] (rows 0..7 used: RoPE position = codebook index, dual_ar.rs:651-655)
    const float* sin_t;
    float eps;
    const float* xf;            // [1024] hidden state of the slow transformer (input of codebook pass 0)
    const float* slow_logits;   // null, or [n_slow] audio-range logits of this step: the slow-token decision runs inside this launch
    int n_slow;                 // <= 2048
    float* cap;                 // null, or [cap_frames][9][2048] (fs_lm_debug_capture): the logits every decision of a frame saw + its pick
    int cap_frames;
    float* const* hid_slot;     // with slow_logits: device pointer cell (may hold null) of the hidden-state rows [iteration][1024]
    float* x;                   // [1024] out: embedded input of the next slow step
    SeqState* state;
    const SampleCfg* cfg;
    RepPenState rp;
    RngState* rng;              // the request's StdRng stream position (sampled requests: 8 words per frame, written back by workgroup 0)
    uint32_t* out_codes;
    int out_cap;
    unsigned long long* edges;  // [PF_RING][PF_REPL][PF_EDGE_CAP] granules (zeroed once at allocation)
    unsigned long long* prof;   // null, or [16 + 8]: workgroup 0 accumulates 10 ns ticks per stage kind (FISHRT_PERSIST_PROF=1); [16..]: launch-boundary stamps
    const unsigned long long* peer_stamps;  // prof only: the slow kernel's stamp area (its [0] = the absolute time its timed workgroup finished)
    uint32_t* ctl;              // [0] launch counter (tag epoch), [1] spin-timeout count (host checks it), [2] launches whose sampling configuration does not match the instantiation
    int naps[6];                // 64-clock naps before the first sweep of S1 / S2 / S3 / S4 / head / decision (lm_persist_dev.h pf_nap_before_sweep)
};

// ---- persistent slow-transformer kernel (lm_persist_slow.hip): one launch = the 24 blocks + the audio-range head of one decode step
constexpr int PS_EDGE_CAP = 16 * 16 * 66 + 64;  // largest edge: attention partials of 16 heads x 16 token slices x {o[64], m, l}
constexpr size_t PS_LAYER_IMAGE = 116736;        // bytes per (layer, workgroup): Wqkv 5 rows, Wo 4, W13 32, W2 4 rows x 4096
constexpr size_t PS_HEAD_IMAGE = 16384;          // bytes per workgroup: 8 head rows
constexpr size_t PS_LAYER_IMAGE_FP8 = 59392;     // FS_FP8 handles: one e4m3 byte per weight (layout: lm_persist_slow.hip)
constexpr size_t PS_HEAD_IMAGE_FP8 = 8192;

struct SlowPersistArgs {
    const void* wpack;      // [n_layer][PF_BLOCKS][PS_LAYER_IMAGE] per-lane weight images (launch_slow_persist_pack)
    const void* hpack;      // [PF_BLOCKS][PS_HEAD_IMAGE] head rows [8b, 8b+8) (zero beyond n_head_rows)
    const float* scales;    // FS_FP8 images: per-row f32 scales [n_layer][PF_BLOCKS][48] (null: bf16 images)
    const float* hscales;   // FS_FP8: head row scales [PF_BLOCKS][8]
    const float* norms;     // [2 * n_layer + 1][1024] f32: attention_norm l, ffn_norm l, ..., norm
    int n_layer, n_head_rows;
    const float* cos_t;     // [max_seq_len][32]
    const float* sin_t;
    float eps;
    float* x;               // [1024] in: embedded input of this position; out: pre-norm hidden state (forward_generate's `hidden`)
    float* logits;          // [n_head_rows] out
    const SeqState* state;  // pos (KV length), rope_off, done
    void* kv_pool;          // bf16; layer l: K pool at l * 2 * layer_half, V pool at + layer_half (elements)
    size_t layer_half;      // n_pages * page_elems
    const int* page_table;
    int n_sl;               // token slices per head of the attention stage (1, 2, 4, 8 or 16)
    unsigned long long* edges;  // [PF_RING][PF_REPL][PS_EDGE_CAP]
    unsigned long long* prof;   // null, or [16 + 8] (see FastPersistArgs)
    const unsigned long long* peer_stamps;  // prof only: the fast kernel's stamp area
    uint32_t* ctl;          // [0] epoch, [1] timeouts
    int naps[6];            // 64-clock naps before the first sweep of S1 / S2 / S3 / S4 / S5 / head
    int prof_wg;            // the workgroup whose stage timers go to `prof` (FISHRT_PERSIST_PROF_WG; default 0 = an attention workgroup)
    int l2_touch;           // 1: workgroups without an attention item request a layer's W13 slice behind S1's publish instead of in S3 (FISHRT_SLOW_NO_EARLY13=1 switches it off)
};
size_t slow_persist_pack_bytes(int n_layer, bool fp8 = false);
size_t slow_persist_scale_floats(int n_layer);
void launch_slow_persist_pack_fp8(const LayerW* layers, int n_layer, const void* head_w, const float* head_s, int n_head_rows,
                                  const float* const* norm_ptrs, void* wpack, void* hpack, float* scales, float* norms_flat, hipStream_t st);
size_t slow_persist_edge_bytes();
void launch_slow_persist_pack(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, const float* const* norm_ptrs,
                              void* wpack, void* hpack, float* norms_flat, hipStream_t st);
void launch_slow_persist(const SlowPersistArgs& a, hipStream_t st);

// true when the model has the geometry the kernel is written for
bool fast_persist_supported(const ModelDims& d, int n_fast_layer, int n_cb, int cb_size);
size_t fast_persist_pack_bytes();
size_t fast_persist_edge_bytes();
// re-lays the four fast blocks' matrices + fast_output into the per-lane image (device to device, once per weight load)
void launch_fast_persist_pack(const LayerW* fast, const void* head_w, void* pack, hipStream_t st, bool fp8 = false, const float* head_s = nullptr,
                              float* scales = nullptr);
// layer-0 qkv table of the codebook passes 1..7 (built from the packed image with stage S1's own arithmetic; scales: FS_FP8 row scales or null)
size_t fast_persist_qkv0_bytes();
void launch_fast_persist_qkv0_table(const void* pack, const float* scales, const float* norm0, const void* fast_emb, float eps, float* tbl,
                                    hipStream_t st);
void launch_fast_persist(const FastPersistArgs& a, bool sampled, hipStream_t st);
// true when the in-launch sampler covers this configuration (0 < top_k <= 256 candidates kept, sampling/mod.rs:51-132); temp == 0 is the greedy kernel
bool fast_persist_samples(float temp, int top_k, int cb_size);
// self-test hook of the multi-value wave reductions: out[w][i] = sum over the 64 lanes of in[lane][i] for N in {4, 16, 32}
void launch_pf_reduce_selftest(const float* in /*[64][32]*/, float* out /*[3][32]*/, hipStream_t st);

}  // namespace fs
