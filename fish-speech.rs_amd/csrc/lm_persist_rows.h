// Persistent decode kernels for R CONCURRENT batch-1 requests ("request rows") on one device (lm_persist_rows.hip, round 4).
//
// The batch-1 persistent kernels (lm_persist.h) are latency-bound: a stage is ~1 us of edge hand-off + ~1 us of arithmetic and one
// generate call owns all 256 CUs.  Here every stage serves R independent requests: the stage's weight slice is streamed ONCE and the
// R activation vectors are the columns of a matrix-core product (v_mfma_f32_16x16x32_bf16; each f32 activation is split into three
// bf16 terms hi + mid + lo == the f32 value exactly, so the products are exact and only the summation order differs from the VALU
// kernels).  Every row keeps its own KV pages, position, sampler / repetition-penalty state and output -- a row computes exactly what
// its own fs_lm_generate call would (generate/single_batch.rs:76-214), the reference's multi-request counterpart being the lock-step
// static batch (generate/static_batch.rs:117-274).
//
// gfx950 only, Fish geometry only (dim 1024, 16 x 64 heads over 2 kv heads, intermediate 4096, 4 fast layers, 8 codebooks x 1024).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lm_kernels.h"
#include "lm_persist.h"

namespace fs {

constexpr int PR_MAX_ROWS = 8;   // request rows of one slow launch (template instantiations: 2, 4, 8)
constexpr int PR_FAST_ROWS = 4;  // request rows of one fast launch (LDS: 16 KB of fast-decoder K/V per row)
constexpr int PR_LD = 2048;      // row stride of the slow logits [R][PR_LD]

struct RowsSlowArgs {
    const void* wimg;       // [n_layer][PF_BLOCKS][PS_LAYER_IMAGE]: MFMA A-fragment images (launch_rows_pack)
    const void* himg;       // [PF_BLOCKS][PS_HEAD_IMAGE]: head rows [8b, 8b+8) as A fragments
    const float* norms;     // [2 * n_layer + 1][1024]
    const float* scales;    // row scales [n_layer][PF_BLOCKS][48]: FS_FP8 handles: the slow persistent kernel's (the images hold the e4m3 weights widened to bf16); bf16 handles: ones
    const float* hscales;   // head row scales [PF_BLOCKS][8] (FS_FP8: the quantiser's; bf16: ones)
    int n_layer, n_head_rows;
    const float* cos_t;
    const float* sin_t;
    float eps;
    float* x;               // [R][1024] in: embedded input of each row; out: its pre-norm hidden state
    float* logits;          // [R][PR_LD] out: audio-range logits
    const SeqState* state;  // [R]: pos, rope_off, done (a row with done != 0 is skipped)
    void* kv_pool;          // bf16, paged (same pool as the other kernels)
    size_t layer_half;
    const int* page_table;  // [R][pt_stride]
    int pt_stride;
    int n_sl;               // token slices per (row, head) of the attention stage: R * n_sl <= 16
    unsigned long long* edges;  // [PF_RING][PF_REPL][R][PS_EDGE_CAP]
    unsigned long long* prof;
    uint32_t* ctl;          // [0] epoch, [1] timeouts
    int naps[6];
};

struct RowsFastArgs {
    const void* wpack;          // the batch-1 fast image (launch_fast_persist_pack): W13 fragments stay in registers; W2 is streamed from it
    const uint32_t* rowpairs;   // [PF_BLOCKS][40][512]: the image's row-pair dwords, dword-major (launch_rows_pack_rowpairs), streamed
    const float* scales;        // row scales of the fast image [PF_BLOCKS][PF_SCL]: FS_FP8 handles: FastPersistArgs::scales (bf16-widened e4m3 image); bf16 handles: ones
    const float* norms[2 * PF_LAYERS + 1];
    const void* fast_emb;
    const float* qkv0_tbl;      // null, or the batch-1 fast decoder's layer-0 qkv table f32 [1024][1280] (lm_persist.h): codebook passes 1..7 skip layer 0's S1
    const void* tok_emb;
    const void* cb_emb;
    const float* cos_t;
    const float* sin_t;
    float eps;
    const float* xf;            // [R][1024] hidden states of the slow transformer
    const float* slow_logits;   // [R][PR_LD]
    int n_slow;
    float* cap;                 // null, or [R][cap_frames][9][2048] decision capture
    int cap_frames;
    float* x;                   // [R][1024] out: embedded input of each row's next slow step
    SeqState* state;            // [R]
    const SampleCfg* cfg;       // [R]
    const int* budget;          // [R] generator iterations allowed (a row is done when frame reaches it)
    RngState* rng;              // [R] StdRng stream positions (sampled rows: up to 9 words per frame)
    float* rp_mask;             // [R][8][1024]
    int* rp_ring;               // [R][8][17]
    int* rp_meta;               // [R][8][2]
    uint32_t* out_codes;        // [R][8][out_cap]
    int out_cap;
    unsigned long long* edges;  // [PF_RING][PF_REPL][R][PF_EDGE_CAP]
    unsigned long long* prof;
    uint32_t* ctl;
    int naps[6];
    int nap_draw;               // sampled instantiation: nap in front of the decision-edge sweep (a draw takes ~7 us on the drawing workgroups)
};

size_t rows_slow_edge_bytes(int R);
size_t rows_fast_edge_bytes(int R);
// re-lays the slow blocks + the audio-range head into MFMA A-fragment images (device to device, once per weight load)
void launch_rows_pack(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, void* wimg, void* himg, hipStream_t st);
void launch_rows_pack_fp8(const LayerW* layers, int n_layer, const void* head_w, int n_head_rows, void* wimg, void* himg, hipStream_t st);  // FS_FP8 matrices -> the same (bf16) images
void launch_rows_pack_rowpairs(const void* fast_pack, void* out /*PF_BLOCKS * 40 * 512 * 4 bytes*/, hipStream_t st);
void launch_rows_slow(const RowsSlowArgs& a, int R, hipStream_t st);   // R in {2, 4, 8}
void launch_rows_fast(const RowsFastArgs& a, int R, bool sampled, hipStream_t st);   // R in {1, 2, 4}; sampled: every row temp > 0, 0 < top_k <= 256

}  // namespace fs
