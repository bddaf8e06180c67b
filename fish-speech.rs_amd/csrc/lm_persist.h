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
constexpr int PF_LDS_CHUNKS = PF_LAYERS * 4;              // w2 of every layer lives in LDS: 16 x 16 B per lane
constexpr int PF_CHUNKS = PF_REG_CHUNKS + PF_LDS_CHUNKS;  // 58 x 16 B x 512 lanes = 475 KB per workgroup

struct FastPersistArgs {
    const void* wpack;          // [PF_BLOCKS][PF_CHUNKS][PF_THREADS] x 16 B: per-lane weight image (launch_fast_persist_pack)
    const float* norms[2 * PF_LAYERS + 1];  // attention_norm l, ffn_norm l (l = 0..3), fast_norm: f32 [1024]
    const void* fast_emb;       // bf16 [1024][1024]
    const void* tok_emb;        // bf16 [V][1024]
    const void* cb_emb;         // bf16 [8 * 1024][1024]
    const float* cos_t;         // [max_seq_len][32] (rows 0..7 used: RoPE position = codebook index, dual_ar.rs:651-655)
    const float* sin_t;
    float eps;
    const float* xf;            // [1024] hidden state of the slow transformer (input of codebook pass 0)
    float* x;                   // [1024] out: embedded input of the next slow step
    SeqState* state;
    const SampleCfg* cfg;
    RepPenState rp;
    uint32_t* out_codes;
    int out_cap;
    unsigned long long* edges;  // [PF_RING][PF_REPL][PF_EDGE_CAP] granules (zeroed once at allocation)
    unsigned long long* prof;   // null, or [16]: workgroup 0 accumulates 10 ns ticks per stage kind (FISHRT_PERSIST_PROF=1)
    uint32_t* ctl;              // [0] launch counter (tag epoch), [1] spin-timeout count (host checks it), [2] launches with temp != 0 (refused)
};

// true when the model has the geometry the kernel is written for
bool fast_persist_supported(const ModelDims& d, int n_fast_layer, int n_cb, int cb_size);
size_t fast_persist_pack_bytes();
size_t fast_persist_edge_bytes();
// re-lays the four fast blocks' matrices + fast_output into the per-lane image (device to device, once per weight load)
void launch_fast_persist_pack(const LayerW* fast, const void* head_w, void* pack, hipStream_t st);
void launch_fast_persist(const FastPersistArgs& a, hipStream_t st);
// self-test hook of the multi-value wave reductions: out[w][i] = sum over the 64 lanes of in[lane][i] for N in {4, 16, 32}
void launch_pf_reduce_selftest(const float* in /*[64][32]*/, float* out /*[3][32]*/, hipStream_t st);

}  // namespace fs
