// Micro-benchmark of the MFMA row path (rows_layer: k_prep, k_gemm3 / k_gemm_big, attention) stage by stage: graph of 24 layers over distinct weights.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <vector>
#include "fs_common.h"
#include "lm_kernels.h"
using namespace fs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float time_graph(hipStream_t st, int nodes, int reps, const std::function<void(int)>& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    enqueue(0); CK(hipStreamSynchronize(st));  // lazy attribute setup outside capture
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) enqueue(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / (reps * nodes);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32;
    typedef bf16_t WT;
    ModelDims d{1024, 4096, 16, 2, 64, 8, 1e-6f};
    const int NL = 24, QKV = 1280;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t per_layer = (size_t)(QKV * 1024 + 1024 * 1024 + 2 * 4096 * 1024 + 1024 * 4096) * sizeof(WT);
    uint8_t* arena; CK(hipMalloc(&arena, per_layer * NL)); CK(hipMemset(arena, 0x3c, per_layer * NL));
    float* norm; CK(hipMalloc(&norm, 4096)); std::vector<float> ones(1024, 1.0f); CK(hipMemcpy(norm, ones.data(), 4096, hipMemcpyHostToDevice));
    std::vector<LayerW> lw(NL);
    for (int l = 0; l < NL; ++l) { uint8_t* b = arena + per_layer * l; lw[l].wqkv = b; b += (size_t)QKV * 2048; lw[l].wo = b; b += (size_t)2 << 20; lw[l].w13 = b; b += (size_t)16 << 20; lw[l].w2 = b; lw[l].attn_norm = norm; lw[l].ffn_norm = norm; }
    RowsCtx c;
    c.Mcap = 2048; c.part_rows = 512; c.down_split = 4;
    auto dalloc = [&](size_t n) { void* p; CK(hipMalloc(&p, n)); CK(hipMemset(p, 0, n)); return p; };
    const int NC = 8192 / LmKernels<WT>::attn_chunk();
    c.X = (float*)dalloc(2048 * 1024 * 4); c.Q = (float*)dalloc(2048 * 1024 * 4); c.part = (float*)dalloc((size_t)512 * 16 * NC * 66 * 4);
    c.P = (float*)dalloc((size_t)4 * 2048 * 1024 * 4); c.A = (uint16_t*)dalloc(2 * 2048 * 1024 * 2);
    c.C = (uint16_t*)dalloc((size_t)2 * 2048 * 4096 * 2); c.A2 = (uint16_t*)dalloc(2 * 2048 * 1024 * 2); c.ss = (float*)dalloc(2048 * 64 * 4);
    c.cos_t = (float*)dalloc(8192 * 32 * 4); c.sin_t = (float*)dalloc(8192 * 32 * 4);
    SeqState hs = {}; hs.pos = getenv("UB_POS") ? atoi(getenv("UB_POS")) : 300; SeqState* state = (SeqState*)dalloc(sizeof(SeqState)); CK(hipMemcpy(state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    c.state = state; c.n_chunks_max = NC; c.nc_launch = argc > 2 ? atoi(argv[2]) : 4; c.pos_step = argc > 3 ? atoi(argv[3]) : 1; c.pt_stride = 0;
    c.chunked_attn = getenv("UB_CHUNKED_ATTN") != nullptr;
    c.seq_rows = argc > 4 ? atoi(argv[4]) : 0;  // > 0: group prefill (M must be a multiple); all sequences share one page table here
    if (c.seq_rows) { hs.pos = 0; CK(hipMemcpy(state, &hs, sizeof(hs), hipMemcpyHostToDevice)); }
    const int max_pages = 128; const size_t page_elems = 2 * KV_PAGE * 64;
    WT* kvpool = (WT*)dalloc((size_t)NL * 2 * max_pages * page_elems * sizeof(WT));
    std::vector<int> pt(max_pages); for (int i = 0; i < max_pages; ++i) pt[i] = i;
    int* d_pt = (int*)dalloc(max_pages * 4); CK(hipMemcpy(d_pt, pt.data(), max_pages * 4, hipMemcpyHostToDevice));
    auto kv = [&](int l) { KVView v; v.k = kvpool + (size_t)l * 2 * max_pages * page_elems; v.v = (WT*)v.k + max_pages * page_elems; v.page_table = d_pt; return v; };
    float us = time_graph(st, NL, 10, [&](int i) { LmKernels<WT>::rows_layer(d, M, c, lw[i % NL], kv(i % NL), i == 0, st); });
    static const char* names[8] = {"prep(attn_norm)", "gemm qkv+rope+kv", "attention", "attn combine", "gemm wo+res", "prep(ffn_norm)", "gemm w13+swiglu", "gemm w2 slabs"};
    for (int sI = 0; sI < 8; ++sI) {
        c.stage_mask = 1u << sI;
        float t = time_graph(st, NL, 10, [&](int i) { LmKernels<WT>::rows_layer(d, M, c, lw[i % NL], kv(i % NL), false, st); });
        printf("  stage %d %-18s %7.2f us/node\n", sI, names[sI], t);
    }
    c.stage_mask = 0xFFu;
    printf("rows_layer M=%d: %.1f us per layer (8 nodes) -> %.1f us/node; 24 layers = %.2f ms\n", M, us, us / 8, us * 24 / 1e3);
    return 0;
}
