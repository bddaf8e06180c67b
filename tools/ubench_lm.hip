// Per-kernel micro-benchmark of the decode kernels (lm_kernels.hip) at Fish-1.5 shapes: each kernel class is captured
// 240x in a hipGraph (cycling over 24 layers' worth of distinct weights so nothing is cache-resident) and replayed;
// prints us/node and the algorithmic GB/s.  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fish-speech.rs_amd/csrc tools/ubench_lm.hip \
//         fish-speech.rs_amd/build/lm_kernels.o -o tools/ubench_lm.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "fs_common.h"
#include "lm_kernels.h"

using namespace fs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float time_graph(hipStream_t st, int nodes, int reps, const std::function<void(int)>& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
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
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * nodes);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 495;  // KV length for the attention kernel
    typedef bf16_t WT;
    ModelDims d{1024, 4096, 16, 2, 64, 8, 1e-6f};
    const int NL = 24, QKV = 1280;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t per_layer = (size_t)(QKV * 1024 + 1024 * 1024 + 2 * 4096 * 1024 + 1024 * 4096) * sizeof(WT);
    uint8_t* arena; CK(hipMalloc(&arena, per_layer * NL)); CK(hipMemset(arena, 0x3c, per_layer * NL));
    float* norm; CK(hipMalloc(&norm, 4096));
    std::vector<float> ones(1024, 1.0f); CK(hipMemcpy(norm, ones.data(), 4096, hipMemcpyHostToDevice));
    std::vector<LayerW> lw(NL);
    for (int l = 0; l < NL; ++l) {
        uint8_t* b = arena + per_layer * l;
        lw[l].wqkv = b; b += (size_t)QKV * 1024 * 2;
        lw[l].wo = b; b += (size_t)1024 * 1024 * 2;
        lw[l].w13 = b; b += (size_t)2 * 4096 * 1024 * 2;
        lw[l].w2 = b;
        lw[l].attn_norm = norm; lw[l].ffn_norm = norm;
    }
    float *x, *q, *act, *part, *logits, *cos_t, *sin_t;
    CK(hipMalloc(&x, 4096)); CK(hipMemcpy(x, ones.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMalloc(&q, 4096)); CK(hipMemset(q, 0, 4096));
    CK(hipMalloc(&act, 4096 * 4)); CK(hipMemset(act, 0, 4096 * 4));
    CK(hipMalloc(&part, 16 * 128 * 66 * 4)); CK(hipMemset(part, 0, 16 * 128 * 66 * 4));
    CK(hipMalloc(&logits, 4096 * 4));
    CK(hipMalloc(&cos_t, 8192 * 32 * 4)); CK(hipMalloc(&sin_t, 8192 * 32 * 4));
    CK(hipMemset(cos_t, 0, 8192 * 32 * 4)); CK(hipMemset(sin_t, 0, 8192 * 32 * 4));
    SeqState hs = {}; hs.pos = T - 1;
    SeqState* state; CK(hipMalloc(&state, sizeof(SeqState))); CK(hipMemcpy(state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    const int max_pages = 128;
    const size_t page_elems = 2 * KV_PAGE * 64;
    WT* kvpool; CK(hipMalloc(&kvpool, (size_t)NL * 2 * max_pages * page_elems * sizeof(WT)));
    CK(hipMemset(kvpool, 0, (size_t)NL * 2 * max_pages * page_elems * sizeof(WT)));
    std::vector<int> pt(max_pages); for (int i = 0; i < max_pages; ++i) pt[i] = i;
    int* d_pt; CK(hipMalloc(&d_pt, max_pages * 4)); CK(hipMemcpy(d_pt, pt.data(), max_pages * 4, hipMemcpyHostToDevice));
    auto kv = [&](int l) { KVView v; v.k = kvpool + (size_t)l * 2 * max_pages * page_elems; v.v = (WT*)v.k + max_pages * page_elems; v.page_table = d_pt; return v; };
    const bool quick = getenv("UBENCH_QUICK") != nullptr;  // few dispatches: for rocprofv3 --pmc passes
    const int N = quick ? 24 : 240, R = quick ? 1 : 20, NC = 8192 / LmKernels<WT>::attn_chunk(); int NCL = 1; while (NCL * LmKernels<WT>::attn_chunk() < T) NCL <<= 1;
    auto report = [&](const char* name, double bytes, float us) { printf("%-34s %7.2f us/node  %8.1f GB/s (%.2f MB)\n", name, us, bytes / us / 1e3, bytes / 1e6); };
    printf("KV length T = %d\n", T);
    report("graph floor (k_advance)", 0, time_graph(st, N, R, [&](int) { launch_advance(state, st); }));
    CK(hipMemcpy(state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    report("qkv  (rmsnorm+GEMV 1280x1024+rope+kv)", QKV * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::qkv(d, x, lw[i % NL], cos_t, sin_t, state, 0, 0, q, kv(i % NL), st); }));
    report("attn_decode (paged, T tokens)", T * 512.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::attn_decode(d, q, kv(i % NL), state, part, NC, NCL, st); }));
    report("wo   (combine + GEMV 1024x1024 + res)", 1024 * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::wo(d, part, NC, NCL, state, nullptr, kv(i % NL), 0, lw[i % NL], x, st); }));
    report("wo fused attn T=4 (fast decoder)", 1024 * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::wo(d, nullptr, 0, 0, nullptr, q, kv(i % NL), 4, lw[i % NL], x, st); }));
    report("ffn_up (rmsnorm+GEMV 8192x1024+swiglu)", 2 * 4096 * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::ffn_up(d, x, lw[i % NL], act, st); }));
    report("ffn_down (GEMV 1024x4096 + res)", 1024 * 4096 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::ffn_down(d, act, lw[i % NL], x, st); }));
    report("head 1024 rows (rmsnorm+GEMV)", 1024 * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::head(d, x, norm, lw[i % NL].wo, nullptr, 1024, logits, st); }));
    report("head 2037 rows (rmsnorm+GEMV)", 2037 * 1024 * 2.0, time_graph(st, N, R, [&](int i) {
        LmKernels<WT>::head(d, x, norm, lw[i % NL].w13, nullptr, 2037, logits, st); }));
    // whole slow layer / fast layer chains
    float us_slow = time_graph(st, NL, R, [&](int i) {
        const LayerW& w = lw[i % NL]; KVView k = kv(i % NL);
        LmKernels<WT>::qkv(d, x, w, cos_t, sin_t, state, 0, 0, q, k, st);
        LmKernels<WT>::attn_decode(d, q, k, state, part, NC, NCL, st);
        LmKernels<WT>::wo(d, part, NC, NCL, state, nullptr, k, 0, w, x, st);
        LmKernels<WT>::ffn_up(d, x, w, act, st);
        LmKernels<WT>::ffn_down(d, act, w, x, st);
    });
    report("SLOW LAYER (5 nodes)", 14944256 * 2.0 + T * 512.0, us_slow);
    float us_fast = time_graph(st, NL, R, [&](int i) {
        const LayerW& w = lw[i % NL]; KVView k = kv(i % NL);
        LmKernels<WT>::qkv(d, x, w, cos_t, sin_t, nullptr, 3, 3, q, k, st);
        LmKernels<WT>::wo(d, nullptr, 0, 0, nullptr, q, k, 4, w, x, st);
        LmKernels<WT>::ffn_up(d, x, w, act, st);
        LmKernels<WT>::ffn_down(d, act, w, x, st);
    });
    report("FAST LAYER (4 nodes)", 14944256 * 2.0, us_fast);
    const double frame_us = 24 * us_slow + 32 * us_fast;
    printf("=> 24 slow + 32 fast layers = %.1f us/frame (+ heads/sampling) -> <= %.0f frames/s\n", frame_us, 1e6 / frame_us);
    return 0;
}
