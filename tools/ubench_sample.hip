// Micro-benchmark of the on-device sampler (k_sample_slow: n candidates, one block): 64 graph nodes, us/node for greedy
// and sampled configurations.  Build: tools/build_ubench.sh.  usage: ubench_sample.bin [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "fs_common.h"
#include "lm_kernels.h"
using namespace fs;
#ifdef FS_SAMPLE_DBG
namespace fs { void fs_dbg_read_ts(unsigned long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float time_graph(hipStream_t st, int nodes, int reps, const std::function<void(int)>& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) enqueue(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * nodes);
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    ModelDims d{1024, 4096, 16, 2, 64, 8, 1e-6f};
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<float> lg(4096);
    uint64_t z = 12345;
    for (auto& v : lg) { z = z * 6364136223846793005ull + 1442695040888963407ull; v = (float)((z >> 33) % 20000) / 2000.0f - 5.f; }
    float *dl, *x, *xf; CK(hipMalloc(&dl, 4096 * 4)); CK(hipMemcpy(dl, lg.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, 4096)); CK(hipMalloc(&xf, 4096)); CK(hipMemset(x, 0, 4096));
    SampleCfg* cfg; CK(hipMalloc(&cfg, sizeof(SampleCfg)));
    RngState* rng; CK(hipMalloc(&rng, sizeof(RngState))); CK(hipMemset(rng, 1, sizeof(RngState)));
    SeqState* state; CK(hipMalloc(&state, sizeof(SeqState))); CK(hipMemset(state, 0, sizeof(SeqState)));
    struct { const char* name; float temp, top_p; int top_k; } cases[] = {
        {"greedy", 0.f, 1.f, 0}, {"t0.7 p0.8 k256", 0.7f, 0.8f, 256}, {"t0.7 p0.9 k50", 0.7f, 0.9f, 50}, {"t0.7 p0.8 k0", 0.7f, 0.8f, 0},
        {"t0.7 p1.0 k256", 0.7f, 1.0f, 256}};
    printf("sampler, n = %d candidates\n", n);
    for (auto& cs : cases) {
        SampleCfg c = {}; c.temp = cs.temp; c.top_p = cs.top_p; c.top_k = cs.top_k; c.rep_pen = 1.f; c.ignore_eos = 1; c.im_end_id = 100011;
        c.sem_lo = 100012; c.sem_hi = 101035;
        CK(hipMemcpy(cfg, &c, sizeof(c), hipMemcpyHostToDevice));
        CK(hipMemset(state, 0, sizeof(SeqState)));
        float us = time_graph(st, 64, 20, [&](int) { SampleKernels<bf16_t>::sample_slow(d, dl, n, cfg, rng, state, x, xf, st); });
        printf("  %-18s %7.2f us/node\n", cs.name, us);
#ifdef FS_SAMPLE_DBG
        unsigned long long ts[64]; fs::fs_dbg_read_ts(ts);
        printf("     cycles: softmax %lld | bisect %lld | compact %lld | sort %lld | chains %lld | zero %lld | barrier %lld | pick %lld\n",
               (long long)(ts[1]-ts[0]), (long long)(ts[2]-ts[1]), (long long)(ts[3]-ts[2]), (long long)(ts[4]-ts[3]), (long long)(ts[5]-ts[4]),
               (long long)(ts[6]-ts[5]), (long long)(ts[7]-ts[6]), (long long)(ts[8]-ts[7]));
#endif
    }
    return 0;
}
