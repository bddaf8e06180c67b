// Phase profile of the block-parallel sampler (lm_bsample_dev.h, 512 threads): cycles per phase on flat and peaked logit rows.
// Build: hipcc --offload-arch=gfx950 -O3 -DBS_PROF -I fish-speech.rs_amd/csrc tools/ubench_bsample.hip -o tools/ubench_bsample.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
namespace fs {
#include "lm_bsample_dev.h"
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_t(const float* logits, int n, int kk, float inv_t, float top_p, uint32_t word, int* out, int reps) {
    __shared__ BSampLds S;
    float lv[EPT];
    for (int s = 0; s < EPT; ++s) { const int i = threadIdx.x * EPT + s; lv[s] = i < n ? logits[i] : 0.f; }
    int used, r = 0;
    for (int it = 0; it < reps; ++it) r += bsample<NT, EPT>(lv, n, kk, inv_t, top_p, word + it, &used, S);
    if (threadIdx.x == 0) out[0] = r;
}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    float* d; int* o; CK(hipMalloc(&d, 2048 * 4)); CK(hipMalloc(&o, 4));
    struct { const char* name; int n; float scale, temp, top_p; int k; } cs[] = {
        {"flat   n1024 t0.7 p0.8 k256", 1024, 1.f, 0.7f, 0.8f, 256}, {"peaked n1024 t0.7 p0.8 k256", 1024, 4.f, 0.7f, 0.8f, 256},
        {"flat   n1024 t0.7 p0.9 k50 ", 1024, 1.f, 0.7f, 0.9f, 50},  {"flat   n1024 t1.0 p0.3 k256", 1024, 1.f, 1.0f, 0.3f, 256},
        {"flat   n2037 t0.7 p0.8 k256", 2037, 1.f, 0.7f, 0.8f, 256}};
    for (auto& c : cs) {
        std::vector<float> h(2048);
        unsigned z = 12345;
        for (auto& v : h) { float a = 0; for (int i = 0; i < 12; ++i) { z = z * 1664525u + 1013904223u; a += (float)(z >> 8) / 16777216.f; } v = (a - 6.f) * c.scale; }
        CK(hipMemcpy(d, h.data(), 2048 * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 200;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            if (c.n <= 1024) hipLaunchKernelGGL((fs::k_t<512, 2>), dim3(1), dim3(512), 0, 0, d, c.n, c.k, 1.f / c.temp, c.top_p, 0x9e3779b9u, o, reps);
            else hipLaunchKernelGGL((fs::k_t<512, 4>), dim3(1), dim3(512), 0, 0, d, c.n, c.k, 1.f / c.temp, c.top_p, 0x9e3779b9u, o, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long ts[16]; CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(fs::g_bs_ts), sizeof(ts)));
        printf("%s: %6.2f us/call | cycles: softmax %llu  select %llu  compact %llu  sum||rank %llu  top-p %llu  draw %llu\n", c.name, ms * 1e3 / reps,
               ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5]);
    }
    return 0;
}
