// Phase profile of the block-parallel sampler (csrc/lm_bsample_dev.h, BS_PROF stamps): 512 threads, n candidates, top-k / top-p, one block.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fish-speech.rs_amd/csrc tools/ubench_bsample.hip -o tools/ubench_bsample.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define BS_PROF 1
#ifndef BS_CH
#define BS_CH 8
#endif
namespace fs {
#include "lm_bsample_dev.h"
template <int EPT>
__global__ __launch_bounds__(512) void k_bs(const float* __restrict__ logits, int n, int kk, float inv_t, float top_p, uint32_t word, int reps, int* out) {
    __shared__ BSampLds S;
    float lv[EPT];
#pragma unroll
    for (int s = 0; s < EPT; ++s) { const int i = threadIdx.x * EPT + s; lv[s] = i < n ? logits[i] : 0.f; }
    int used = 0, r = 0;
    for (int i = 0; i < reps; ++i) r += bsample<512, EPT, BS_CH>(lv, n, kk, inv_t, top_p, word + i, &used, S);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}
}  // namespace fs
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    std::vector<float> lg(4096);
    unsigned long long z = 12345;
    for (auto& v : lg) { z = z * 6364136223846793005ull + 1442695040888963407ull; v = (float)((z >> 33) % 20000) / 2000.0f - 5.f; }
    float* dl; int* out;
    CK(hipMalloc(&dl, 4096 * 4)); CK(hipMemcpy(dl, lg.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMalloc(&out, 1024));
    struct { const char* name; float temp, top_p; int k; } cs[] = {{"t0.7 p0.8 k256", 0.7f, 0.8f, 256}, {"t0.7 p0.9 k50", 0.7f, 0.9f, 50}, {"t0.7 p1.0 k256", 0.7f, 1.0f, 256}, {"t0.02 p0.8 k256 (peaked)", 0.02f, 0.8f, 256}};
    for (auto& c : cs) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 200;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            if (n <= 1024) hipLaunchKernelGGL((fs::k_bs<2>), dim3(1), dim3(512), 0, 0, dl, n, c.k, 1.0f / c.temp, c.top_p, 0x9e3779b9u, reps, out);
            else hipLaunchKernelGGL((fs::k_bs<4>), dim3(1), dim3(512), 0, 0, dl, n, c.k, 1.0f / c.temp, c.top_p, 0x9e3779b9u, reps, out);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long ts[16];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(fs::g_bs_ts), sizeof(ts)));
        printf("n %d  %-26s %6.2f us per call | clocks (last call, shader clock): A softmax %llu  B select %llu  C keep %llu  D sum||rank %llu  E top-p %llu  F draw %llu\n",
               n, c.name, ms * 1e3f / reps, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5]);
    }
    return 0;
}
