// Micro-benchmark: what does one dependent kernel node cost inside a hipGraph on this box?
//   (a) trivial 1-thread kernel, (b) 256 blocks x 256 threads doing nothing, (c) a 2 MB GEMV-like streaming kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o gpurun_out/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_wide(int* p) { if (threadIdx.x == 0 && blockIdx.x == 1000000) p[0] += 1; }
// each wave streams `rows` rows of 2 KB (1024 bf16) and reduces them against x held in registers
__global__ __launch_bounds__(256) void k_gemv(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float xr[16];
    for (int i = 0; i < 16; ++i) xr[i] = x[(i / 8) * 512 + lane * 8 + (i % 8)];
    for (int r = 0; r < rows_per_wave; ++r) {
        const u32x4* row = W + (size_t)(wave * rows_per_wave + r) * 128;
        u32x4 a = __builtin_nontemporal_load(row + lane), b = __builtin_nontemporal_load(row + 64 + lane);
        float acc = 0.f;
        unsigned int v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        for (int i = 0; i < 8; ++i) { acc += __uint_as_float(v[i] << 16) * xr[2 * i] + __uint_as_float(v[i] & 0xFFFF0000u) * xr[2 * i + 1]; }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) y[wave * rows_per_wave + r] = acc;
    }
}

template <typename F>
static float time_graph(hipStream_t st, int nodes, int reps, F&& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) enqueue(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * nodes);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    const size_t total_rows = 1 << 20;  // 2 GB of weights: far beyond MALL
    u32x4* W; CK(hipMalloc(&W, total_rows * 2048)); CK(hipMemset(W, 0x3c, total_rows * 2048));
    float *x, *y; CK(hipMalloc(&x, 4096)); CK(hipMalloc(&y, total_rows * 4)); CK(hipMemset(x, 0, 4096));
    printf("graph node cost (us/node), 266-node graphs:\n");
    printf("  trivial <<<1,1>>>            : %.2f\n", time_graph(st, 266, 50, [&](int) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(1), 0, st, d); }));
    printf("  empty   <<<256,256>>>        : %.2f\n", time_graph(st, 266, 50, [&](int) { hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, d); }));
    printf("  empty   <<<1024,256>>>       : %.2f\n", time_graph(st, 266, 50, [&](int) { hipLaunchKernelGGL(k_wide, dim3(1024), dim3(256), 0, st, d); }));
    for (int rows : {1024, 4096, 8192}) {
        for (int rpw : {1, 2, 4}) {
            const int waves = rows / rpw, blocks = (waves + 3) / 4;
            // walk through distinct 2 GB so that nothing is cache resident
            float us = time_graph(st, 256, 20, [&](int i) {
                const size_t off = ((size_t)i * rows) % (total_rows - rows);
                hipLaunchKernelGGL(k_gemv, dim3(blocks), dim3(256), 0, st, W + off * 128, x, y, rpw);
            });
            printf("  gemv %5d rows x 2 KB (%5.1f MB), %d rows/wave, %4d blocks: %.2f us/node -> %.2f TB/s\n", rows, rows * 2048 / 1e6, rpw, blocks, us,
                   rows * 2048.0 / us / 1e6);
        }
    }
    return 0;
}
