// Experiment: does warming the NEXT kernel's weights (same block index => same XCD L2) from inside the current kernel
// shorten the dependent-node chain?  Chain of GEMV nodes over distinct weight regions; variant A: plain; variant B: each
// block also loads (and discards) the region block b of the next node will read.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool PF, bool NT>
__global__ __launch_bounds__(256) void k_gemv(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ y,
                                              const u32x4* __restrict__ Wnext, int* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32x4* row = W + (size_t)wave * 128;
    u32x4 a = NT ? __builtin_nontemporal_load(row + lane) : row[lane];
    u32x4 b = NT ? __builtin_nontemporal_load(row + 64 + lane) : row[64 + lane];
    u32x4 p0, p1;
    if (PF) {  // this block's 4 rows of the next matrix (8 KB): 2 x 16 B per thread
        const u32x4* nx = Wnext + (size_t)blockIdx.x * 512;
#ifdef PF_ASM
        // fire-and-forget: results never read, no s_waitcnt emitted by the compiler for them
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(p0) : "v"(nx + threadIdx.x));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(p1) : "v"(nx + 256 + threadIdx.x));
#else
        p0 = nx[threadIdx.x]; p1 = nx[256 + threadIdx.x];
#endif
    }
    float xr[16];
    for (int i = 0; i < 16; ++i) xr[i] = x[(i / 8) * 512 + lane * 8 + (i % 8)];
    float acc = 0.f;
    unsigned int v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    for (int i = 0; i < 8; ++i) acc += __uint_as_float(v[i] << 16) * xr[2 * i] + __uint_as_float(v[i] & 0xFFFF0000u) * xr[2 * i + 1];
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) y[wave] = acc;
#ifndef PF_ASM
    if (PF) { if ((p0.x ^ p1.x) == 0x12345677u && p0.y == 0x7654321u) sink[0] = 1; }
#endif
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
    return ms * 1e3f / (reps * nodes);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t total_rows = 1 << 20;
    u32x4* W; CK(hipMalloc(&W, total_rows * 2048)); CK(hipMemset(W, 0x3c, total_rows * 2048));
    float *x, *y; int* sink; CK(hipMalloc(&x, 4096)); CK(hipMalloc(&y, total_rows * 4)); CK(hipMemset(x, 0, 4096)); CK(hipMalloc(&sink, 64));
    for (int rows : {1024, 4096, 8192}) {
        const int blocks = rows / 4;
        auto reg = [&](int i) { return W + (((size_t)i * rows) % (total_rows - 2 * rows)) * 128; };
        float a = time_graph(st, 256, 20, [&](int i) { hipLaunchKernelGGL((k_gemv<false, true>), dim3(blocks), dim3(256), 0, st, reg(i), x, y, reg(i + 1), sink); });
        float b = time_graph(st, 256, 20, [&](int i) { hipLaunchKernelGGL((k_gemv<true, true>), dim3(blocks), dim3(256), 0, st, reg(i), x, y, reg(i + 1), sink); });
        float c = time_graph(st, 256, 20, [&](int i) { hipLaunchKernelGGL((k_gemv<true, false>), dim3(blocks), dim3(256), 0, st, reg(i), x, y, reg(i + 1), sink); });
        float d = time_graph(st, 256, 20, [&](int i) { hipLaunchKernelGGL((k_gemv<false, false>), dim3(blocks), dim3(256), 0, st, reg(i), x, y, reg(i + 1), sink); });
        printf("%5d rows (%5.1f MB): plain nt %.2f | prefetch-next + nt own %.2f | prefetch-next + cached own %.2f | plain cached %.2f us/node\n",
               rows, rows * 2048 / 1e6, a, b, c, d);
    }
    return 0;
}
