// LDS-DMA streaming rate per CU: W loader waves per workgroup (one workgroup per CU) stream a private contiguous region of `kb` KiB
// per CU, `reps` times, with global_load_lds_dwordx4 (1 KiB per wave instruction) keeping at most D instructions outstanding per wave.
// Answers: is a CU's LDS-DMA stream limited per wave or per CU, and what does the source (HBM vs cache-resident) change?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsdma.hip -o tools/ubench_ldsdma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int D, bool REG>
__global__ __launch_bounds__(1024) void k_dma(const unsigned char* __restrict__ src, size_t per_cu, int chunks, int reps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * per_cu + (size_t)lane * 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (int c = wave; c < chunks; c += nw) {
            if (REG) {
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)c * 1024));
                acc ^= v;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)c * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + ((c / nw) % 32) * 1024 + wave * 32768), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D) : "memory");
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (REG && acc.x == 0x12345678u) sink[0] = 1.f;
}

template <int D, bool REG>
static void run(const unsigned char* d, size_t per_cu, int W, int kb, int reps, float* sink, const char* what) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<D, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
    for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_dma<D, REG>), dim3(256), dim3(64 * W), 32768 * (W > 4 ? 4 : W), 0, d, per_cu, kb, reps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * kb * 1024.0 * reps;
    printf("  %-28s W=%d D=%2d region %5d KiB/CU: %7.1f GB/s per CU, %6.2f TB/s chip, %6.1f ns per KiB per CU\n", what, W, D, kb, bytes / 256 / (ms * 1e6), bytes / (ms * 1e9), ms * 1e6 / ((double)kb * reps));
}

int main() {
    const size_t per_cu = (size_t)8 << 20;  // 8 MiB per CU: 2 GiB total, beyond the 256 MiB Infinity Cache
    unsigned char* d; CK(hipMalloc(&d, per_cu * 256)); CK(hipMemset(d, 1, per_cu * 256));
    float* sink; CK(hipMalloc(&sink, 4));
    printf("HBM-sized source (8 MiB per CU, read once):\n");
    for (int W : {1, 2, 4}) {
        run<8, false>(d, per_cu, W, 8192, 1, sink, "LDS-DMA");
        run<16, false>(d, per_cu, W, 8192, 1, sink, "LDS-DMA");
        run<48, false>(d, per_cu, W, 8192, 1, sink, "LDS-DMA");
    }
    for (int W : {1, 2, 8}) run<0, true>(d, per_cu, W, 8192, 1, sink, "register loads (nt)");
    printf("cache-resident source (114 KiB per CU, 64 passes):\n");
    for (int W : {1, 2, 4}) { run<16, false>(d, per_cu, W, 114, 64, sink, "LDS-DMA"); run<48, false>(d, per_cu, W, 114, 64, sink, "LDS-DMA"); }
    return 0;
}
