// Edge-sweep experiment (round 3): does the FIRST sweep of an edge have to go to the memory side once per workgroup?
// 256 workgroups x 512 threads, a chain of S dependent all-gather stages (1024 values, {value, tag} granules, 8 replicas; replica r is
// read by the workgroups with blockIdx % 8 == r, i.e. by ONE XCD when workgroups are dealt round-robin).  Per stage: nap -> sweep -> tiny
// reduction + barrier -> 4 lanes x 8 replicas publish.
//   mode 0: sc1 sweep loads on a ring of 4 buffers (the product's protocol: every load goes to the memory side)
//   mode 1: every stage has its OWN buffer (never cached before in this launch); the first attempt uses plain loads, so the 32 workgroups
//           of an XCD share one L2 fill per line; a lane whose tags are not there retries with sc1 loads (its L2 line may be stale)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_edge2.hip -o tools/ubench_edge2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
constexpr int NT = 512, NB = 256, R = 8, N = 1024;

template <int MODE>
__global__ __launch_bounds__(NT) void k_chain(u64* gran, int stages, int nap, float* out, unsigned* tmo, unsigned long long* stat) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, rep = b % R;
    auto buf = [&](int s, int r) { return gran + ((size_t)(MODE == 1 ? s : (s & 3)) * R + r) * N; };
    float last = 0.f;
    unsigned long long retries = 0;
    bool fail = false;
    for (int s = 0; s < stages && !fail; ++s) {
        const unsigned tag = (unsigned)s + 1u;
        for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(1);
        const u64* p = buf(s, rep) + (size_t)tid * 2;
        u32x4 v;
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        unsigned spins = 0;
        while (v.y != tag || v.w != tag) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            ++retries;
            if (++spins > (1u << 20)) { fail = true; break; }
        }
        float a = __uint_as_float(v.x) + __uint_as_float(v.z);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) red[wave] = a;
        __syncthreads();
        if (tid < 4 * R) {
            const int j = tid & 3, rr = tid >> 2;
            float t = 0.f;
            for (int w = 0; w < 8; ++w) t += red[w];
            t = 1.0f + t * 1e-6f + (float)j;
            last = t;
            __hip_atomic_store((gu64*)(buf(s + 1, rr) + (size_t)b * 4 + j), ((u64)(tag + 1u) << 32) | __float_as_uint(t), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (tid == 0) { out[b] = last; if (fail) atomicAdd(tmo, 1u); }
    if (lane == 0) atomicAdd(stat, retries);
}

template <int MODE>
static void run(int stages, int nap) {
    const size_t nbuf = MODE == 1 ? stages + 1 : 4, gb = nbuf * R * N * 8;
    u64* gran; float* out; unsigned* tmo; unsigned long long* stat;
    CK(hipMalloc(&gran, gb)); CK(hipMalloc(&out, NB * 4)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&stat, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; unsigned long long rt = 0; unsigned f = 0;
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<u64> h(gb / 8, 0);
        const float one = 1.0f; unsigned u = *reinterpret_cast<const unsigned*>(&one);
        for (int r = 0; r < R; ++r) for (int i = 0; i < N; ++i) h[(size_t)r * N + i] = ((u64)1 << 32) | u;
        CK(hipMemcpy(gran, h.data(), gb, hipMemcpyHostToDevice)); CK(hipMemset(tmo, 0, 4)); CK(hipMemset(stat, 0, 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain<MODE>, dim3(NB), dim3(NT), 0, 0, gran, stages, nap, out, tmo, stat);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&rt, stat, 8, hipMemcpyDeviceToHost)); }
        CK(hipMemcpy(&f, tmo, 4, hipMemcpyDeviceToHost));
    }
    printf("  mode %d nap %2d: %6.3f us/stage, sc1 retries per (wave, stage) %.2f %s\n", MODE, nap, best * 1e3f / stages,
           (double)rt / ((double)stages * NB * 8), f ? "TIMEOUT" : "");
    CK(hipFree(gran)); CK(hipFree(out)); CK(hipFree(tmo)); CK(hipFree(stat));
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 1000;
    printf("edge chain, 256 workgroups x 512 threads, %d stages, 1024 values, 8 replicas\n", stages);
    for (int nap : {0, 4, 8, 12, 16, 20, 24, 32}) { run<0>(stages, nap); run<1>(stages, nap); }
    return 0;
}
