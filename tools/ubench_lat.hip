// Micro-benchmark: the latency primitives behind an in-launch hand-off on this box.
//   (1) dependent-load round trip of one lane: plain (L1 hit), sc1 (agent scope: L1 bypass), sc0 sc1, nt; small buffer
//   (2) ping-pong of an 8-byte {value, tag} granule between two workgroups (same XCD: blocks 0 and 8; other XCD: 0 and 1):
//       one-way hand-off latency = round trip / 2, with sc1 stores + sc1 polling loads
//   (3) the same ping-pong with N pollers on the consumer side issuing staggered polls (does polling harder help?)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lat.hip -o tools/ubench_lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int KIND>
__device__ __forceinline__ unsigned ld(const unsigned* p) {
    unsigned v;
    if (KIND == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (KIND == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (KIND == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (KIND == 3) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (KIND == 4) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int KIND>
__global__ void k_chase(const unsigned* buf, int n, int iters, u64* out) {
    if (threadIdx.x != 0) return;
    unsigned idx = 0;
    for (int i = 0; i < 64; ++i) idx = ld<KIND>(buf + idx);
    const u64 t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) idx = ld<KIND>(buf + idx);
    const u64 t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = idx;
}

// block `a` and block `b` bounce a granule; everybody else exits
__global__ void k_pingpong(gu64* g, int a, int b, int iters, u64* out) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    gu64* mine = g + (me == a ? 0 : 64);    // written by me (different 512-B regions)
    gu64* theirs = g + (me == a ? 64 : 0);
    u64 t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 17) t0 = wall_clock64();
        if (me == a) {
            __hip_atomic_store(mine, ((u64)i << 32) | (unsigned)i, RLX_AGENT);
            unsigned spins = 0;
            while ((__hip_atomic_load(theirs, RLX_AGENT) >> 32) != (u64)i) if (++spins > (1u << 24)) return;
        } else {
            unsigned spins = 0;
            while ((__hip_atomic_load(theirs, RLX_AGENT) >> 32) != (u64)i) if (++spins > (1u << 24)) return;
            __hip_atomic_store(mine, ((u64)i << 32) | (unsigned)i, RLX_AGENT);
        }
    }
    if (me == a) { out[0] = wall_clock64() - t0; out[1] = iters - 16; }
}

// XCD id of every block (to pick same-XCD / other-XCD pairs from what the hardware actually did)
__global__ void k_census(int* xcc) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        xcc[blockIdx.x] = (int)(v & 0xF);
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    u64* out; CK(hipMalloc(&out, 64)); u64 h[2];
    printf("dependent single-lane load round trip (ns), 10 ns clock:\n");
    for (int n : {256, 1 << 16, 1 << 22}) {   // 1 KB (one CU's L1), 256 KB (L2), 16 MB
        std::vector<unsigned> hb(n);
        for (int i = 0; i < n; ++i) hb[i] = (unsigned)(((u64)i * 40503u + 977u) % n);
        // make it one cycle over a stride pattern
        for (int i = 0; i < n; ++i) hb[i] = (unsigned)((i + (n >= 4096 ? 1031 * 16 : 17)) % n);
        unsigned* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemcpy(buf, hb.data(), n * 4, hipMemcpyHostToDevice));
        const int iters = 4000;
        const char* names[5] = {"plain", "sc1", "sc0 sc1", "nt", "sc0"};
        float ns[5];
        hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(64), 0, st, buf, n, iters, out); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); ns[0] = h[0] * 10.0f / iters;
        hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(64), 0, st, buf, n, iters, out); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); ns[1] = h[0] * 10.0f / iters;
        hipLaunchKernelGGL(k_chase<2>, dim3(1), dim3(64), 0, st, buf, n, iters, out); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); ns[2] = h[0] * 10.0f / iters;
        hipLaunchKernelGGL(k_chase<3>, dim3(1), dim3(64), 0, st, buf, n, iters, out); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); ns[3] = h[0] * 10.0f / iters;
        hipLaunchKernelGGL(k_chase<4>, dim3(1), dim3(64), 0, st, buf, n, iters, out); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); ns[4] = h[0] * 10.0f / iters;
        printf("  buffer %8d B:", n * 4);
        for (int k = 0; k < 5; ++k) printf("  %s %.0f", names[k], ns[k]);
        printf("\n");
        CK(hipFree(buf));
    }
    int* xcc; CK(hipMalloc(&xcc, 256 * 4));
    hipLaunchKernelGGL(k_census, dim3(256), dim3(64), 0, st, xcc);
    std::vector<int> hx(256); CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
    printf("XCC id of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf("\n");
    gu64* g; CK(hipMalloc((void**)&g, 4096));
    for (int pair = 0; pair < 4; ++pair) {
        const int a = 0, b = pair == 0 ? 8 : pair == 1 ? 1 : pair == 2 ? 4 : 128;
        CK(hipMemset((void*)g, 0, 4096));
        hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(64), 0, st, g, a, b, 2016, out);
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("ping-pong blocks %d (xcc %d) <-> %d (xcc %d): one-way %.0f ns\n", a, hx[a], b, hx[b], h[0] * 10.0 / h[1] / 2);
    }
    return 0;
}
