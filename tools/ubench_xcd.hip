// Micro-benchmark (round 5): what does an all-gather stage cost when it stays INSIDE one XCD (32 workgroups, through that
// XCD's L2) compared with the product's 256-way cross-XCD stage (sc1 write-through granules, 8 replicas)?
//
// 256 workgroups x 512 threads, one per CU (96 KB of LDS requested), exactly the geometry of k_slow_persist.  Every workgroup reads its XCC id
// (s_getreg_b32 HW_REG_XCC_ID) and takes a rank inside its XCD from a per-XCD atomic counter -- no assumption about block -> XCD placement.
// A chain of dependent MINIMAL stages (sweep 1024 values, block sum, publish): chain time / stages = what one stage costs with no arithmetic.
//   mode G : the product's stage -- every workgroup publishes 4 values x 8 replicas with sc1 stores, sweeps its replica with sc1 loads
//   mode X : each XCD runs its own chain -- a workgroup publishes 32 values into its XCD's buffer, the 32 workgroups of the XCD sweep it
//            store flavour: plain (line stays in the XCD's L2) or sc1 (write-through: the line is dropped from L2, MI355X guide)
//   mode M : the proposed slow-transformer layer: in-edges [G, X, X, X, G] per 5 stages (x all-gather, qkv, attention partials, h, activations)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_xcd.hip -o tools/ubench_xcd.bin ; run: tools/ubench_xcd.bin [stages]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int NB = 256, NT = 512, N = 1024, RING = 4, REPL = 8;
constexpr unsigned SPIN_MAX = 1u << 20;

__device__ __forceinline__ float mix(float s, int idx, int e) {
    unsigned h = __float_as_uint(s) * 2654435761u + (unsigned)idx * 40503u + (unsigned)e * 97u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return 0.5f + (float)(h >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ u32x4 ld16_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st8_plain(void* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_sc1(void* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_sc0sc1(void* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 15u);
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

struct Args {
    u64* gbuf;      // [RING][REPL][N]   global edges (sc1)
    u64* xbuf;      // [8 xcd][RING][N]  XCD-local edges
    unsigned* ctr;  // [8] monotonic rank counters (+ [8] launch epoch)
    float* out;     // [NB]
    unsigned* info; // [NB] xcc << 8 | rank ; [NB] timeouts
    unsigned long long* clk;  // [2] workgroup 0: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) of the whole chain
    int stages, mode, store_kind, nap;
    unsigned tag_base;
};

// pattern of in-edge kinds (1 = XCD-local) for stage s
__device__ __forceinline__ bool in_is_x(int mode, int s) { return mode == 1 ? true : (mode == 2 ? ((s % 5) >= 1 && (s % 5) <= 3) : false); }

__global__ __launch_bounds__(NT) void k_chain(Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    __shared__ int s_rank;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int xcc = xcc_id();
    if (tid == 0) s_rank = (int)(atomicAdd(A.ctr + xcc, 1u) & 31u);
    __syncthreads();
    const int rank = s_rank;
    const int vb = rank * 8 + xcc;  // virtual block id: this workgroup's share of a global edge
    const int rep = xcc;            // replica swept by this XCD
    bool fail = false;
    float S = 0.f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned ge = 0, xe = 0;  // edges consumed so far of each kind
    for (int s = 0; s < A.stages && !fail; ++s) {
        const bool ix = in_is_x(A.mode, s), ox = in_is_x(A.mode, s + 1);
        for (int i = 0; i < A.nap; ++i) __builtin_amdgcn_s_sleep(1);
        // ---- sweep: unit tid = granules 2 tid, 2 tid + 1
        const u64* src = ix ? A.xbuf + ((size_t)xcc * RING + (xe & 3)) * N : A.gbuf + ((size_t)(ge & 3) * REPL + rep) * N;
        const unsigned tag = A.tag_base + (ix ? xe : ge) + 1u;
        u32x4 v;
        for (unsigned spins = 0;; ++spins) {
            v = ld16_sc1(src + 2 * tid);
            if ((v.y == tag && v.w == tag)) break;
            if (spins > SPIN_MAX) { fail = true; break; }
        }
        if (ix) ++xe; else ++ge;
        S = block_sum(__uint_as_float(v.x) + __uint_as_float(v.z), red);
        // ---- publish
        if (ox) {  // 32 values per workgroup into the XCD's buffer
            if (tid < 32) {
                const int idx = rank * 32 + tid;
                u64* dst = A.xbuf + ((size_t)xcc * RING + (xe & 3)) * N + idx;
                const u64 g = ((u64)(A.tag_base + xe + 1u) << 32) | __float_as_uint(mix(S, idx, s));
                if (A.store_kind == 0) st8_plain(dst, g); else if (A.store_kind == 1) st8_sc1(dst, g); else st8_sc0sc1(dst, g);
            }
        } else {   // 4 values x 8 replicas, write-through
            if (tid < 32) {
                const int idx = vb * 4 + (tid & 3), rr = tid >> 2;
                u64* dst = A.gbuf + ((size_t)(ge & 3) * REPL + rr) * N + idx;
                st8_sc1(dst, ((u64)(A.tag_base + ge + 1u) << 32) | __float_as_uint(mix(S, idx, s)));
            }
        }
    }
    if (tid == 0) {
        A.out[b] = S;
        A.info[b] = (unsigned)(xcc << 8 | rank);
        A.info[NB + b] = fail ? 1u : 0u;
        if (b == 0) { A.clk[0] = clock64() - c0; A.clk[1] = wall_clock64() - w0; }
    }
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs; chains of %d dependent minimal stages (sweep 1024 values, block sum, publish), 256 workgroups x 512 threads\n", p.gcnArchName,
           p.multiProcessorCount, stages);
    Args A;
    CK(hipMalloc(&A.gbuf, sizeof(u64) * RING * REPL * N)); CK(hipMalloc(&A.xbuf, sizeof(u64) * 8 * RING * N));
    CK(hipMalloc(&A.ctr, 64)); CK(hipMemset(A.ctr, 0, 64));
    CK(hipMalloc(&A.out, 4 * NB)); CK(hipMalloc(&A.info, 8 * NB)); CK(hipMalloc(&A.clk, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned tag_base = 0;
    bool printed_map = false;
    const char* mname[3] = {"G  256-way cross-XCD (product)", "X  32-way inside each XCD", "M  layer = [G, X, X, X, G]"};
    const char* sname[3] = {"plain", "sc1", "sc0 sc1"};
    for (int mode = 0; mode < 3; ++mode)
        for (int sk = 0; sk < (mode == 0 ? 1 : 3); ++sk)
            for (int nap : {0, 2, 4, 6, 8, 12, 16}) {
                float best = 1e30f; unsigned tmo = 0; bool agree = true;
                for (int rep = 0; rep < 3; ++rep) {
                    // edge 0 of both kinds: ones with tag_base + 1
                    std::vector<u64> hg((size_t)RING * REPL * N, 0), hx((size_t)8 * RING * N, 0);
                    const float one = 1.f; unsigned u; memcpy(&u, &one, 4);
                    for (int r = 0; r < REPL; ++r) for (int i = 0; i < N; ++i) hg[(size_t)r * N + i] = ((u64)(tag_base + 1) << 32) | u;
                    for (int x = 0; x < 8; ++x) for (int i = 0; i < N; ++i) hx[(size_t)x * RING * N + i] = ((u64)(tag_base + 1) << 32) | u;
                    CK(hipMemcpy(A.gbuf, hg.data(), hg.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(A.xbuf, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
                    A.stages = stages; A.mode = mode; A.store_kind = sk; A.nap = nap; A.tag_base = tag_base;
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(k_chain, dim3(NB), dim3(NT), 96 * 1024, 0, A);
                    CK(hipEventRecord(e1, 0));
                    CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    std::vector<float> ho(NB); std::vector<unsigned> hi(2 * NB);
                    CK(hipMemcpy(ho.data(), A.out, 4 * NB, hipMemcpyDeviceToHost)); CK(hipMemcpy(hi.data(), A.info, 8 * NB, hipMemcpyDeviceToHost));
                    for (int i = 0; i < NB; ++i) { tmo += hi[NB + i]; agree &= ho[i] == ho[0]; }
                    if (!printed_map) {
                        int cnt[16] = {0}, match = 0;
                        for (int i = 0; i < NB; ++i) { cnt[hi[i] >> 8]++; match += (int)(hi[i] >> 8) == (i % 8); }
                        printf("workgroups per XCC id:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
                        printf("   blocks with xcc == blockIdx %% 8: %d of %d\n", match, NB);
                        printed_map = true;
                    }
                    tag_base += (unsigned)stages + 8u;
                }
                unsigned long long hc[2]; CK(hipMemcpy(hc, A.clk, 16, hipMemcpyDeviceToHost));
                printf("  mode %-32s stores %-8s nap %2d: %6.3f us/stage%s  (all workgroups agree=%d, timeouts=%u; shader clock during the chain %.0f MHz)\n", mname[mode],
                       mode == 0 ? "sc1" : sname[sk], nap, best * 1e3f / stages, mode == 2 ? "  (x5 = per layer)" : "", (int)agree, tmo, (double)hc[0] / ((double)hc[1] * 0.01));
            }
    return 0;
}
