// Micro-benchmark (round 6, VERDICT r5 item 1: "measure one prototype stage at 32 rows first and commit the number"): what does ONE dependent
// all-to-all stage of a persistent fast decoder cost when it carries 32 activation rows instead of 1..4?
//
// 256 workgroups x 512 threads, one per CU (the geometry of k_fast_persist / k_fast_rows).  A chain of dependent stages; in every stage a
// workgroup (1) waits until all 256 producers of the previous stage have signalled, (2) reads the WHOLE stage input (32 rows x K values: what a
// row-parallel GEMM stage with resident weights needs; 128 KB for K = 1024, 512 KB for the W2 stage's K = 4096) into registers, (3) does a token
// amount of arithmetic (sum), one block barrier, (4) publishes its share of the output (32 rows x 4 or 16 values) and signals.
// Two hand-off protocols:
//   mode G: the product's tagged granules -- 8-byte {value, tag} sc1 stores, consumers sweep 16-byte sc1 loads and retry (lm_persist_dev.h); the input of
//           a stage is 2 x its payload (tags), every load goes to the memory side
//   mode F: bulk data + one flag per producer -- payload as packed f32 with sc1 (write-through) stores, s_waitcnt vmcnt(0), then an 8-byte flag;
//           consumers poll the 256 flags (one 16-byte sc1 load per lane of every wave), then read the payload with (F1) sc1 loads or (F2) buffer_inv sc1 +
//           plain loads (the first workgroup of an XCD pulls a line into that XCD's L2, the other 31 hit it)
// Output: us per stage for payload K in {256, 1024, 4096} values per row x 32 rows.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stage32.hip -o tools/ubench_stage32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int NB = 256, NT = 512, ROWS = 32, RING = 4;
constexpr unsigned SPIN_MAX = 1u << 18;

__device__ __forceinline__ u32x4 ld16_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_sc1(void* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

struct Args {
    unsigned char* data;   // [RING][ROWS * Kmax * 8] payload (mode G: granules; mode F: f32)
    u64* flags;            // [RING][NB]
    float* out;            // [NB]
    unsigned* fails;       // [NB]
    unsigned long long* clk;
    int stages, mode, K, nap;
    unsigned tag_base;
};

// K values per row; this workgroup produces K / NB values per row (x 32 rows) per stage and consumes all 32 x K
template <int MODE>  // 0 = G, 1 = F1 (sc1 payload loads), 2 = F2 (buffer_inv + plain loads)
__global__ __launch_bounds__(NT) void k_chain(Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = A.K;
    const size_t vals = (size_t)ROWS * K;                               // payload values of a stage
    const size_t bytes_in = MODE == 0 ? vals * 8 : vals * 4;            // what a consumer reads
    const int n16 = (int)(bytes_in / ((size_t)NT * 16));               // 16-byte loads per lane
    bool fail = false;
    float S = 1.f;
    const unsigned long long w0 = wall_clock64();
    for (int s = 0; s < A.stages && !fail; ++s) {
        const unsigned tag = A.tag_base + (unsigned)s + 1u;
        const unsigned char* in = A.data + (size_t)(s & 3) * ((size_t)ROWS * 4096 * 8);
        for (int i = 0; i < A.nap; ++i) __builtin_amdgcn_s_sleep(1);
        float acc = 0.f;
        if (MODE == 0) {
            // tagged granules: 16 loads in flight per round, each retried until both tags match
            for (int j0 = 0; j0 < n16 && !fail; j0 += 8) {
                u32x4 v[8];
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const void* p = in + ((size_t)(j0 + j) * NT + tid) * 16;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(p) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(v[j])); ok &= (v[j].y == tag && v[j].w == tag); }
                    if (ok) break;
                    if (spins > SPIN_MAX) { fail = true; break; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += __uint_as_float(v[j].x) + __uint_as_float(v[j].z);
            }
        } else {
            // flags: every wave polls all 256 (lane l: flags 4l .. 4l+3 as two 16-byte loads)
            const u64* fl = A.flags + (size_t)(s & 3) * NB;
            for (unsigned spins = 0;; ++spins) {
                const u32x4 f0 = ld16_sc1(fl + 4 * lane), f1 = ld16_sc1(fl + 4 * lane + 2);
                const bool ok = f0.x == tag && f0.z == tag && f1.x == tag && f1.z == tag;
                if (__all(ok)) break;
                if (spins > SPIN_MAX) { fail = true; break; }
            }
            if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
            for (int j0 = 0; j0 < n16; j0 += 16) {
                u32x4 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const void* p = in + ((size_t)(j0 + j) * NT + tid) * 16;
                    if (j0 + j < n16) {
                        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(p) : "memory");
                        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(p) : "memory");
                    } else v[j] = u32x4{0, 0, 0, 0};
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    asm volatile("" : "+v"(v[j]));
                    acc += (__uint_as_float(v[j].x) + __uint_as_float(v[j].y)) + (__uint_as_float(v[j].z) + __uint_as_float(v[j].w));
                }
            }
        }
        // token arithmetic: block sum (one barrier, as every product stage has)
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) red[(s & 1) * 8 + wave] = acc;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[(s & 1) * 8 + i];
        S = t;
        // expected: every value of stage s's input is (s % 7) + 1
        if (t != (float)vals * (float)((s % 7) + 1)) fail = true;
        // ---- publish this workgroup's share of the next stage's input: ROWS x (K / NB) values = ((s + 1) % 7) + 1
        unsigned char* outp = A.data + (size_t)((s + 1) & 3) * ((size_t)ROWS * 4096 * 8);
        const unsigned ntag = tag + 1u;
        const float val = (float)(((s + 1) % 7) + 1);
        const int per = (int)(vals / NB);  // values this workgroup writes
        if (MODE == 0) {
            // granule i of this workgroup at index b * per + i (16-byte pairs per lane)
            if (tid < per / 2) {
                u32x4 g = {__float_as_uint(val), ntag, __float_as_uint(val), ntag};
                st16_sc1(outp + ((size_t)b * per + 2 * tid) * 8, g);
            }
        } else {
            if (tid < per / 4) {
                u32x4 g = {__float_as_uint(val), __float_as_uint(val), __float_as_uint(val), __float_as_uint(val)};
                st16_sc1(outp + ((size_t)b * per + 4 * tid) * 4, g);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st8_sc1(A.flags + (size_t)((s + 1) & 3) * NB + b, ((u64)0 << 32) | ntag);
        }
    }
    if (tid == 0) {
        A.out[b] = S;
        A.fails[b] = fail ? 1u : 0u;
        if (b == 0) A.clk[0] = wall_clock64() - w0;
    }
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs; chains of %d dependent stages, 256 workgroups x 512 threads; a stage = wait for 256 producers, read 32 rows x K values, block sum, publish 32 x K/256\n",
           p.gcnArchName, p.multiProcessorCount, stages);
    Args A;
    const size_t dbytes = (size_t)RING * ROWS * 4096 * 8;
    CK(hipMalloc(&A.data, dbytes)); CK(hipMalloc(&A.flags, sizeof(u64) * RING * NB));
    CK(hipMalloc(&A.out, 4 * NB)); CK(hipMalloc(&A.fails, 4 * NB)); CK(hipMalloc(&A.clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    unsigned tag_base = 0;
    const char* mname[3] = {"G  tagged granules, sc1 sweeps", "F1 flags + sc1 payload loads", "F2 flags + buffer_inv sc1 + plain loads"};
    for (int K : {256, 1024, 4096})
        for (int mode = 0; mode < 3; ++mode)
            for (int nap : {0, 8, 16, 32}) {
                float best = 1e30f; unsigned nfail = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    // stage 0's input: value 1 everywhere, tag / flag = tag_base + 1
                    const size_t vals = (size_t)ROWS * K;
                    std::vector<unsigned char> h(dbytes, 0);
                    const float one = 1.f; unsigned u; memcpy(&u, &one, 4);
                    if (mode == 0) { u64* g = reinterpret_cast<u64*>(h.data()); for (size_t i = 0; i < vals; ++i) g[i] = ((u64)(tag_base + 1) << 32) | u; }
                    else { unsigned* g = reinterpret_cast<unsigned*>(h.data()); for (size_t i = 0; i < vals; ++i) g[i] = u; }
                    std::vector<u64> hf((size_t)RING * NB, 0);
                    for (int i = 0; i < NB; ++i) hf[i] = (u64)(tag_base + 1);
                    CK(hipMemcpy(A.data, h.data(), dbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(A.flags, hf.data(), hf.size() * 8, hipMemcpyHostToDevice));
                    A.stages = stages; A.mode = mode; A.K = K; A.nap = nap; A.tag_base = tag_base;
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(NB), dim3(NT), 96 * 1024, 0, A);
                    else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(NB), dim3(NT), 96 * 1024, 0, A);
                    else hipLaunchKernelGGL(k_chain<2>, dim3(NB), dim3(NT), 96 * 1024, 0, A);
                    CK(hipEventRecord(e1, 0));
                    CK(hipDeviceSynchronize());
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    std::vector<unsigned> fl(NB); CK(hipMemcpy(fl.data(), A.fails, 4 * NB, hipMemcpyDeviceToHost));
                    for (unsigned f : fl) nfail += f;
                    best = ms < best ? ms : best;
                    tag_base += (unsigned)stages + 8u;
                }
                printf("  K %4d (%3zu KB read per workgroup%s)  mode %-40s nap %2d: %6.2f us/stage  (failed workgroups: %u)\n", K,
                       (size_t)ROWS * K * (mode == 0 ? 8 : 4) / 1024, mode == 0 ? ", tags included" : "", mname[mode], nap, best * 1e3f / stages, nfail);
            }
    return 0;
}
