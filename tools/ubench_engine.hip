// Micro-benchmark / skeleton of a persistent decode "layer engine": what does ONE dependent GEMV stage cost when the whole
// chain of stages runs inside one launch (256 workgroups, one per CU, 512 threads), the stage's weights are prefetched into
// VGPRs one stage ahead (in the shadow of the previous all-to-all edge) and the stage output travels as 8-byte {value, tag}
// granules that every workgroup sweeps?
//
// Stage s (per workgroup):  [issue the nt loads of stage s+1's weight slice: W bytes per CU]  ->  sweep the N_in granules of edge s
// (16-B sc1 loads = 2 granules, all of a thread's loads in flight)  ->  values to LDS, barrier  ->  K-split GEMV of the
// workgroup's `rows` output rows against the weights already in registers (wave w owns K range w), DPP wave sums, 8 partials per
// row meet in LDS, barrier  ->  `rows` lanes publish (R replicas: replica r is read by the workgroups with blockIdx % R == r).
//
// Reported: us per stage for several (N_in, rows, W) shapes, R in {1, 8}; "local" = same code with the sweep reading this
// workgroup's private, pre-tagged buffer (no cross-CU dependency: the compute + barrier floor of a stage).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_engine.hip -o tools/ubench_engine.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NT = 512, NW = NT / 64;
static int g_nb = 256;   // participating workgroups (runtime: how does the edge cost scale with the number of CUs that must all be on time?)
constexpr int NMAX = 4096;
constexpr unsigned SPIN_MAX = 1u << 20;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
__device__ __forceinline__ float readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return (readlane(v, 15) + readlane(v, 31)) + (readlane(v, 47) + readlane(v, 63));
}

// NL = 16-B sweep loads per thread (N_in = NT * 2 * NL granules); ROWS = output rows per workgroup; WL = 16-B weight loads per thread
// per stage (W = NT * 16 * WL bytes per CU); R = replicas; LOCAL = no cross-CU dependency
template <int NL, int ROWS, int WL, int R, bool LOCAL, int PM = 0, int KP = 4, int DS = 5>
__global__ __launch_bounds__(NT) void k_engine(gu64* gran, int stages, const u32x4* __restrict__ wbuf, size_t w_n16, float* out, unsigned* tmo, unsigned long long* prof, int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);              // [N_in]
    float* red = xs + NMAX;                                   // [NW][ROWS]
    constexpr int N_IN = NT * 2 * NL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int rep = b % R;
    // edge buffers: ring of 4, each [R][NMAX] granules
    auto edge = [&](int e, int r) { return gran + ((size_t)(e & 3) * R + r) * NMAX; };
    u32x4 wcur[WL > 0 ? WL : 1], wnext[WL > 0 ? WL : 1];
#pragma unroll
    for (int j = 0; j < WL; ++j) wcur[j] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bool fail = false;
    float last = 0.f;
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, npoll = 0;
    size_t woff = ((size_t)b * NT + tid);
    for (int s = 0; s < stages && !fail; ++s) {
        // (1) next stage's weights: W bytes per CU, coalesced 16 B per lane, non-temporal
#pragma unroll
        for (int j = 0; j < WL; ++j) wnext[j] = __builtin_nontemporal_load(wbuf + (woff + (size_t)j * NB * NT) % w_n16);
        woff += (size_t)WL * NB * NT + 4099 * 64;
        const unsigned long long c0 = clock64();
        // (2) sweep edge s
        const unsigned tag = (unsigned)s + 1u;
        gu64* g = LOCAL ? gran + (size_t)(4 * R + b) * NMAX : edge(s, rep);
        u32x4 v[NL];
        unsigned spins = 0;
        if (PM == 0) {
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const gu64* p = g + (size_t)(j * NT + tid) * 2;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(p) : "memory");
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[j]) :: "memory");  // ties the data to the wait
#pragma unroll
            for (int j = 0; j < NL; ++j) ok &= LOCAL || (v[j].y == tag && v[j].w == tag);
            ++npoll;
            if (ok) break;
            if (++spins > SPIN_MAX) { fail = true; break; }
        }
        } else {
            // staggered polls landing in LDS (LDS-DMA, sc1): KP sweeps in flight DS x 64 clocks apart; a sweep that comes back too early
            // costs one stagger period instead of a full round trip, and sweeps still in flight after the hit land in slots nobody reads
            u32x4* land = reinterpret_cast<u32x4*>(smem + 32 * 1024) + (size_t)wave * KP * NL * 64;
            auto issue = [&](int i) {
#pragma unroll
                for (int j = 0; j < NL; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)(j * NT + tid) * 2),
                                                     (__attribute__((address_space(3))) void*)(land + ((i % KP) * NL + j) * 64), 16, 0, 16);
            };
#pragma unroll
            for (int i = 0; i < KP; ++i) { issue(i); if (i + 1 < KP) __builtin_amdgcn_s_sleep(DS); }
            for (int c = 0;; ++c) {
                if (KP == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * NL) : "memory");
                else if (KP == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NL) : "memory");
                bool ok = true;
#pragma unroll
                for (int j = 0; j < NL; ++j) { v[j] = land[((c % KP) * NL + j) * 64 + lane]; ok &= LOCAL || (v[j].y == tag && v[j].w == tag); }
                ++npoll;
                if (ok) break;
                if (++spins > SPIN_MAX) { fail = true; break; }
                issue(c + KP);
            }
        }
        const unsigned long long c1 = clock64();
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            xs[(j * NT + tid) * 2] = __uint_as_float(v[j].x);
            xs[(j * NT + tid) * 2 + 1] = __uint_as_float(v[j].z);
        }
        __syncthreads();
        const unsigned long long c2 = clock64();
        // (3) K-split GEMV: wave w owns K range [w * N_IN / NW, ...), lane owns KL consecutive elements; weights: register words
        // reused round-robin (the arithmetic volume of the real stage: ROWS * N_IN MACs per workgroup)
        constexpr int KL = N_IN / NT;
        float xr[KL];
#pragma unroll
        for (int i = 0; i < KL; ++i) xr[i] = xs[tid * KL + i];
        float acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                const unsigned wv = WL > 0 ? wcur[((r * KL + i) / 2 / 4) % (WL > 0 ? WL : 1)][((r * KL + i) / 2) % 4] : 0x3f803f80u;
                const float wf = ((r * KL + i) & 1) ? __uint_as_float(wv & 0xFFFF0000u) : __uint_as_float(wv << 16);
                a = fmaf(wf * (1.0f / 4096.f), xr[i], a);
            }
            acc[r] = wave_sum(a);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) red[wave * ROWS + r] = acc[r];
        }
        const unsigned long long c3 = clock64();
        __syncthreads();
        const unsigned long long c4 = clock64();
        // (4) publish: PER values per workgroup into each of the R replicas; consecutive lanes -> consecutive granules of one replica
        {
            const int PER = N_IN / NB;
            for (int idx = tid; idx < PER * R; idx += NT) {
                const int j = idx % PER, rr = idx / PER, r = j % ROWS;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += red[w * ROWS + r];
                t = 1.0f + t * 1e-3f + (float)((b * ROWS + r) & 7) * 0.125f;
                last = t;
                if (!LOCAL) __hip_atomic_store(edge(s + 1, rr) + (size_t)b * PER + j, ((u64)(tag + 1u) << 32) | __float_as_uint(t), RLX_AGENT);
            }
        }
        const unsigned long long c5 = clock64();
        tp[0] += c1 - c0; tp[1] += c2 - c1; tp[2] += c3 - c2; tp[3] += c4 - c3; tp[4] += c5 - c4;
#pragma unroll
        for (int j = 0; j < WL; ++j) wcur[j] = wnext[j];
    }
    if (tid == 0) { out[b] = last; if (fail) atomicAdd(tmo, 1u); }
    if (tid == 511 && b == 0) { unsigned long long* pr = prof + 8; for (int i = 0; i < 5; ++i) pr[i] = tp[i]; pr[5] = npoll; }
    if (tid == 0 && (b == 0)) { unsigned long long* pr = prof + (b ? 8 : 0); for (int i = 0; i < 5; ++i) pr[i] = tp[i]; pr[5] = npoll; }
}

template <int NL, int ROWS, int WL, int R, bool LOCAL, int PM = 0, int KP = 4, int DS = 5>
static float run(int stages, const u32x4* wbuf, size_t w_n16, bool* ok, bool show = false) {
    unsigned long long* prof; CK(hipMalloc(&prof, 16 * 8)); CK(hipMemset(prof, 0, 128));
    const int NB = g_nb;
    gu64* gran; float* out; unsigned* tmo;
    const size_t gb = (size_t)(4 * R + 256) * NMAX * 8;
    CK(hipMalloc((void**)&gran, gb)); CK(hipMalloc(&out, NB * 4)); CK(hipMalloc(&tmo, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    *ok = true;
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<u64> h(gb / 8, 0);
        const float one = 1.0f; unsigned u; memcpy(&u, &one, 4);
        for (int r = 0; r < R; ++r) for (int i = 0; i < NMAX; ++i) h[(size_t)r * NMAX + i] = ((u64)1 << 32) | u;  // edge 0, every replica
        CK(hipMemcpy((void*)gran, h.data(), gb, hipMemcpyHostToDevice)); CK(hipMemset(tmo, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_engine<NL, ROWS, WL, R, LOCAL, PM, KP, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_engine<NL, ROWS, WL, R, LOCAL, PM, KP, DS>), dim3(NB), dim3(NT), 160 * 1024, st, gran, stages, wbuf, w_n16, out, tmo, prof, NB);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned f; CK(hipMemcpy(&f, tmo, 4, hipMemcpyDeviceToHost));
        if (f) *ok = false;
    }
    if (show) { unsigned long long hp[16]; CK(hipMemcpy(hp, prof, 128, hipMemcpyDeviceToHost));
        for (int k = 0; k < 2; ++k) printf("      [blk 0 tid %d] cycles/stage: poll %.0f  bar1 %.0f  gemv %.0f  bar2 %.0f  publish %.0f ; polls/stage %.2f\n", k ? 511 : 0, hp[8*k+0] / (double)stages, hp[8*k+1] / (double)stages, hp[8*k+2] / (double)stages, hp[8*k+3] / (double)stages, hp[8*k+4] / (double)stages, hp[8*k+5] / (double)stages); }
    CK(hipFree(prof));
    CK(hipFree((void*)gran)); CK(hipFree(out)); CK(hipFree(tmo)); CK(hipStreamDestroy(st));
    return best * 1e3f / stages;
}

#define ROW(NL, ROWS, WL, label)                                                                                         \
    {                                                                                                                    \
        bool o1, o2, o3;                                                                                                 \
        const float a = run<NL, ROWS, WL, 1, false>(stages, wbuf, n16, &o1), b8 = run<NL, ROWS, WL, 8, false>(stages, wbuf, n16, &o2), \
                    l = run<NL, ROWS, WL, 1, true>(stages, wbuf, n16, &o3);                                               \
        bool o4, o5, o6;                                                                                                 \
        const float b32 = run<NL, ROWS, WL, 32, false>(stages, wbuf, n16, &o4), b64 = run<NL, ROWS, WL, 64, false>(stages, wbuf, n16, &o5), \
                    b256 = run<NL, ROWS, WL, 256, false>(stages, wbuf, n16, &o6, true);                                   \
        printf("  %-44s R=1 %6.3f  R=8 %6.3f  R=32 %6.3f  R=64 %6.3f  R=256 %6.3f  local %6.3f us/stage %s\n", label, a, b8, b32, b64, b256, l, (o1 && o2 && o3 && o4 && o5 && o6) ? "" : "TIMEOUT");       \
    }

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t wbytes = 2ull << 30;
    u32x4* wbuf; CK(hipMalloc(&wbuf, wbytes)); CK(hipMemset(wbuf, 0x3c, wbytes));
    const size_t n16 = wbytes / 16;
    printf("persistent stage chain, 256 workgroups x 512 threads, %d stages; weights prefetched one stage ahead into VGPRs\n", stages);
    for (int nb : {2, 8, 32, 64, 128, 256}) {
        g_nb = nb;
        bool o1, o2;
        const float a = run<1, 4, 0, 1, false>(stages, wbuf, n16, &o1), b8 = run<1, 4, 0, 8, false>(stages, wbuf, n16, &o2);
        printf("  %3d workgroups, N_in 1024, resident weights: R=1 %6.3f  R=8 %6.3f us/stage %s\n", nb, a, b8, (o1 && o2) ? "" : "TIMEOUT");
    }
    g_nb = 256;
    {
        bool o;
        printf("  staggered LDS-DMA polls, N_in 1024, resident, R=8, 256 workgroups:\n");
        printf("    single poll (baseline)      %6.3f\n", run<1, 4, 0, 8, false, 0>(stages, wbuf, n16, &o));
        printf("    KP=2 DS=8  (0.22 us apart)  %6.3f %s\n", run<1, 4, 0, 8, false, 1, 2, 8>(stages, wbuf, n16, &o, true), o ? "" : "TIMEOUT");
        printf("    KP=3 DS=5  (0.14 us apart)  %6.3f %s\n", run<1, 4, 0, 8, false, 1, 3, 5>(stages, wbuf, n16, &o, true), o ? "" : "TIMEOUT");
        printf("    KP=4 DS=5                   %6.3f %s\n", run<1, 4, 0, 8, false, 1, 4, 5>(stages, wbuf, n16, &o, true), o ? "" : "TIMEOUT");
        printf("    KP=4 DS=3  (0.08 us apart)  %6.3f %s\n", run<1, 4, 0, 8, false, 1, 4, 3>(stages, wbuf, n16, &o, true), o ? "" : "TIMEOUT");
        printf("    KP=4 DS=8                   %6.3f %s\n", run<1, 4, 0, 8, false, 1, 4, 8>(stages, wbuf, n16, &o, true), o ? "" : "TIMEOUT");
        printf("    N_in 4096: single %6.3f", run<4, 4, 0, 8, false, 0>(stages, wbuf, n16, &o));
        printf("  KP=3 DS=5 %6.3f", run<4, 4, 0, 8, false, 1, 3, 5>(stages, wbuf, n16, &o));
        printf("  KP=2 DS=8 %6.3f\n", run<4, 4, 0, 8, false, 1, 2, 8>(stages, wbuf, n16, &o));
    }
    ROW(1, 4, 0, "N_in 1024,  4 rows/CU, resident weights");
    ROW(1, 4, 1, "N_in 1024,  4 rows/CU,  8 KB/CU/stage ( 2 MB: wo)");
    ROW(1, 32, 0, "N_in 1024, 32 rows/CU, resident weights");
    ROW(1, 32, 8, "N_in 1024, 32 rows/CU, 64 KB/CU/stage (16 MB: w13)");
    ROW(4, 4, 0, "N_in 4096,  4 rows/CU, resident weights");
    ROW(4, 4, 4, "N_in 4096,  4 rows/CU, 32 KB/CU/stage ( 8 MB: w2)");
    return 0;
}
