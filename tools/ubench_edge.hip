// Micro-benchmark: what does one ALL-TO-ALL dependency edge cost INSIDE a persistent launch on this box?
// (the question behind DESIGN.md "batch-1 frame = 266 seams": a seam as a kernel boundary costs 1.55 us of graph floor
//  + ~1.1 us of ramp; what does the same seam cost as an in-launch all-gather of the op's output vector?)
//
// 256 workgroups (one per CU, forced by a large LDS request) run a chain of `stages` dependent stages.  Stage e: every
// workgroup gathers the N-value vector of edge e (published by all workgroups), reduces it (stand-in for the GEMV), and
// publishes its N/256 values of edge e+1.  Every value depends on every value of the previous edge, so the chain time /
// stages is the full edge latency (publish -> visible -> swept by every CU -> block reduction).
//
// Transport variants:
//   MODE 0  TAG8 : 8-byte {value, tag} granules, relaxed agent-scope (sc1) 64-bit stores / loads (guide: Guideline 16 R2)
//   MODE 1  SENT4: 4-byte values, tag-free: a slot holds the sentinel 0xFFFFFFFF until its producer overwrites it; the
//                  producer re-arms its own slots of the buffer three edges ahead (ring of 4); 16-byte sc1 loads
//   MODE 2  TAG16: 16-byte {v0, v1, v2, tag} granules, one 16-byte sc1 store / load
// BG > 0: BG extra waves per workgroup stream a 1 GB buffer (nt loads) for the whole run = the CU is a "loaded" endpoint.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_edge.hip -o tools/ubench_edge.bin
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

static int NB = 256;              // workgroups (<= CUs)
constexpr int NMAX = 4096;        // values per edge (max)
constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ float mix(float s, int idx, int e) {  // cheap deterministic function of (sum, index, edge), in [0.5, 1.5)
    unsigned h = __float_as_uint(s) * 2654435761u + (unsigned)idx * 40503u + (unsigned)e * 97u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return 0.5f + (float)(h >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ u32x4 ld16_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// block-wide sum in a fixed order (NT threads); every thread gets the result
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

// background streamers (second stream, no LDS, launched after the chain kernel): every wave keeps 8 x 16 B per lane of nt loads
// in flight over a 1 GB buffer until the chain kernel raises *stop
__global__ void k_bg(const u32x4* bgbuf, size_t bg_n16, gu32* stop, float* bg_out, int NB) {
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    float acc = 0.f;
    for (int it = 0; it < (1 << 22); ++it) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(bgbuf + (i + j * nthr) % bg_n16);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __uint_as_float(v[j].x & 0x3F800000u);
        i += nthr * 8;
        if ((it & 7) == 7 && __hip_atomic_load(stop, RLX_AGENT)) break;
    }
    if (acc == 123.f) bg_out[threadIdx.x] = acc;
}

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k_chain(void* bufv, int N, int stages, float* out, unsigned* tmo, gu32* stop, int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, b = blockIdx.x;
    const int per = N / NB;  // values this workgroup publishes per edge
    bool fail = false;
    float S = 0.f;
    for (int e = 0; e < stages && !fail; ++e) {
        float part = 0.f;
        if (MODE == 0) {
            gu64* g = (gu64*)(bufv) + (size_t)(e & 3) * NMAX;
            const unsigned tag = (unsigned)e + 1u;
            for (int i = tid; i < N; i += NT) {
                unsigned spins = 0;
                u64 x;
                while (((x = __hip_atomic_load(g + i, RLX_AGENT)) >> 32) != tag)
                    if (++spins > SPIN_MAX) { fail = true; break; }
                part += __uint_as_float((unsigned)x);
            }
        } else if (MODE == 1) {
            unsigned* g = reinterpret_cast<unsigned*>(bufv) + (size_t)(e & 3) * NMAX;
            for (int i = tid * 4; i < N; i += NT * 4) {
                unsigned spins = 0;
                u32x4 x;
                for (;;) {
                    x = ld16_sc1(g + i);
                    if (x.x != SENT && x.y != SENT && x.z != SENT && x.w != SENT) break;
                    if (++spins > SPIN_MAX) { fail = true; break; }
                }
                part += (__uint_as_float(x.x) + __uint_as_float(x.y)) + (__uint_as_float(x.z) + __uint_as_float(x.w));
            }
        } else {
            u32x4* g = reinterpret_cast<u32x4*>(bufv) + (size_t)(e & 3) * NMAX;
            const unsigned tag = (unsigned)e + 1u;
            for (int i = tid; i < N / 3 + 1; i += NT) {  // granule i carries values 3i .. 3i+2 (the last one is padded with zeros)
                unsigned spins = 0;
                u32x4 x;
                for (;;) {
                    x = ld16_sc1(g + i);
                    if (x.w == tag) break;
                    if (++spins > SPIN_MAX) { fail = true; break; }
                }
                part += __uint_as_float(x.x) + __uint_as_float(x.y) + __uint_as_float(x.z);
            }
        }
        S = block_sum<NT>(part, red);
        // publish this workgroup's `per` values of edge e + 1
        if (MODE == 0) {
            gu64* g = (gu64*)(bufv) + (size_t)((e + 1) & 3) * NMAX;
            if (tid < per) {
                const int idx = b * per + tid;
                __hip_atomic_store(g + idx, ((u64)((unsigned)e + 2u) << 32) | __float_as_uint(mix(S, idx, e)), RLX_AGENT);
            }
        } else if (MODE == 1) {
            unsigned* g = reinterpret_cast<unsigned*>(bufv) + (size_t)((e + 1) & 3) * NMAX;
            unsigned* g3 = reinterpret_cast<unsigned*>(bufv) + (size_t)((e + 3) & 3) * NMAX;
            if (tid < per / 4) {
                const int idx = b * per + tid * 4;
                u32x4 v = {__float_as_uint(mix(S, idx, e)), __float_as_uint(mix(S, idx + 1, e)), __float_as_uint(mix(S, idx + 2, e)),
                           __float_as_uint(mix(S, idx + 3, e))};
                st16_sc1(g + idx, v);
                u32x4 s4 = {SENT, SENT, SENT, SENT};
                st16_sc1(g3 + idx, s4);  // re-arm the slots of edge e + 3 (its previous content, edge e - 1, has been consumed by everybody)
            }
        } else {
            u32x4* g = reinterpret_cast<u32x4*>(bufv) + (size_t)((e + 1) & 3) * NMAX;
            // workgroup b owns granules [b * pg, (b + 1) * pg) with pg = ceil((N/3 + 1) / NB)
            const int ng = N / 3 + 1, pg = (ng + NB - 1) / NB;
            if (tid < pg && b * pg + tid < ng) {
                const int gi = b * pg + tid, idx = gi * 3;
                u32x4 v = {idx < N ? __float_as_uint(mix(S, idx, e)) : 0u, idx + 1 < N ? __float_as_uint(mix(S, idx + 1, e)) : 0u,
                           idx + 2 < N ? __float_as_uint(mix(S, idx + 2, e)) : 0u, (unsigned)e + 2u};
                st16_sc1(g + gi, v);
            }
        }
    }
    if (tid == 0) {
        out[b] = S;
        if (fail) atomicAdd(tmo, 1u);
        if (b == 0) __hip_atomic_store(stop, 1u, RLX_AGENT);
    }
}

// host model of the chain (same arithmetic order: per-thread strided partial sums, wave butterfly, wave order)
template <int NT>
static float host_chain(int N, int stages, int mode) {
    std::vector<float> cur(N), nxt(N);
    for (int i = 0; i < N; ++i) cur[i] = 1.0f;
    float S = 0.f;
    auto mixh = [](float s, int idx, int e) {
        unsigned u; memcpy(&u, &s, 4);
        unsigned h = u * 2654435761u + (unsigned)idx * 40503u + (unsigned)e * 97u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        return 0.5f + (float)(h >> 8) * (1.0f / 16777216.0f);
    };
    for (int e = 0; e < stages; ++e) {
        std::vector<float> part(NT, 0.f);
        if (mode == 0) { for (int t = 0; t < NT; ++t) for (int i = t; i < N; i += NT) part[t] += cur[i]; }
        else if (mode == 1) { for (int t = 0; t < NT; ++t) for (int i = t * 4; i < N; i += NT * 4) part[t] += (cur[i] + cur[i + 1]) + (cur[i + 2] + cur[i + 3]); }
        else { for (int t = 0; t < NT; ++t) for (int i = t; i < N / 3 + 1; i += NT) { float a = 3 * i < N ? cur[3 * i] : 0.f, b2 = 3 * i + 1 < N ? cur[3 * i + 1] : 0.f, c = 3 * i + 2 < N ? cur[3 * i + 2] : 0.f; part[t] += a + b2 + c; } }
        float tot = 0.f;
        for (int w = 0; w < NT / 64; ++w) {
            float v[64];
            for (int l = 0; l < 64; ++l) v[l] = part[w * 64 + l];
            for (int m = 32; m >= 1; m >>= 1) { float t2[64]; for (int l = 0; l < 64; ++l) t2[l] = v[l] + v[l ^ m]; memcpy(v, t2, sizeof(v)); }
            tot += v[0];
        }
        S = tot;
        for (int i = 0; i < N; ++i) nxt[i] = mixh(S, i, e);
        cur.swap(nxt);
    }
    return S;
}

template <int NT, int MODE, int BG>
static void run(const char* name, int N, int stages, const u32x4* bgbuf, size_t bg_n16) {
    void* buf; float* out; unsigned* tmo; float* bgo; unsigned* stop;
    CK(hipMalloc(&stop, 4));
    hipStream_t st2; CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    const size_t gran = MODE == 0 ? 8 : MODE == 1 ? 4 : 16;
    CK(hipMalloc(&buf, 4 * NMAX * gran)); CK(hipMalloc(&out, NB * 4)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&bgo, 4096 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, S = 0.f; unsigned fails = 0; bool same = true;
    for (int rep = 0; rep < 3; ++rep) {
        // (re-)initialise every polled word: edge 0 = ones, the rest unarmed
        std::vector<unsigned char> h(4 * NMAX * gran, MODE == 1 ? 0xFF : 0x00);
        for (int i = 0; i < N; ++i) {
            const float one = 1.0f; unsigned u; memcpy(&u, &one, 4);
            if (MODE == 0) { u64 g = ((u64)1 << 32) | u; memcpy(&h[i * 8], &g, 8); }
            else if (MODE == 1) memcpy(&h[i * 4], &u, 4);
        }
        if (MODE == 2) for (int gi = 0; gi < N / 3 + 1; ++gi) {
            unsigned v[4] = {0, 0, 0, 1};
            const float one = 1.0f; unsigned u; memcpy(&u, &one, 4);
            for (int k = 0; k < 3; ++k) if (3 * gi + k < N) v[k] = u;
            memcpy(&h[gi * 16], v, 16);
        }
        CK(hipMemcpy(buf, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemset(tmo, 0, 4)); CK(hipMemset(out, 0, NB * 4)); CK(hipMemset(stop, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((k_chain<NT, MODE>), dim3(NB), dim3(NT), 100 * 1024, st, buf, N, stages, out, tmo, (gu32*)stop, NB);
        CK(hipEventRecord(e1, st));
        if (BG > 0) hipLaunchKernelGGL(k_bg, dim3(256), dim3(BG * 64), 0, st2, bgbuf, bg_n16, (gu32*)stop, bgo, NB);
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        std::vector<float> ho(NB); CK(hipMemcpy(ho.data(), out, NB * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fails, tmo, 4, hipMemcpyDeviceToHost));
        S = ho[0];
        for (int i = 1; i < NB; ++i) same &= ho[i] == ho[0];
    }
    const float ref = host_chain<NT>(N, stages, MODE);
    printf("  %-34s N=%4d  %6.3f us/edge   (S=%.6g ref=%.6g %s, all-CUs-agree=%d, timeouts=%u)\n", name, N, best * 1e3f / stages, S, ref,
           S == ref ? "OK" : "MISMATCH", (int)same, fails);
    CK(hipFree(buf)); CK(hipFree(out)); CK(hipFree(tmo)); CK(hipFree(bgo));
    CK(hipStreamDestroy(st)); CK(hipStreamDestroy(st2)); CK(hipFree(stop));
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs; %d dependent all-to-all edges per launch, %d workgroups (1 per CU)\n", p.gcnArchName, p.multiProcessorCount, stages, NB);
    const size_t bg_bytes = 1ull << 30;
    u32x4* bg; CK(hipMalloc(&bg, bg_bytes)); CK(hipMemset(bg, 0x3c, bg_bytes));
    const size_t n16 = bg_bytes / 16;
    for (int N : {1024, 4096}) {
        run<256, 0, 0>("TAG8  256 thr, idle CUs", N, stages, bg, n16);
        run<256, 1, 0>("SENT4 256 thr, idle CUs", N, stages, bg, n16);
        run<256, 2, 0>("TAG16 256 thr, idle CUs", N, stages, bg, n16);
        run<512, 0, 0>("TAG8  512 thr, idle CUs", N, stages, bg, n16);
        run<512, 1, 0>("SENT4 512 thr, idle CUs", N, stages, bg, n16);
        run<1024, 0, 0>("TAG8  1024 thr, idle CUs", N, stages, bg, n16);
        run<256, 0, 4>("TAG8  256 thr + 4 streaming waves", N, stages, bg, n16);
        run<256, 1, 4>("SENT4 256 thr + 4 streaming waves", N, stages, bg, n16);
        run<256, 0, 12>("TAG8  256 thr + 12 streaming waves", N, stages, bg, n16);
        run<256, 1, 12>("SENT4 256 thr + 12 streaming waves", N, stages, bg, n16);
    }
    return 0;
}
