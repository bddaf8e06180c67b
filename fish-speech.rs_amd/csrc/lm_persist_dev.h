// Device-side building blocks shared by the persistent decode kernels (lm_persist.hip: fast decoder, lm_persist_slow.hip: slow
// transformer): bf16 unpacking at the use site, multi-value wave reductions, and the in-launch edge protocol -- 8-byte {f32 value, tag}
// granules published with relaxed agent-scope (sc1, write-through) stores and swept with 16-byte sc1 loads (MI355X guide, Guideline
// 16 R2).  Include inside namespace fs { namespace { ... } }.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define PF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned PF_SPIN_MAX = 1u << 17;  // ~0.1 s of polling before a thread gives up

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
// bf16 pair -> two f32, AT THE USE SITE: as plain C++ the shifts / masks of all 168 weight dwords get placed right behind the
// top-of-pass pin (336 live floats, spills); volatile asm keeps them behind the sweep of the stage that consumes them
__device__ __forceinline__ void pf_unpack(uint32_t w, float& lo, float& hi) {
    asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2" : "=&v"(lo), "=v"(hi) : "v"(w));
}
__device__ __forceinline__ float pf_dot2(uint32_t w, float c0, float c1, float acc) {
    float lo, hi;
    pf_unpack(w, lo, hi);
    acc = fmaf(lo, c0, acc);
    return fmaf(hi, c1, acc);
}
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// 1 / sqrt(mean(x^2) + eps) over 1024 elements (candle_nn::RmsNorm divides by the square root; v_rsq_f32 is within 1 ulp of that
// quotient -- the publishing lanes sit on the critical path of every edge, and an IEEE sqrt + divide is ~60 dependent instructions)
__device__ __forceinline__ float pf_rms_inv(float sumsq, float eps) { return __builtin_amdgcn_rsqf(fmaf(sumsq, 1.0f / 1024.f, eps)); }
// candle silu = x / (1 + exp(-x)), with the hardware reciprocal (1 ulp)
__device__ __forceinline__ float pf_silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <int CTRL>
__device__ __forceinline__ float pf_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int PF_XOR1 = 0xB1, PF_XOR2 = 0x4E, PF_HALF_MIRROR = 0x141, PF_MIRROR = 0x140;

// Sum of N per-lane values over the 64 lanes of a wave, all N at once: levels [lane^32, lane^16, row mirror, half mirror, xor 2,
// xor 1]; while more than one value is alive a level HALVES the value set (the lanes on either side keep different values and
// exchange the other half), afterwards it is a plain butterfly.  Returns the total of value index
//   N = 32: lane >> 1;   N = 16: lane >> 2;   N = 8: lane >> 3;   N = 4: lane >> 4     (every lane of the group holds it).
template <int N>
__device__ __forceinline__ float pf_reduce(float (&v)[N], int lane) {
    static_assert(N == 4 || N == 8 || N == 16 || N == 32, "value counts used by the kernel");
    int n = N;
    // level lane^32
    {
        const int h = n / 2;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + h]), false, false);
            v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        n = h;
    }
    // level lane^16
    {
        const int h = n / 2;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + h]), false, false);
            v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        n = h;
    }
    if (N == 4) {  // one value left: butterfly over the 16 lanes of the row
        float t = v[0];
        t += pf_dpp<PF_MIRROR>(t); t += pf_dpp<PF_HALF_MIRROR>(t); t += pf_dpp<PF_XOR2>(t); t += pf_dpp<PF_XOR1>(t);
        return t;
    }
    // N >= 8: n = N / 4 values (8, 4 or 2) alive
    {   // row mirror: lanes 0..7 <-> 15..8
        const bool pred = (lane & 8) != 0;
        const int h = n / 2;
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
            const float keep = pred ? v[i + h] : v[i], send = pred ? v[i] : v[i + h];
            v[i] = keep + pf_dpp<PF_MIRROR>(send);
        }
        n = h;
    }
    if (N == 8) {  // one value left
        float t = v[0];
        t += pf_dpp<PF_HALF_MIRROR>(t); t += pf_dpp<PF_XOR2>(t); t += pf_dpp<PF_XOR1>(t);
        return t;
    }
    {   // half mirror: lanes 0..3 <-> 7..4
        const bool pred = (lane & 4) != 0;
        const int h = n / 2;
#pragma unroll
        for (int i = 0; i < N / 16; ++i) {
            const float keep = pred ? v[i + h] : v[i], send = pred ? v[i] : v[i + h];
            v[i] = keep + pf_dpp<PF_HALF_MIRROR>(send);
        }
        n = h;
    }
    if (N == 16) {  // one value left
        float t = v[0];
        t += pf_dpp<PF_XOR2>(t); t += pf_dpp<PF_XOR1>(t);
        return t;
    }
    {   // N == 32: two values left; xor 2 halves them
        const bool pred = (lane & 2) != 0;
        const float keep = pred ? v[1] : v[0], send = pred ? v[0] : v[1];
        float t = keep + pf_dpp<PF_XOR2>(send);
        t += pf_dpp<PF_XOR1>(t);
        return t;
    }
}

__device__ __forceinline__ float pf_wave_sum(float v) {
    v += pf_dpp<PF_XOR1>(v); v += pf_dpp<PF_XOR2>(v); v += pf_dpp<PF_HALF_MIRROR>(v); v += pf_dpp<PF_MIRROR>(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

// Sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), left in all four with identical bits (both levels add the same
// two operands on either side).  The publishing wave of a stage uses it to add the 8 wave partials behind the stage's block barrier: lane
// (value r = lane & 15, row k = lane >> 4) adds partials k and k + 4 from LDS, the rows meet through two v_permlane swaps, and row k stores
// edge replicas k and k + 4 -- r consecutive granules per replica and store instruction, as before.  The chain behind the barrier is 2 LDS
// reads + 3 adds + 2 swaps instead of 8-24 dependent LDS reads + adds on a lone wave (round 5: a stage's epilogue measured ~0.35 us of
// dependent issue latency; an 8-lanes-per-value DPP variant shortened it as well but scattered the write-through stores over 8 replicas per
// value and the NEXT stage's wait grew by more than the epilogue lost, profiles/r05_stage_profile.txt).
__device__ __forceinline__ float pf_sum_rows(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

// value barrier: what is derived from the result is not loop-invariant, so per-lane addresses / predicates are recomputed per stage
// (a few VALU ops) instead of being hoisted out of the pass loop into ~150 extra live registers
__device__ __forceinline__ int pf_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// A consumer knows that an edge cannot be complete before its producers have swept the previous edge, computed and published: polling
// from the first instant only adds sweep traffic (every workgroup re-reads 8-32 KB per round) in front of the granules everybody is
// waiting for.  Each stage therefore naps before its first sweep (units of 64 clocks, tuned per stage kind: profiles/r03_poll_naps.txt).
__device__ __forceinline__ void pf_nap_before_sweep(int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

// ---- edge sweeps: NL 16-byte units per lane (unit u = granules 2u, 2u+1), all of a lane's loads in flight, retried until both
// tags of every unit match.  `dead` latches after a timeout: the thread then stops waiting for anything.
__device__ __forceinline__ bool pf_tags_ok(const u32x4& v, unsigned tag) { return v.y == tag && v.w == tag; }

__device__ __forceinline__ void pf_sweep1(const u64* base, int unit, unsigned tag, u32x4& v, bool& dead, uint32_t* ctl) {
    const u64* p = base + 2 * (size_t)unit;
    for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (pf_tags_ok(v, tag) || dead) return;
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }
    }
}
__device__ __forceinline__ void pf_sweep2(const u64* base, int unit0, int unit1, unsigned tag, u32x4& v0, u32x4& v1, bool& dead,
                                          uint32_t* ctl) {
    const u64 *p0 = base + 2 * (size_t)unit0, *p1 = base + 2 * (size_t)unit1;
    for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
        if ((pf_tags_ok(v0, tag) && pf_tags_ok(v1, tag)) || dead) return;
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }
    }
}
__device__ __forceinline__ void pf_sweep3(const u64* base, int unit0, int unit1, int unit2, unsigned tag, u32x4& v0, u32x4& v1, u32x4& v2, bool& dead,
                                          uint32_t* ctl) {
    const u64 *p0 = base + 2 * (size_t)unit0, *p1 = base + 2 * (size_t)unit1, *p2 = base + 2 * (size_t)unit2;
    for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
        if ((pf_tags_ok(v0, tag) && pf_tags_ok(v1, tag) && pf_tags_ok(v2, tag)) || dead) return;
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }
    }
}
// sum over the 32 lanes of a half-wave, left in every one of them (lane ^ 16 through v_permlane16_swap, then the 16-lane row by DPP)
__device__ __forceinline__ float pf_allsum32(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    t += pf_dpp<PF_MIRROR>(t); t += pf_dpp<PF_HALF_MIRROR>(t); t += pf_dpp<PF_XOR2>(t); t += pf_dpp<PF_XOR1>(t);
    return t;
}
__device__ __forceinline__ void pf_sweep4(const u64* base, int tid, unsigned tag, u32x4 (&v)[4], bool& dead, uint32_t* ctl) {
    const u64 *p0 = base + 2 * (size_t)tid, *p1 = p0 + 2 * PF_THREADS, *p2 = p1 + 2 * PF_THREADS, *p3 = p2 + 2 * PF_THREADS;
    for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        if ((pf_tags_ok(v[0], tag) && pf_tags_ok(v[1], tag) && pf_tags_ok(v[2], tag) && pf_tags_ok(v[3], tag)) || dead) return;
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }
    }
}

__device__ __forceinline__ void pf_publish(u64* edges, unsigned e, int rep, int index, unsigned tag, float value) {
    gu64* g = (gu64*)(edges + ((size_t)(e & (PF_RING - 1)) * PF_REPL + rep) * PF_EDGE_CAP + index);
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(value), PF_RLX_AGENT);
}

