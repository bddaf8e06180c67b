// Device-side pieces shared by the request-row persistent kernels (lm_persist_rows.hip): the matrix-core GEMV stage (activations of R
// requests as B columns, exact 3-term bf16 split, wave-local swizzled LDS transpose) and edge sweeps that keep up to 16 x 16-byte sc1
// loads of one lane in flight (uniform SGPR bases + per-lane VGPR offsets: an inline-asm statement takes at most 30 operands).
// Include inside namespace fs { namespace { ... } } after lm_persist_dev.h.
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// f32 pair (elements k, k + 1 of one activation vector) -> three bf16 pairs: truncation split, hi + mid + lo == the f32 value exactly
__device__ __forceinline__ void pr_split3(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    const uint32_t h0 = __float_as_uint(a) & 0xFFFF0000u, h1 = __float_as_uint(b) & 0xFFFF0000u;
    const float r0 = a - __uint_as_float(h0), r1 = b - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xFFFF0000u, m1 = __float_as_uint(r1) & 0xFFFF0000u;
    const uint32_t l0 = __float_as_uint(r0 - __uint_as_float(m0)), l1 = __float_as_uint(r1 - __uint_as_float(m1));
    p0 = (h0 >> 16) | h1; p1 = (m0 >> 16) | m1; p2 = (l0 >> 16) | (l1 & 0xFFFF0000u);
}

// B-operand staging of ONE wave for one 128-deep K segment and one column tile: 16 columns x 16 slots of 16 B (8 consecutive k of one
// column) = 4 KB.  Lane t' of the wave owns elements (2 t', 2 t' + 1) of the segment, i.e. dword t' & 3 of slot jq = t' >> 2 of every
// column; MFMA lane (n, q4) reads slot jq = 4 j + q4 of column n for k-step j.  The XOR keeps both sides bank-conflict free (a write
// instruction covers the 16 slots of one column, a read instruction one slot of the 16 columns).
__device__ __forceinline__ int pr_slot(int n, int jq) { return n * 16 + (jq ^ n); }

// one request row's pair -> columns n0, n0 + 1, n0 + 2 (hi, mid, lo) of the wave's staging tile `xt` (dwords)
__device__ __forceinline__ void pr_stage_pair(uint32_t* xt, int n0, int lane, float a, float b) {
    uint32_t p0, p1, p2;
    pr_split3(a, b, p0, p1, p2);
    const int jq = lane >> 2, d = lane & 3;
    xt[pr_slot(n0, jq) * 4 + d] = p0;
    xt[pr_slot(n0 + 1, jq) * 4 + d] = p1;
    xt[pr_slot(n0 + 2, jq) * 4 + d] = p2;
}

// acc[t][c] += A[t][j] . B_c[j] over the 4 k-steps of one 128-deep segment (A: resident / streamed weight fragments of NT 16-row tiles)
template <int NT, int NCT>
__device__ __forceinline__ void pr_mfma_seg(const u32x4* wA /*[NT][stride]*/, int stride, int j0, const u32x4* xt /*[NCT][256]*/, int n, int q4,
                                            f32x4_t (&acc)[NT][NCT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, xt[c * 256 + pr_slot(n, 4 * j + q4)]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wA[t * stride + j0 + j]), bv, acc[t][c], 0, 0, 0);
        }
    }
}

// D -> per-wave row partials: column 3 i of tile c is request row c * RPC + i; rows >= ROWS are padding.  redw: [R][RW] of this wave
template <int NT, int NCT, int RPC, int ROWS, int RW>
__device__ __forceinline__ void pr_extract(const f32x4_t (&acc)[NT][NCT], float* redw, int n, int q4) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i] + pf_dpp<0x101>(acc[t][c][i]) + pf_dpp<0x102>(acc[t][c][i]);  // hi + mid + lo columns
            if (n < 3 * RPC && n % 3 == 0) {
                const int r = c * RPC + n / 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * t + 4 * q4 + i;
                    if (m < ROWS) redw[r * RW + m] = v[i];
                }
            }
        }
}

__device__ __forceinline__ bool pr_ok(const u32x4& v, unsigned tag) { return v.y == tag && v.w == tag; }

// ---- sweeps.  b[i]: uniform base pointers (SGPR pairs), off: per-lane byte offsets.  Every loaded unit's two tags must match.
#define PR_SPIN_TAIL(okexpr)                                                        \
        if ((okexpr) || dead) return;                                               \
        if (spins > PF_SPIN_MAX) { dead = true; atomicAdd(ctl + 1, 1u); return; }

template <int N>
__device__ __forceinline__ void pr_sweep_rows(const u64* const (&b)[N], unsigned off, unsigned tag, u32x4 (&v)[N], bool& dead, uint32_t* ctl) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8, "row counts");
    for (unsigned spins = 0;; ++spins) {
        if constexpr (N == 1) {
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(off), "s"(b[0]) : "memory");
        } else if constexpr (N == 2) {
            asm volatile("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %4 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]) : "v"(off), "s"(b[0]), "s"(b[1]) : "memory");
        } else if constexpr (N == 4) {
            asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %4, %6 sc1\n\t"
                         "global_load_dwordx4 %2, %4, %7 sc1\n\tglobal_load_dwordx4 %3, %4, %8 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(off), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %8, %9 sc1\n\tglobal_load_dwordx4 %1, %8, %10 sc1\n\t"
                         "global_load_dwordx4 %2, %8, %11 sc1\n\tglobal_load_dwordx4 %3, %8, %12 sc1\n\t"
                         "global_load_dwordx4 %4, %8, %13 sc1\n\tglobal_load_dwordx4 %5, %8, %14 sc1\n\t"
                         "global_load_dwordx4 %6, %8, %15 sc1\n\tglobal_load_dwordx4 %7, %8, %16 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(off), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "s"(b[4]), "s"(b[5]), "s"(b[6]), "s"(b[7]) : "memory");
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok &= pr_ok(v[i], tag);
        PR_SPIN_TAIL(ok)
    }
}

// NR rows x 4 K segments (the W2 stage): unit (i, q) = base b[i] + off[q]
template <int NR>
__device__ __forceinline__ void pr_sweep_seg4(const u64* const (&b)[NR], const unsigned (&off)[4], unsigned tag, u32x4 (&v)[NR][4], bool& dead, uint32_t* ctl) {
    static_assert(NR == 1 || NR == 2 || NR == 4, "rows per column tile");
    for (unsigned spins = 0;; ++spins) {
        if constexpr (NR == 1) {
            asm volatile("global_load_dwordx4 %0, %4, %8 sc1\n\tglobal_load_dwordx4 %1, %5, %8 sc1\n\t"
                         "global_load_dwordx4 %2, %6, %8 sc1\n\tglobal_load_dwordx4 %3, %7, %8 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3])
                         : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(b[0]) : "memory");
        } else if constexpr (NR == 2) {
            asm volatile("global_load_dwordx4 %0, %8, %12 sc1\n\tglobal_load_dwordx4 %1, %9, %12 sc1\n\t"
                         "global_load_dwordx4 %2, %10, %12 sc1\n\tglobal_load_dwordx4 %3, %11, %12 sc1\n\t"
                         "global_load_dwordx4 %4, %8, %13 sc1\n\tglobal_load_dwordx4 %5, %9, %13 sc1\n\t"
                         "global_load_dwordx4 %6, %10, %13 sc1\n\tglobal_load_dwordx4 %7, %11, %13 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])
                         : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(b[0]), "s"(b[1]) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %16, %20 sc1\n\tglobal_load_dwordx4 %1, %17, %20 sc1\n\t"
                         "global_load_dwordx4 %2, %18, %20 sc1\n\tglobal_load_dwordx4 %3, %19, %20 sc1\n\t"
                         "global_load_dwordx4 %4, %16, %21 sc1\n\tglobal_load_dwordx4 %5, %17, %21 sc1\n\t"
                         "global_load_dwordx4 %6, %18, %21 sc1\n\tglobal_load_dwordx4 %7, %19, %21 sc1\n\t"
                         "global_load_dwordx4 %8, %16, %22 sc1\n\tglobal_load_dwordx4 %9, %17, %22 sc1\n\t"
                         "global_load_dwordx4 %10, %18, %22 sc1\n\tglobal_load_dwordx4 %11, %19, %22 sc1\n\t"
                         "global_load_dwordx4 %12, %16, %23 sc1\n\tglobal_load_dwordx4 %13, %17, %23 sc1\n\t"
                         "global_load_dwordx4 %14, %18, %23 sc1\n\tglobal_load_dwordx4 %15, %19, %23 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3]),
                           "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[2][2]), "=&v"(v[2][3]), "=&v"(v[3][0]), "=&v"(v[3][1]), "=&v"(v[3][2]), "=&v"(v[3][3])
                         : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]) : "memory");
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) ok &= pr_ok(v[i][q], tag);
        PR_SPIN_TAIL(ok)
    }
}

// the same 2 x 4 units issued WITHOUT waiting (speculative: the producers publish all their rows together, so when rows 0, 1 of an edge
// are complete rows 2, 3 almost always are), and the wait + tag check behind the arithmetic on the first rows.  Nothing may touch v
// between the two calls; the compiler's own loads must have been waited for before the issue (its next vmcnt wait would cover these)
__device__ __forceinline__ void pr_issue_seg4_2(const u64* const (&b)[2], const unsigned (&off)[4], u32x4 (&v)[2][4]) {
    asm volatile("global_load_dwordx4 %0, %8, %12 sc1\n\tglobal_load_dwordx4 %1, %9, %12 sc1\n\t"
                 "global_load_dwordx4 %2, %10, %12 sc1\n\tglobal_load_dwordx4 %3, %11, %12 sc1\n\t"
                 "global_load_dwordx4 %4, %8, %13 sc1\n\tglobal_load_dwordx4 %5, %9, %13 sc1\n\t"
                 "global_load_dwordx4 %6, %10, %13 sc1\n\tglobal_load_dwordx4 %7, %11, %13 sc1"
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(b[0]), "s"(b[1]) : "memory");
}
__device__ __forceinline__ bool pr_finish_seg4_2(u32x4 (&v)[2][4], unsigned tag) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]) : : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) ok &= pr_ok(v[i][q], tag);
    return ok;
}

// 8 attention slices: per slice one o-pair unit (b[k] + off_o) and the {m, l} unit (b[k] + off_ml)
__device__ __forceinline__ void pr_sweep_att8(const u64* const (&b)[8], unsigned off_o, unsigned off_ml, unsigned tag, u32x4 (&vo)[8], u32x4 (&vm)[8],
                                              bool& dead, uint32_t* ctl) {
    for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %16, %18 sc1\n\tglobal_load_dwordx4 %8, %17, %18 sc1\n\t"
                     "global_load_dwordx4 %1, %16, %19 sc1\n\tglobal_load_dwordx4 %9, %17, %19 sc1\n\t"
                     "global_load_dwordx4 %2, %16, %20 sc1\n\tglobal_load_dwordx4 %10, %17, %20 sc1\n\t"
                     "global_load_dwordx4 %3, %16, %21 sc1\n\tglobal_load_dwordx4 %11, %17, %21 sc1\n\t"
                     "global_load_dwordx4 %4, %16, %22 sc1\n\tglobal_load_dwordx4 %12, %17, %22 sc1\n\t"
                     "global_load_dwordx4 %5, %16, %23 sc1\n\tglobal_load_dwordx4 %13, %17, %23 sc1\n\t"
                     "global_load_dwordx4 %6, %16, %24 sc1\n\tglobal_load_dwordx4 %14, %17, %24 sc1\n\t"
                     "global_load_dwordx4 %7, %16, %25 sc1\n\tglobal_load_dwordx4 %15, %17, %25 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(vo[0]), "=&v"(vo[1]), "=&v"(vo[2]), "=&v"(vo[3]), "=&v"(vo[4]), "=&v"(vo[5]), "=&v"(vo[6]), "=&v"(vo[7]),
                       "=&v"(vm[0]), "=&v"(vm[1]), "=&v"(vm[2]), "=&v"(vm[3]), "=&v"(vm[4]), "=&v"(vm[5]), "=&v"(vm[6]), "=&v"(vm[7])
                     : "v"(off_o), "v"(off_ml), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "s"(b[4]), "s"(b[5]), "s"(b[6]), "s"(b[7]) : "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok &= pr_ok(vo[i], tag) && pr_ok(vm[i], tag);
        PR_SPIN_TAIL(ok)
    }
}
