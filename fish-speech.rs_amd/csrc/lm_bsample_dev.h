// Block-parallel top-k / top-p / WeightedIndex sampler (device code; include inside namespace fs).
//
// Same decision procedure, the same f32 operations in the same order, as the one-wave sampler of lm_kernels.hip (wave_topk_select +
// wave_pick) and as oracle::LogitsProcessor (candle LogitsProcessor / BatchedLogitsProcessor, sampling/mod.rs:51-132; rand 0.8.5
// WeightedIndex<f32> + UniformFloat<f32>): softmax(logits / temp) with an f64 denominator -> the top_k largest probabilities (ties: lower
// index first), kept in ascending index order -> if top_p < their sequential f32 sum: zero every probability from the rank (descending
// order) at which the running f32 sum has reached top_p -> cumulative f32 weights in index order -> first entry whose cumulative weight
// exceeds the uniform draw.  What is different is WHO does the work: all NT threads of the block instead of one wave, so the call costs a
// few microseconds instead of 12-13 -- it sits on the critical path of every codebook decision of a sampled request, 8 times per frame,
// on every workgroup of the persistent fast decoder (lm_persist.hip), and once per row in the batched samplers.
//   A  softmax: thread t owns the EPT consecutive candidates t*EPT ..; block max and the f64 sum meet in LDS (2 barriers)
//   B  k-th largest probability by radix select over the 30-bit patterns (8 + 8 + 8 + 6 bits): every thread counts its candidates into a
//      256-bin LDS histogram (ds_add), wave 0 scans the bins (4 per lane, DPP scan) while wave 1 clears the other histogram (2 barriers
//      per pass)
//   C  keep p > T and the first ties in index order: two block prefix scans (wave DPP scan + one LDS hop each), compaction into the
//      index-ordered arrays kp / ki; a tree sum of the kept probabilities decides whether a top-p cut is possible at all
//   D  in parallel: wave 0 runs the sequential ascending-index f32 sum of the kept probabilities (leaving the prefix sums, which ARE the
//      WeightedIndex cumulative weights when nothing is cut); if a cut is possible the other waves rank the entries by counting
//      (thread j: #{i: p_i > p_j or (p_i == p_j and i < j)}) and scatter the probabilities into descending order
//   E  top-p (only if top_p < sum): wave 0 walks the descending array until the running sum reaches top_p; entries ranked at or after the
//      cut are zeroed in parallel; the surviving weights (typically a few dozen) are compacted and summed by wave 0
//   F  draw: every thread tests its own entry against the uniform draw; LDS atomicMin / atomicMax pick the first / last non-zero entry
// Requirements: NT a multiple of 64, NT >= top_k + 64, top_k <= 256, n <= NT * EPT <= 2048.  All NT threads must call.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int BS_MAXK = 256;

// rand 0.8.5 StdRng == ChaCha12 (rand_chacha 0.3.1): word `n` of the keystream, 64-bit block counter, stream id 0.
__device__ inline uint32_t chacha12_word(const uint32_t* key, unsigned long long n) {
    const unsigned long long ctr = n >> 4;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = s[i];
#define FS_ROTL(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
#define FS_QR(a, b, c, d)                                   \
    w[a] += w[b]; w[d] = FS_ROTL(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = FS_ROTL(w[b] ^ w[c], 12); \
    w[a] += w[b]; w[d] = FS_ROTL(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = FS_ROTL(w[b] ^ w[c], 7);
    for (int r = 0; r < 6; ++r) {
        FS_QR(0, 4, 8, 12) FS_QR(1, 5, 9, 13) FS_QR(2, 6, 10, 14) FS_QR(3, 7, 11, 15)
        FS_QR(0, 5, 10, 15) FS_QR(1, 6, 11, 12) FS_QR(2, 7, 8, 13) FS_QR(3, 4, 9, 14)
    }
#undef FS_QR
#undef FS_ROTL
    uint32_t out = 0;
    const int idx = (int)(n & 15);
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i == idx) out = w[i] + s[i];
    return out;
}


struct alignas(16) BSampLds {  // LDS scratch of one call (7.9 KB)
    float kp[BS_MAXK + 32];    // kept probabilities, ascending token index
    float cumk[BS_MAXK + 32];  // their sequential prefix sums (later: of the compacted survivors)
    float sp[BS_MAXK + 32];    // kept probabilities, descending (later: compacted survivors)
    int ki[BS_MAXK];           // token index of kept entry j
    int rnk[BS_MAXK];          // descending-order rank of kept entry j (later: position of survivor j)
    int bkt[BS_MAXK];          // fast tail: entry index at each position of the bin-ordered scratch (the patterns sit in cumk)
    uint32_t hist[2][256];
    double wsum[16];
    float wmax[16];
    int wcnt[2][16];
    int misc[16];  // 0 bin, 1 above, 2 cut, 3 first, 4 last, 5 n_surv, 6 sum bits, 7 chosen bits, 8 fast-tail result, 9 candidates in the selected bin, 10 / 11 their max / min pattern, 12 zero-probability entries of the kept set (fast tail)
};

#ifdef BS_PROF
__device__ unsigned long long g_bs_ts[16];
#define BS_TS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_bs_ts[i] = clock64(); } while (0)
#else
#define BS_TS(i) do {} while (0)
#endif

template <int CTRL>
__device__ __forceinline__ float bs_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float bs_wave_sum(float v) {
    v += bs_dpp<0xB1>(v); v += bs_dpp<0x4E>(v); v += bs_dpp<0x141>(v); v += bs_dpp<0x140>(v);
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31))) +
           (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)));
}
__device__ __forceinline__ float bs_wave_max(float m) {
    m = fmaxf(m, bs_dpp<0xB1>(m)); m = fmaxf(m, bs_dpp<0x4E>(m)); m = fmaxf(m, bs_dpp<0x141>(m)); m = fmaxf(m, bs_dpp<0x140>(m));
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 47)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63))));
}
// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts: no LDS, no scalar hop)
__device__ __forceinline__ int bs_wave_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}
// the same for f32 (a fixed tree: used where the summation order is free)
__device__ __forceinline__ float bs_wave_scan_f(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return v;
}
// CH (32 | 16) consecutive floats of an LDS array with CH / 4 independent 16-byte reads; entries >= n read as +0.0 (x + 0.0f == x for the
// non-negative sums taken here)
template <int CH>
__device__ __forceinline__ void bs_fetch(const float* a, int j, int n, float (&v)[CH]) {
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(a + j + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    if (j + CH > n) {
#pragma unroll
        for (int e = 0; e < CH; ++e) if (j + e >= n) v[e] = 0.f;
    }
}

// lv: this thread's EPT logits (already penalised / masked), candidates tid * EPT + s; entries at or beyond n are ignored.
// word: the StdRng output word this draw would consume.  Returns the picked candidate index on every thread; *consumed = 1 iff the
// draw consumed `word` (rand's WeightedIndex needs a positive total).
// CH: chunk of the three sequential f32 chains (32: fewest loop trips; 16: 16 registers less at their peak -- the persistent fast decoder calls
// with its weights resident in all but ~20 registers, and every spilled value is a scratch round trip per decision)
template <int NT, int EPT, int CH = 32>
__device__ int bsample(const float (&lv)[EPT], int n, int kk, float inv_t, float top_p, uint32_t word, int* consumed, BSampLds& S,
                       bool batch = false, double top_p64 = 0.0) {  // batch: BatchedLogitsProcessor's f64 comparison (sampling/mod.rs:68)
    static_assert(NT % 64 == 0 && NT >= BS_MAXK + 64 && NT * EPT <= 2048, "block shape");
    constexpr int W = NT / 64;
    // (opaque: inside a persistent kernel's pass loop every per-lane LDS address of this function is loop-invariant; hoisted, the ~25 of them
    // outlive the loop next to 168 resident weight registers and were SPILLED -- a scratch load + vmcnt(0) in front of each LDS access, ~2 us
    // per call on the decision's critical path.  Re-deriving them from an opaque thread id costs a few VALU operations per use instead.)
    int tid_o = threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, wv = tid >> 6;
    const int base = tid * EPT;
    BS_TS(0);
    // ---- A: softmax
    for (int i = tid; i < 512; i += NT) (&S.hist[0][0])[i] = 0u;
    if (tid == 0) { S.misc[3] = 0x7FFFFFFF; S.misc[4] = -1; S.misc[2] = kk; }
    uint32_t u[EPT];
    uint32_t umax_bits = 0u;  // pattern of the largest probability (1 / denominator)
    {
        float v[EPT];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            v[s] = (base + s < n) ? lv[s] * inv_t : -INFINITY;
            mx = fmaxf(mx, v[s]);
        }
        mx = bs_wave_max(mx);
        if (lane == 0) S.wmax[wv] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < W; ++w) mx = fmaxf(mx, S.wmax[w]);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            v[s] = (base + s < n) ? expf(v[s] - mx) : 0.f;
            part += (double)v[s];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
        if (lane == 0) S.wsum[wv] = part;
        __syncthreads();
        double dsum = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) dsum += S.wsum[w];
        const float denom = (float)dsum;
#pragma unroll
        for (int s = 0; s < EPT; ++s) u[s] = __float_as_uint(v[s] / denom);  // 0 for slots past n
        // ---- shortcut: the largest probability alone exceeds top_p (peaked rows: most decisions of a trained model).  The chain below
        // would keep it in the top-k set, find top_p < sum (the f32 sum of the kept probabilities is >= any of them), rank it first (ties:
        // the lower index) and cut right behind it (its cumulative sum 0 + p_max has reached top_p): the draw runs over ONE positive weight,
        // consumes its word and returns that entry.  p_max = expf(0) / denom is known to every thread; only the lowest index carrying it
        // has to be found.  (Strict inequality: at top_p == sum the reference takes the no-cut branch.)
        const float pmax = 1.0f / denom;
        umax_bits = __float_as_uint(pmax);
        const bool cut1 = batch ? (top_p64 > 0.0 && (double)pmax > top_p64 && pmax >= top_p) : (top_p > 0.f && pmax > top_p);
        if (cut1) {
            int mine = 0x7FFFFFFF;
#pragma unroll
            for (int s = EPT - 1; s >= 0; --s) if (base + s < n && u[s] == __float_as_uint(pmax)) mine = base + s;
            mine = min(mine, __shfl_xor(mine, 32, 64)); mine = min(mine, __shfl_xor(mine, 16, 64)); mine = min(mine, __shfl_xor(mine, 8, 64));
            mine = min(mine, __shfl_xor(mine, 4, 64)); mine = min(mine, __shfl_xor(mine, 2, 64)); mine = min(mine, __shfl_xor(mine, 1, 64));
            if (lane == 0 && mine != 0x7FFFFFFF) atomicMin(&S.misc[3], mine);
            __syncthreads();
            const int pick = S.misc[3];
            __syncthreads();  // (the next call re-initialises S.misc)
            *consumed = 1;
            return pick;
        }
    }
    BS_TS(1);
    // ---- B: T = the kk-th largest pattern (#{u > T} < kk <= #{u >= T}); slots past n count as zeros, exactly as in the one-wave version
    uint32_t prefix = 0u;
    int krem = kk;
    bool done_early = false;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass == 0 ? 22 : (pass == 1 ? 14 : (pass == 2 ? 6 : 0)), bits = pass == 3 ? 6 : 8;
        uint32_t* h = S.hist[pass & 1];
#pragma unroll
        for (int s = 0; s < EPT; ++s)
            if (pass == 0 || (u[s] >> (shift + bits)) == prefix) atomicAdd(&h[(u[s] >> shift) & ((1u << bits) - 1u)], 1u);
        __syncthreads();
        if (wv == 0) {
            const uint4 hv = *reinterpret_cast<const uint4*>(h + lane * 4);
            const int h4[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
            const int mine = (h4[0] + h4[1]) + (h4[2] + h4[3]);
            const int incl = bs_wave_scan(mine);
            int above = __builtin_amdgcn_readlane(incl, 63) - incl;  // candidates in bins of higher lanes
#pragma unroll
            for (int j = 3; j >= 0; --j) {  // exactly one (lane, j) holds the krem-th candidate from the top
                if (above < krem && krem <= above + h4[j]) { S.misc[0] = lane * 4 + j; S.misc[1] = above; S.misc[9] = h4[j]; S.misc[10] = 0; S.misc[11] = 0x7FFFFFFF; }
                above += h4[j];
            }
        } else if (wv == 1) {
            *reinterpret_cast<uint4*>(S.hist[(pass + 1) & 1] + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        prefix = (prefix << bits) | (uint32_t)S.misc[0];
        krem -= S.misc[1];
        // round 6: one or two candidates left in the selected bin (the usual state after two passes: ~170 candidates of a flat row over 256
        // bins) -- their owners post the patterns themselves (LDS max / min) and the remaining passes, two barriers each, are skipped
        if (pass < 3 && S.misc[9] <= 2) {
#pragma unroll
            for (int s = 0; s < EPT; ++s)
                if ((u[s] >> shift) == prefix) { atomicMax(&S.misc[10], (int)u[s]); atomicMin(&S.misc[11], (int)u[s]); }
            __syncthreads();
            prefix = (uint32_t)(krem == 1 ? S.misc[10] : S.misc[11]);  // the krem-th largest of the bin's one or two patterns
            done_early = true;
            break;
        }
    }
    (void)done_early;  // (S.misc[9..11] are next written behind the barriers of the next call's softmax)
    const uint32_t T = prefix;
    BS_TS(2);
    // ---- C: keep p > T and the first kk - #{p > T} ties in index order (thread-major, then slot)
    if (tid == 0) S.misc[12] = 0;
    if (tid < 256) S.hist[0][tid] = 0u;  // (the fast tail's counting sort; nobody reads the selection's histograms any more, and C's barriers come before its atomics)
    const int nvalid = min(max(n - base, 0), EPT);  // only matters for ties at T == 0
    int pos;
    uint32_t keepbits = 0u;
    {
        int my_gt = 0, my_eq = 0;
#pragma unroll
        for (int s = 0; s < EPT; ++s) { my_gt += u[s] > T ? 1 : 0; my_eq += (u[s] == T && s < nvalid) ? 1 : 0; }
        const int packed = my_gt | (my_eq << 16);  // both counts stay below 2^12
        const int incl = bs_wave_scan(packed);
        if (lane == 63) S.wcnt[0][wv] = incl;
        __syncthreads();
        int lower = 0, total = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) { const int c = S.wcnt[0][w]; total += c; lower += w < wv ? c : 0; }
        const int r_ties = kk - (total & 0xFFFF);  // ties to keep (>= 1)
        int run_eq = ((incl + lower) >> 16) - my_eq;  // ties in lower threads
        int my_keep = 0;
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            const bool eq = u[s] == T && s < nvalid;
            const bool keep = u[s] > T || (eq && run_eq < r_ties);
            run_eq += eq ? 1 : 0;
            keepbits |= keep ? (1u << s) : 0u;
            my_keep += keep ? 1 : 0;
        }
        const int incl2 = bs_wave_scan(my_keep);
        if (lane == 63) S.wcnt[1][wv] = incl2;
        __syncthreads();
        pos = incl2 - my_keep;
#pragma unroll
        for (int w = 0; w < W; ++w) pos += w < wv ? S.wcnt[1][w] : 0;
    }
    float kept = 0.f;
#pragma unroll
    for (int s = 0; s < EPT; ++s)
        if (keepbits & (1u << s)) {
            S.kp[pos] = __uint_as_float(u[s]);
            S.ki[pos] = base + s;
            kept += __uint_as_float(u[s]);
            ++pos;
        }
    if (tid < 8 && kk + tid < BS_MAXK + 32) S.kp[kk + tid] = 0.f;  // (the rank loop reads whole groups of 8)
    kept = bs_wave_sum(kept);
    if (lane == 0) S.wmax[wv] = kept;  // (wmax was last read before the second softmax barrier)
    __syncthreads();
    // Is a top-p cut POSSIBLE?  The exact test compares top_p with the SEQUENTIAL f32 sum (below); any other summation order differs
    // from it by at most kk ulps of the total (< 2e-5 relative), so a tree sum with a 1e-4 margin decides the common "no cut" case
    // without waiting for the chain -- and then nobody needs the descending order at all.
    float approx = 0.f;
#pragma unroll
    for (int w = 0; w < W; ++w) approx += S.wmax[w];
    const bool maybe_topp = top_p > 0.f && top_p < approx * 1.0001f;
    BS_TS(3);
    // ---- fast tail (round 6): the three sequential f32 chains below (ascending-index sum, descending top-p walk, cumulative weights of the
    // survivors) are the reference's own sums and their rounding is part of the result -- but only through two COMPARISONS: "has the running
    // sum reached top_p" and "does the cumulative weight exceed the draw".  Any summation order of n <= 256 non-negative f32 terms lies within
    // gamma_255 = 255 * 2^-24 / (1 - 255 * 2^-24) < 1.53e-5 (relative) of the exact sum, so two orders differ by < 3.05e-5 of it.  Wave 0
    // therefore takes the sums as PARALLEL prefix scans (4 entries per lane + a DPP wave scan, ~100 instructions instead of a 150-256-step
    // dependent chain) and accepts a comparison only where every prefix sum stays at least 6.2e-5 x total (top-p walk) / 1.3e-4 x total
    // (draw: the threshold itself, u * scale(total), inherits the total's error) away from the threshold; otherwise -- a few per cent of
    // the calls on flat rows -- the call falls through to the exact chains below, unchanged.  Same picks, token for token
    // (tests/test_sampler_gpu.py, tests/test_persist_sampled_gpu.py); flat rows 11.5 -> ~5 us per call (profiles/r06_ubench_bsample.txt).
    {
        const double tp = batch ? top_p64 : (double)top_p;
        const bool cut_sure = tp > 0.0 && tp < (double)approx * 0.9999, nocut_sure = !(tp > 0.0) || tp >= (double)approx * 1.0001;
        if (cut_sure || nocut_sure) {  // (uniform over the block)
            if (cut_sure) {
                // Descending-order ranks by a counting sort on the kept patterns (round 6; the all-pairs count it replaces was 65 536 compares,
                // ~4.5 k clocks).  The kept patterns lie in [T, umax] (umax = the pattern of 1 / denom); bin = the top 8 bits of (umax - u) at
                // that range's scale, so a flat row spreads its 256 entries over ~256 bins and a bin holds 1-3 of them.  (1) LDS histogram +
                // slot by ds_add_rtn, (2) wave 0 scans the bins, (3) entries scatter into bin order, (4) each entry ranks itself among its
                // bin's members: rank = #{u > u_j} + #{u == u_j, index < j} -- exactly the order the exact tail uses.  Worst case (every
                // entry in one bin: equal patterns) degenerates to the old all-pairs count.
                const bool ent = tid < kk;
                const uint32_t uj = ent ? __float_as_uint(S.kp[tid]) : T;
                const uint32_t range = umax_bits - T;
                const int sh = max(0, 24 - (int)__builtin_clz(range | 1u));  // (range >> sh) < 256
                const int bin = 255 - (int)((uj - T) >> sh);
                uint32_t* bk_u = reinterpret_cast<uint32_t*>(S.cumk);       // (cumk is not needed before the exact tail, which rewrites it)
                // (zero probabilities -- underflowed candidates a sharp row keeps by the hundred -- would all land in one bin and each walk it:
                // 20 k clocks measured.  They rank behind every positive entry and move no sum, so their order among themselves is free:
                // a counter hands them the last ranks.)
                const bool zent = ent && uj == 0u;
                int slot = 0;
                if (zent) slot = atomicAdd(&S.misc[12], 1);
                else if (ent) slot = (int)atomicAdd(&S.hist[0][bin], 1u);   // (hist[0] and misc[12] were zeroed at the top of phase C)
                __syncthreads();
                if (wv == 0) {
                    const uint4 hv = *reinterpret_cast<const uint4*>(S.hist[0] + lane * 4);
                    const int mine = (int)((hv.x + hv.y) + (hv.z + hv.w));
                    const int excl = bs_wave_scan(mine) - mine;
                    *reinterpret_cast<uint4*>(S.hist[1] + lane * 4) = make_uint4((uint32_t)excl, (uint32_t)excl + hv.x, (uint32_t)excl + hv.x + hv.y, (uint32_t)excl + hv.x + hv.y + hv.z);
                }
                __syncthreads();
                int start = 0, cnt = 0;
                if (ent && !zent) {
                    start = (int)S.hist[1][bin]; cnt = (int)S.hist[0][bin];
                    bk_u[start + slot] = uj;
                    S.bkt[start + slot] = tid;
                }
                __syncthreads();
                if (ent) {
                    int r = zent ? kk - S.misc[12] + slot : start;
                    for (int e = 0; e < cnt; ++e) {  // (cnt == 0 for a zero entry)
                        const uint32_t ue = bk_u[start + e];
                        const int je = S.bkt[start + e];
                        r += (ue > uj || (ue == uj && je < tid)) ? 1 : 0;
                    }
                    S.rnk[tid] = r;
                    S.sp[r] = __uint_as_float(uj);
                }
                __syncthreads();
            }
            if (wv == 0) {
                const int i0 = 4 * lane;
                bool amb = false;
                int cut = kk;
                if (cut_sure) {  // first rank whose inclusive running sum (descending order) has reached top_p
                    const float4 s4 = *reinterpret_cast<const float4*>(S.sp + i0);
                    const float c0 = i0 < kk ? s4.x : 0.f, c1 = c0 + (i0 + 1 < kk ? s4.y : 0.f), c2 = c1 + (i0 + 2 < kk ? s4.z : 0.f), c3 = c2 + (i0 + 3 < kk ? s4.w : 0.f);
                    const float off = bs_wave_scan_f(c3) - c3, eps = 6.2e-5f * approx;
                    const float d[4] = {off + c0, off + c1, off + c2, off + c3};
                    int below = 0;
                    bool near = false;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i0 + e < kk) { below += d[e] < top_p ? 1 : 0; near |= fabsf(d[e] - top_p) < eps; }
                    below = bs_wave_scan(below);
                    const int q = __builtin_amdgcn_readlane(below, 63);  // == the first rank whose sum is >= top_p (kk: none)
                    amb = __ballot(near) != 0ull;
                    cut = q < kk ? q + 1 : kk;
                }
                // cumulative weights of the survivors in index order (a cut entry weighs zero and moves no sum) and the draw over them
                const float4 k4 = *reinterpret_cast<const float4*>(S.kp + i0);
                float w[4] = {k4.x, k4.y, k4.z, k4.w};
                if (cut_sure) {
                    const int4 r4 = *reinterpret_cast<const int4*>(S.rnk + i0);
                    if (r4.x >= cut) w[0] = 0.f;
                    if (r4.y >= cut) w[1] = 0.f;
                    if (r4.z >= cut) w[2] = 0.f;
                    if (r4.w >= cut) w[3] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) if (i0 + e >= kk) w[e] = 0.f;
                const float c0 = w[0], c1 = c0 + w[1], c2 = c1 + w[2], c3 = c2 + w[3];
                const float incl = bs_wave_scan_f(c3), off = incl - c3;
                const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
                const float d[4] = {off + c0, off + c1, off + c2, off + c3};
                int tok = -1;
                if (!(total > 0.f)) amb = true;  // (all-zero weights: the exact tail knows the reference's answer)
                else {
                    // WeightedIndex::sample -- UniformFloat<f32>::sample_single over [0, total), as in F below
                    const float max_rand = __uint_as_float((0xFFFFFFFFu >> 9) | (127u << 23)) - 1.0f;
                    float scale = total;
                    while (scale * max_rand + 0.f >= total) scale = __uint_as_float(__float_as_uint(scale) - 1u);
                    const float chosen = (__uint_as_float((word >> 9) | (127u << 23)) - 1.0f) * scale + 0.f, eps2 = 1.3e-4f * total;
                    int first = 4;
                    bool near = false;
#pragma unroll
                    for (int e = 3; e >= 0; --e) {
                        if (w[e] != 0.f && d[e] > chosen) first = e;
                        near |= fabsf(d[e] - chosen) < eps2;
                    }
                    amb |= __ballot(near && i0 < kk) != 0ull;
                    const unsigned long long m_hit = __ballot(first < 4);
                    if (m_hit == 0ull) amb = true;
                    else {
                        const int src = __builtin_ctzll(m_hit);
                        const int jsel = 4 * src + __builtin_amdgcn_readlane(first, src);
                        tok = S.ki[jsel];
                    }
                }
                if (lane == 0) S.misc[8] = amb ? -1 : tok;
            }
            __syncthreads();
            const int fast = S.misc[8];
            __syncthreads();  // (the scratch may be re-used by the caller; the exact tail below re-uses it too)
            if (fast >= 0) {
                BS_TS(4); BS_TS(5); BS_TS(6);
                *consumed = 1;
                return fast;
            }
        }
    }
    // ---- D: ascending-index sum (wave 0) || ranks by counting (threads 64 .. 64 + kk)
    if (wv == 0) {
        float cum = 0.f;
        for (int j = 0; j < kk; j += CH) {
            float v[CH];
            bs_fetch<CH>(S.kp, j, kk, v);
#pragma unroll
            for (int e = 0; e < CH; ++e) { cum += v[e]; v[e] = cum; }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) *reinterpret_cast<float4*>(S.cumk + j + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
        }
        if (lane == 0) S.misc[6] = __float_as_int(cum);
    } else if (maybe_topp && tid - 64 < ((kk + 63) & ~63)) {
        // rank of entry j in descending order = #{i : p_i > p_j or (p_i == p_j and i < j)} (ties: the lower position first).  Patterns are
        // < 2^30, so they compare as signed ints and "p_i >= p_j" is "p_i > p_j - 1".  A wave holds 64 consecutive j: entries below its
        // first j are "i < j" for every lane, entries above its last are "i > j", only its own 64 need the per-lane choice.  Entries
        // kk .. kk8 are zero (never counted: 0 > p_j is false).
        const int j = tid - 64, jb = __builtin_amdgcn_readfirstlane(j & ~63), kk8 = (kk + 7) & ~7;
        const int pj = __float_as_int(S.kp[min(j, kk - 1)]), pjm1 = pj - 1;
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        auto count8 = [&](int i, auto thr) {
            const float4 a = *reinterpret_cast<const float4*>(S.kp + i), b4 = *reinterpret_cast<const float4*>(S.kp + i + 4);
            r0 += __float_as_int(a.x) > thr(i) ? 1 : 0; r1 += __float_as_int(a.y) > thr(i + 1) ? 1 : 0;
            r2 += __float_as_int(a.z) > thr(i + 2) ? 1 : 0; r3 += __float_as_int(a.w) > thr(i + 3) ? 1 : 0;
            r0 += __float_as_int(b4.x) > thr(i + 4) ? 1 : 0; r1 += __float_as_int(b4.y) > thr(i + 5) ? 1 : 0;
            r2 += __float_as_int(b4.z) > thr(i + 6) ? 1 : 0; r3 += __float_as_int(b4.w) > thr(i + 7) ? 1 : 0;
        };
        const int lo_end = min(jb, kk8), mid_end = min(jb + 64, kk8);
        for (int i = 0; i < lo_end; i += 8) count8(i, [&](int) { return pjm1; });
        for (int i = lo_end; i < mid_end; i += 8) count8(i, [&](int ii) { return ii < j ? pjm1 : pj; });
        for (int i = mid_end; i < kk8; i += 8) count8(i, [&](int) { return pj; });
        const int r = (r0 + r1) + (r2 + r3);
        if (j < kk) {
            S.rnk[j] = r;
            S.sp[r] = __int_as_float(pj);
        }
    }
    __syncthreads();
    BS_TS(4);
    const float sum_p = __int_as_float(S.misc[6]);
    const bool do_topp = maybe_topp && (batch ? !(top_p64 <= 0.0 || top_p64 >= (double)sum_p) : !(top_p <= 0.f || top_p >= sum_p));  // (maybe_topp is implied)
    const float* cumw = S.cumk;  // cumulative weights of the entries the draw runs over
    const int* wpos = nullptr;   // their positions in (kp, ki); null: identity
    int cnt = kk;
    float total = sum_p;
    if (do_topp) {
        // ---- E: the first rank whose EXCLUSIVE running sum (descending order) has reached top_p == 1 + first q with inclusive sum >= top_p
        if (wv == 0) {
            float cum = 0.f;
            int first = -1;
            for (int j = 0; j < kk && first < 0; j += CH) {
                float v[CH];
                bs_fetch<CH>(S.sp, j, kk, v);
#pragma unroll
                for (int e = 0; e < CH; ++e) { cum += v[e]; v[e] = cum; }
#pragma unroll
                for (int e = CH - 1; e >= 0; --e) if (j + e < kk && v[e] >= top_p) first = j + e;
            }
            if (lane == 0) S.misc[2] = first < 0 ? kk : first + 1;
        }
        __syncthreads();
        const int cut = S.misc[2];
        if (tid < kk && S.rnk[tid] >= cut) S.kp[tid] = 0.f;
        __syncthreads();
        // compact the non-zero weights in index order (zero weights do not move a cumulative f32 sum and are never picked)
        if (wv == 0) {
            const float4 wv4 = *reinterpret_cast<const float4*>(S.kp + lane * 4);
            const float ws[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
            int mine = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) mine += (lane * 4 + s < kk && ws[s] != 0.f) ? 1 : 0;
            const int incl = bs_wave_scan(mine);
            const int m = __builtin_amdgcn_readlane(incl, 63);
            int p2 = incl - mine;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (lane * 4 + s < kk && ws[s] != 0.f) { S.sp[p2] = ws[s]; S.rnk[p2] = lane * 4 + s; ++p2; }
            float cum = 0.f;  // (one wave: its LDS operations stay in order)
            for (int j = 0; j < m; j += CH) {
                float v[CH];
                bs_fetch<CH>(S.sp, j, m, v);
#pragma unroll
                for (int e = 0; e < CH; ++e) { cum += v[e]; v[e] = cum; }
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < CH / 4; ++q) *reinterpret_cast<float4*>(S.cumk + j + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
            }
            if (lane == 0) { S.misc[5] = m; S.misc[6] = __float_as_int(cum); }
        }
        __syncthreads();
        cnt = S.misc[5];
        total = __int_as_float(S.misc[6]);
        wpos = S.rnk;
    }
    BS_TS(5);
    // ---- F: WeightedIndex::sample -- UniformFloat<f32>::sample_single over [0, total), first entry whose cumulative weight exceeds it
    if (!(total > 0.f) || cnt == 0) {
        *consumed = 0;
        const int res = S.ki[0];
        __syncthreads();
        return res;
    }
    *consumed = 1;
    const float max_rand = __uint_as_float((0xFFFFFFFFu >> 9) | (127u << 23)) - 1.0f;
    float scale = total;
    while (scale * max_rand + 0.f >= total) scale = __uint_as_float(__float_as_uint(scale) - 1u);
    const float chosen = (__uint_as_float((word >> 9) | (127u << 23)) - 1.0f) * scale + 0.f;
    if (wv * 64 < cnt) {  // one LDS atomic per wave, not per entry (same-address LDS atomics serialise)
        const bool nz = tid < cnt && (do_topp || S.kp[tid] != 0.f);  // (compacted entries are non-zero by construction)
        const unsigned long long m_nz = __ballot(nz), m_hit = __ballot(nz && cumw[tid] > chosen);
        if (lane == 0) {
            if (m_nz) atomicMax(&S.misc[4], wv * 64 + 63 - __builtin_clzll(m_nz));
            if (m_hit) atomicMin(&S.misc[3], wv * 64 + __builtin_ctzll(m_hit));
        }
    }
    __syncthreads();
    const int f = S.misc[3], l = S.misc[4];
    const int jsel = f != 0x7FFFFFFF ? f : l;
    const int res = jsel < 0 ? S.ki[0] : S.ki[wpos ? wpos[jsel] : jsel];
    __syncthreads();  // the scratch may be re-used by the caller
    BS_TS(6);
    return res;
}
