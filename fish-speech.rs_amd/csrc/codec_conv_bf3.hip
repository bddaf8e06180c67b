// Causal conv1d of the Firefly vocoder on the bf16 matrix cores with split operands ("bf16x3"), gfx950.
//
// Same contract as k_conv1d / k_conv1d_mfma (codec_kernels.hip; hifi_gan.rs:74-85,208-216, convnext.rs:110-122): stride 1, left zero
// pad (K-1)*dil, SiLU pre-activation, bias / GELU / gamma+residual / residual / tanh epilogues, polyphase transposed convs.
//
// Arithmetic.  The convolution is the GEMM Y[o][t] = sum_{(i,k)} W[o][i][k] * X[i][t + k*dil - halo].  Every f32 operand v is split
// into two bf16 parts v = hi + lo + e, hi = bf16_rne(v), lo = bf16_rne(v - hi), |e| <= 2^-17 |v|, and a product is evaluated as
// W_hi*X_hi + W_hi*X_lo + W_lo*X_hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation: three matrix instructions of 16 reduction
// items each instead of eight v_mfma_f32_32x32x2_f32 of two items (the f32 matrix rate is 1/16 of the bf16 rate).  The dropped
// W_lo*X_lo term and the two split residuals are each <= 2^-16 relative per product, i.e. ~1e-5: the PCM stays within the 1e-4 RMS
// acceptance bound of the f32 oracle (measured in tests/test_codec_gpu.py), but it is NOT the exact-f32 product chain of
// k_conv1d_mfma, which is kept as the codec's "f32" precision mode and for the encoder.
// The summation order of one output element is fixed -- 16-channel blocks ascending, taps ascending, (hi*hi, hi*lo, lo*hi) -- and
// does not depend on the tile shape, so a code prefix still decodes to the bit-identical PCM prefix whatever T is.
//
// Layout.  Weights are packed once at load time into MFMA A-operand order:
//   wp[ib][k][plane][o (Cout padded to 64)][8 bf16],  plane = part * 2 + half, part 0 = hi / 1 = lo,
//   element e of (ib, half) = input channel ib*16 + half*8 + e  (zero beyond Cin / Cout)
// so that the 32 lanes of a half-wave read 32 consecutive 16-byte slots (conflict-free ds_read_b128) and a block's weight tile
// for one (k, plane) is one contiguous run in global memory.  The x window is staged per 16-channel block as
//   xs[plane][window position][8 bf16]  (same planes)
// by the thread that owns the window position: 16 coalesced f32 loads (one per channel), SiLU, split, four 16-byte LDS stores.
// A block is 4 waves = OT output channels x TT samples; OT = 64: wave = 32 channels x TT/2 samples, OT = 32: 32 channels x TT/4.
// The global loads of stage n+1 are in flight (registers) while stage n runs on the matrix cores.  Layers whose whole weight
// set fits the stage (Cin <= 16 * NIBS: the thin 16/32-channel late stages) keep it resident in LDS and loop over time tiles.
#include <hip/hip_runtime.h>
#include <set>

#include "codec_kernels.h"
#include "fs_common.h"

namespace fs {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float c3_silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float c3_gelu(float x) { return 0.5f * x * (1.f + tanhf(0.7978845608028654f * x * (1.f + 0.044715f * x * x))); }
__device__ __forceinline__ uint32_t c3_bf16(float f) {  // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void c3_split(float v, uint32_t& hi, uint32_t& lo) {
    hi = c3_bf16(v);
    lo = c3_bf16(v - __uint_as_float(hi << 16));
}

// dst[((ib*K + k)*4 + p) * Cp + o][e] from the re-laid f32 weight src[(i*K + k)*Cout + o]
__global__ void k_pack_bf3(const float* __restrict__ src, uint16_t* __restrict__ dst, int Cin, int K, int Cout, int Cp) {
    const int nib = (Cin + 15) / 16;
    const size_t n = (size_t)nib * K * 4 * Cp * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        size_t r = idx >> 3;
        const int o = (int)(r % Cp); r /= Cp;
        const int p = (int)(r & 3); r >>= 2;
        const int k = (int)(r % K);
        const int ib = (int)(r / K);
        const int i = ib * 16 + (p & 1) * 8 + e;
        uint32_t hi = 0, lo = 0;
        if (i < Cin && o < Cout) c3_split(src[((size_t)i * K + k) * Cout + o], hi, lo);
        dst[idx] = (uint16_t)((p >> 1) ? lo : hi);
    }
}

// OT: output channels per block (64 | 32); TT: samples per block; NIBS: 16-channel blocks per stage; KMAX: largest tap count the
// register prefetch is sized for; NPX: window positions per thread (XS = TT + halo <= 256 * NPX)
template <int OT, int TT, int NIBS, int KMAX, int NPX>
__global__ __launch_bounds__(256) void k_conv1d_bf3(const float* __restrict__ x, int Cin, int T, const uint16_t* __restrict__ wp, int Cp,
                                                    const float* __restrict__ bias, int Cout, int K, int dil, int pre_silu, int epi,
                                                    const float* __restrict__ res, const float* __restrict__ gamma, float* __restrict__ y,
                                                    int ps, int ntiles) {
    constexpr int WT_ = OT == 64 ? TT / 2 : TT / 4;  // samples per wave
    constexpr int NT = WT_ / 32;                     // 32-sample MFMA tiles per wave
    static_assert(NT >= 1 && (OT == 64 || OT == 32), "block shape");
    constexpr int NWC = (NIBS * KMAX * 4 * OT + 255) / 256;  // 16-byte weight chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int halo = (K - 1) * dil, XS = TT + halo;
    const int nib = (Cin + 15) >> 4, nst = (nib + NIBS - 1) / NIBS;
    const bool resident = nst == 1;
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);      // [NIBS][4][XS]
    u32x4* ws = xs + NIBS * 4 * XS;                      // [NIBS][K][4][OT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int ob = OT == 64 ? (wave & 1) * 32 : 0, tb = OT == 64 ? (wave >> 1) * WT_ : wave * WT_;
    const int o0 = blockIdx.y * OT;
    const size_t boff_in = (size_t)blockIdx.z * Cin * T, boff_out = (size_t)blockIdx.z * Cout * T;
    const int wchunks = K * 4 * OT;  // per 16-channel block

    float xr[NPX][NIBS][16];
    u32x4 wr[NWC];
    auto load_x = [&](int st, int t0) {
#pragma unroll
        for (int b = 0; b < NIBS; ++b) {
            const int i0 = (st * NIBS + b) * 16;
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = tid + 256 * q, t = t0 + tl - halo;
                const bool tv = tl < XS && t >= 0 && t < T;
#pragma unroll
                for (int i = 0; i < 16; ++i) xr[q][b][i] = (tv && i0 + i < Cin) ? x[boff_in + (size_t)(i0 + i) * T + t] : 0.f;
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int b = 0; b < NIBS; ++b)
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int tl = tid + 256 * q;
                if (tl < XS) {
                    uint32_t hi[16], lo[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) c3_split(pre_silu ? c3_silu(xr[q][b][i]) : xr[q][b][i], hi[i], lo[i]);
                    u32x4* d = xs + (b * 4) * XS + tl;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        u32x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vh[e] = hi[hf * 8 + 2 * e] | (hi[hf * 8 + 2 * e + 1] << 16);
                            vl[e] = lo[hf * 8 + 2 * e] | (lo[hf * 8 + 2 * e + 1] << 16);
                        }
                        d[hf * XS] = vh;
                        d[(2 + hf) * XS] = vl;
                    }
                }
            }
    };
    // weight chunks of stage st: chunk e -> (block b, row kp = k*4 + plane, column o): contiguous OT-chunk runs in global memory
    auto load_w = [&](int st) {
#pragma unroll
        for (int j = 0; j < NWC; ++j) {
            const int e = j * 256 + tid, b = NIBS == 1 ? 0 : e / wchunks, r = e - b * wchunks, kp = r / OT, o = r % OT;
            const int ib = st * NIBS + b;
            if (e < NIBS * wchunks && ib < nib) wr[j] = *reinterpret_cast<const u32x4*>(wp + (((size_t)ib * K * 4 + kp) * Cp + o0 + o) * 8);
            else wr[j] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < NWC; ++j) {
            const int e = j * 256 + tid;
            if (e < NIBS * wchunks) ws[e] = wr[j];
        }
    };

    if (resident) {
        load_w(0);
        store_w();
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int t0 = tile * TT;
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        load_x(0, t0);
        if (!resident) load_w(0);
        for (int st = 0; st < nst; ++st) {
            __syncthreads();  // previous stage's (or tile's) LDS reads are done
            store_x();
            if (!resident) store_w();
            __syncthreads();
            if (st + 1 < nst) {
                load_x(st + 1, t0);
                load_w(st + 1);
            }
            const int nb = min(NIBS, nib - st * NIBS);
            for (int b = 0; b < nb; ++b) {
                const u32x4* wl = ws + (b * K * 4 + h) * OT + ob + c;
                const u32x4* xl = xs + (b * 4 + h) * XS + tb + c;
                for (int k = 0; k < K; ++k) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wl[k * 4 * OT]), al = __builtin_bit_cast(bf16x8, wl[(k * 4 + 2) * OT]);
                    const u32x4* xk = xl + k * dil;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, xk[32 * j]), bl = __builtin_bit_cast(bf16x8, xk[2 * XS + 32 * j]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                    }
                }
            }
        }
        // D[row][col]: register r of lane (h, c) holds row (r/4)*8 + h*4 + r%4, column c
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + ob + (r >> 2) * 8 + h * 4 + (r & 3);
            if (o >= Cout) continue;
            const float bv = bias[o / ps];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + tb + j * 32 + c;
                if (t >= T) continue;
                float v = acc[j][r] + bv;
                const size_t oi = boff_out + (size_t)(o / ps) * T * ps + (size_t)t * ps + o % ps;  // (polyphase rows: see k_conv1d)
                if (epi == CODEC_EPI_GELU) v = c3_gelu(v);
                else if (epi == CODEC_EPI_GAMMA_RES) v = res[oi] + gamma[o] * v;
                else if (epi == CODEC_EPI_RES) v = res[oi] + v;
                else if (epi == CODEC_EPI_TANH) v = tanhf(v);
                y[oi] = v;
            }
        }
    }
}

}  // namespace

size_t codec_pack_bf3_elems(int Cin, int K, int Cout) {
    const int Cp = (Cout + 63) / 64 * 64;
    return (size_t)((Cin + 15) / 16) * K * 4 * Cp * 8;
}

void codec_pack_bf3(const float* relaid, uint16_t* dst, int Cin, int K, int Cout, hipStream_t st) {
    const int Cp = (Cout + 63) / 64 * 64;
    const size_t n = codec_pack_bf3_elems(Cin, K, Cout);
    hipLaunchKernelGGL(k_pack_bf3, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, relaid, dst, Cin, K, Cout, Cp);
    FS_HIP(hipGetLastError());
}

bool codec_conv1d_bf3_ok(int Cin, int Cout, int K, int dil) { return Cin >= 16 && Cout >= 16 && K <= 13 && (K - 1) * dil <= 256; }

// x (B, Cin, T) -> y; `Cout` = GEMM rows (channels * ps for a polyphase transposed conv)
void codec_conv1d_bf3(const float* x, int B, int Cin, int T, const uint16_t* wp, const float* bias, int Cout, int K, int dil, bool pre_silu,
                      int epi, const float* res, const float* gamma, float* y, int ps, hipStream_t st) {
    FS_REQUIRE(codec_conv1d_bf3_ok(Cin, Cout, K, dil), "conv shape outside the bf16x3 kernel's range");
    const int Cp = (Cout + 63) / 64 * 64, halo = (K - 1) * dil, nib = (Cin + 15) / 16;
    auto go = [&](auto kern, int OT, int TT, int NIBS, bool loop_tiles) {
        const int XS = TT + halo;
        const size_t smem = 16 * ((size_t)NIBS * 4 * XS + (size_t)NIBS * K * 4 * OT);
        FS_REQUIRE(smem <= 160 * 1024, "conv tile does not fit LDS");
        if (smem > 64 * 1024) {  // above the default dynamic-LDS limit: raise it once per kernel
            static thread_local std::set<const void*> raised;
            if (raised.insert((const void*)kern).second)
                FS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        const int ntiles = (T + TT - 1) / TT, ytiles = (Cout + OT - 1) / OT;
        int gx = ntiles;
        if (loop_tiles) gx = std::max(1, std::min(ntiles, 1024 / std::max(1, ytiles * B)));  // weights stay in LDS over a loop of time tiles
        hipLaunchKernelGGL(kern, dim3(gx, ytiles, B), dim3(256), smem, st, x, Cin, T, wp, Cp, bias, Cout, K, dil, pre_silu ? 1 : 0, epi, res,
                           gamma, y, ps, ntiles);
    };
    if (K == 1 && Cin >= 128) {
        // pointwise convs of the ConvNeXt blocks (frame-rate T, 512..2048 channels): 8 channel blocks per barrier pair
        go(k_conv1d_bf3<32, 128, 8, 1, 1>, 32, 128, 8, false);
    } else if (nib <= 2) {
        // thin late stages (16 / 32 channels): the whole weight set is resident, the block walks over time tiles
        if (nib == 1) go(k_conv1d_bf3<32, 256, 1, 13, 2>, 32, 256, 1, true);
        else go(k_conv1d_bf3<32, 128, 2, 13, 2>, 32, 128, 2, true);
    } else {
        const bool tall = Cout >= 64 && (long long)((T + 127) / 128) * ((Cout + 63) / 64) * B >= 256;
        const int OT = tall ? 64 : 32;
        const bool wide = (long long)((T + 255) / 256) * ((Cout + OT - 1) / OT) * B >= 512;
        if (OT == 64 && wide) go(k_conv1d_bf3<64, 256, 1, 13, 2>, 64, 256, 1, false);
        else if (OT == 64) go(k_conv1d_bf3<64, 128, 1, 13, 2>, 64, 128, 1, false);
        else if (wide) go(k_conv1d_bf3<32, 256, 1, 13, 2>, 32, 256, 1, false);
        else go(k_conv1d_bf3<32, 128, 1, 13, 2>, 32, 128, 1, false);
    }
    FS_HIP(hipGetLastError());
}

}  // namespace fs
