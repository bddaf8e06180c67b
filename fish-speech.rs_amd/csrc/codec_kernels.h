// Launch interface of the Firefly vocoder kernels (codec_kernels.hip).  gfx950 only, f32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace fs {

struct ConvW {
    const float* wt;  // re-laid weight [Cin/groups][K][Cout]
    const float* b;   // [Cout]
    int cout, k;
    const uint16_t* wp = nullptr;  // bf16 hi/lo split or single f16, MFMA A-operand order (codec_pack_bf3); nullptr = exact-f32 kernels only
    bool f16 = false;              // format of `wp` (and of the activation planes exchanged with it): "f16" precision mode
};

constexpr int CODEC_PLANE_PAD = 64;  // zero slots in front of every activation-plane row (>= the largest halo of a plane consumer)
enum { CODEC_EPI_NONE = 0, CODEC_EPI_GELU = 1, CODEC_EPI_GAMMA_RES = 2, CODEC_EPI_RES = 3, CODEC_EPI_TANH = 4 };

void codec_fsq_project(const uint32_t* codes, int B, int G, int T, const float* pw, const float* pb, int dg, float* z, hipStream_t st);
// causal conv1d, stride 1: y = epi(bias + W * pre(x)); x (B, Cin, T) -> y (B, Cout, T)
void codec_conv1d(const float* x, int B, int Cin, int T, const ConvW& w, int dil, bool pre_silu, int epi, const float* res,
                  const float* gamma, float* y, hipStream_t st, const float* ctx = nullptr);
// transposed conv1d with right trim: x (B, Cin, Tin) -> y (B, Cout, Tin * stride); w = polyphase layout (codec_relayout_tconv)
void codec_tconv1d(const float* x, int B, int Cin, int Tin, const ConvW& w, int stride, bool pre_silu, float* y, hipStream_t st);
// f32 left contexts of the streaming decode: ctx [C][CODEC_CTX_F32] = the last CODEC_CTX_F32 samples of the previous chunk (null = zeros)
constexpr int CODEC_CTX_F32 = 16;
void codec_dwconv_ln(const float* x, int B, int C, int T, const float* dw, const float* db, const float* lnw, const float* lnb, float* y,
                     hipStream_t st, const float* ctx = nullptr);
void codec_save_tail_f32(const float* x, int C, int T, float* ctx_out, hipStream_t st);
void codec_mean3(const float* a, const float* b, const float* c, float* y, size_t n, hipStream_t st);
void codec_relayout(const float* src, float* dst, int Cout, int CinG, int K, bool transposed, hipStream_t st);
// ConvTranspose1d [Cin][Cout][K] -> polyphase causal-conv layout [Cin][K/stride][Cout*stride] (see codec_tconv1d)
void codec_relayout_tconv(const float* src, float* dst, int Cout, int Cin, int K, int stride, hipStream_t st);
// ---- bf16x3 convolution (codec_conv_bf3.hip): W and X split into bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 per 16 reduction items
// (f16 = true: the same kernels with ONE f16 value per operand and one v_mfma_f32_32x32x16_f16 per 16 items -- the "f16" precision mode)
size_t codec_pack_bf3_elems(int Cin, int K, int Cout, bool f16);
void codec_pack_bf3(const float* relaid /*[Cin][K][Cout]*/, uint16_t* dst, int Cin, int K, int Cout, bool f16, hipStream_t st);
bool codec_conv1d_bf3_ok(int Cin, int Cout, int K, int dil);
// input: f32 `x` (B, Cin, T) or activation planes `xp` ([B][2][Cin/8][T][8] bf16 hi / lo, the consumer's SiLU already applied);
// output: f32 `y` and / or planes `yp` = split(post_silu ? silu(v) : v) (plain convs only)
void codec_conv1d_bf3(const float* x, const uint16_t* xp, int B, int Cin, int T, const uint16_t* wp, bool f16, const float* bias, int Cout, int K,
                      int dil, bool pre_silu, int epi, const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, int ps,
                      hipStream_t st, const uint16_t* ctx_in = nullptr, uint16_t* ctx_out = nullptr, const float* mean_a = nullptr,
                      const float* mean_b = nullptr);
// mean_a / mean_b (plane-input residual convs only): the ParallelBlock mean folded into the epilogue -- the stored / split value is
// ((mean_a + mean_b) + (res + conv)) / 3, bit-identical to k_mean3_planes / k_mean3 on the three ResBlock outputs.
// Streaming (fs_codec_stream_*): `ctx_in` = the left context of the plane tensor being written ([parts][C/8][CODEC_PLANE_PAD][8], the last
// CODEC_PLANE_PAD slots of the same tensor in the previous chunk; null = zeros, i.e. the start of a signal), `ctx_out` receives this
// chunk's last CODEC_PLANE_PAD slots.  B == 1 and T >= CODEC_PLANE_PAD.
void codec_act_split(const float* x, int B, int C, int T, bool silu, uint16_t* planes, bool f16, hipStream_t st, const uint16_t* ctx_in = nullptr,
                     uint16_t* ctx_out = nullptr);
void codec_mean3_planes(const float* a, const float* b, const float* c, int B, int C, int T, bool silu, uint16_t* planes, bool f16, hipStream_t st,
                        const uint16_t* ctx_in = nullptr, uint16_t* ctx_out = nullptr);
// conv / transposed conv of the plane data flow (decode path, bf16x3 mode): see codec_conv1d_bf3
void codec_conv1d_planes(const float* x, const uint16_t* xp, int B, int Cin, int T, const ConvW& w, int dil, bool pre_silu, int epi,
                         const float* res, const float* gamma, float* y, uint16_t* yp, bool post_silu, hipStream_t st,
                         const uint16_t* ctx_in = nullptr, uint16_t* ctx_out = nullptr, const float* mean_a = nullptr, const float* mean_b = nullptr);
void codec_tconv1d_planes(const uint16_t* xp, int B, int Cin, int Tin, const ConvW& w, int stride, float* y, hipStream_t st);
// one ResBlock pair x' = x + conv2(silu(conv1(silu(x)))) of a thin stage (C = 16 | 32, f16 mode) in one kernel: the intermediate stays in LDS
// (codec_conv_bf3.hip: k_respair_f16t).  xp = planes of silu(x), res = x (f32); outputs as codec_conv1d_bf3's residual conv.  mid_ctx_*: the
// streaming context of the INTERMEDIATE's planes (the tensor that is no longer written), ctx_*: of the output planes.
bool codec_respair_ok(int C, int K, int dil, bool f16);
void codec_respair_f16(const uint16_t* xp, int B, int C, int T, const uint16_t* w1p, const float* b1, const uint16_t* w2p, const float* b2, int K, int dil,
                       const float* res, float* y, uint16_t* yp, hipStream_t st, const uint16_t* mid_ctx_in = nullptr, uint16_t* mid_ctx_out = nullptr,
                       const uint16_t* ctx_in = nullptr, uint16_t* ctx_out = nullptr, const float* mean_a = nullptr, const float* mean_b = nullptr);
// f16 range diagnostic (fs_codec_set_range_check): while on, the launchers above run the range-checked twin of codec_conv_bf3.hip, whose
// f32 -> f16 conversions count saturated (|x| > 65504) and flushed (0 < |x| < 2^-24) operands; reset / read the two device counters
void codec_range_check(bool on);
void codec_range_reset(hipStream_t st);
void codec_range_read(unsigned long long* out2, hipStream_t st);
// ---- encoder side (FireflyCodec::encode)
void codec_stft_mag(const float* pcm, int n, int n_fft, int hop, int n_frames, float* lin /*[n_fft/2+1][frames]*/, hipStream_t st);
void codec_mel_log(const float* lin, const float* fb /*[nf][n_mels]*/, int nf, int n_mels, int F, float* mel /*[n_mels][F]*/, hipStream_t st);
void codec_layernorm_cf(const float* x, int C, int T, const float* w, const float* b, float* y, hipStream_t st);
void codec_space_to_depth(const float* x, int C, int T, int s, float* y /*[C*s][T/s]*/, hipStream_t st);
void codec_fsq_encode(const float* z, int C, int T, int G, const float* pin_w, const float* pin_b, uint32_t* codes, hipStream_t st);
void codec_synth_fill(float* dst, uint64_t key, size_t n, float mean, float scale, hipStream_t st);

}  // namespace fs
