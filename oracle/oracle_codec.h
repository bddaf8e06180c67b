// TEST INFRASTRUCTURE ONLY (oracle/).  CPU f32 restatement of the Firefly-GAN-VQ vocoder
// (FireflyCodec::decode).  PARITY UNPINNED against the reference binary (see oracle_lm.h); pinned by the
// weight-free FSQ known answers (fsq.rs:53-58,119-144) and by the independent PyTorch restatement
// in tests/golden/make_golden.py.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace oracle {

struct Conv {  // Conv1d weight [cout, cin/groups, k]  or ConvTranspose1d weight [cin, cout, k]
    std::vector<float> w, b;
    int cout = 0, k = 0;
};
struct ConvNeXt {
    Conv dwconv;
    std::vector<float> norm_w, norm_b, pw1_w, pw1_b, pw2_w, pw2_b, gamma;
};
struct ResBlock1 {
    std::vector<Conv> c1, c2;
};

struct Codec {
    int n_groups = 8, input_dim = 512, init_ch = 512, pre_k = 13, post_k = 13;
    std::vector<int> levels, downsample, up_rates, up_kernels, res_kernels, res_dils;
    std::vector<std::vector<float>> proj_w, proj_b;  // per group project_out [dim/groups, 4], [dim/groups]
    std::vector<Conv> up_conv;                        // quantizer.upsample.{i}.0
    std::vector<ConvNeXt> up_block;                   // quantizer.upsample.{i}.1
    Conv conv_pre, conv_post;
    std::vector<Conv> ups;
    std::vector<std::vector<ResBlock1>> res;  // [stage][kernel]

    // ---- encoder side (FireflyCodec::encode, firefly.rs:37-40): mel front-end + ConvNeXt backbone + downsample + FSQ
    int n_mels = 160, n_fft = 2048, hop_length = 512, enc_k = 7;
    std::vector<int> enc_dims, enc_depths;            // backbone dims / depths (config.rs:47-57)
    Conv stem_conv;                                   // backbone.downsample_layers.0.0 (FishConvNet k = 7)
    std::vector<float> stem_ln_w, stem_ln_b;          // backbone.downsample_layers.0.1
    std::vector<std::vector<float>> mid_ln_w, mid_ln_b;  // backbone.downsample_layers.i.0, i = 1..3
    std::vector<Conv> mid_conv;                       // backbone.downsample_layers.i.1 (Conv1d k = 1)
    std::vector<std::vector<ConvNeXt>> stages;        // backbone.stages.i.j
    std::vector<float> enc_norm_w, enc_norm_b;        // backbone.norm
    std::vector<Conv> down_conv;                      // quantizer.downsample.i.0 (FishConvNet k = stride = factor)
    std::vector<ConvNeXt> down_block;                 // quantizer.downsample.i.1
    std::vector<std::vector<float>> pin_w, pin_b;     // per group project_in [4, dim/groups], [4]
    std::vector<float> mel_fb;                        // [n_fft/2+1][n_mels], slaney filterbank (see mel_filterbank)

    static std::vector<float> mel_filterbank(int sample_rate, int n_fft, int n_mels);
    // audio/spectrogram.rs:29-158 + stft.rs:52-90: pcm (n) -> log-mel (n_mels, frames), channel-first
    std::vector<float> log_mel(const float* pcm, int n, int* frames) const;
    // encoder.rs:38-42 + quantizer.rs:104-124: log-mel (n_mels, frames) -> indices (n_groups, L)
    std::vector<uint32_t> encode_mel(const std::vector<float>& mel, int frames, int* L, std::vector<std::vector<float>>* stages_out = nullptr) const;
    std::vector<uint32_t> encode(const float* pcm, int n, int* L) const;

    void init_fish15();
    void init_tiny();
    void load_synthetic(uint64_t seed);
    // decode-side tensor by its checkpoint name (the reference's VarBuilder paths: quantizer.rs:69-94, hifi_gan.rs:128-205, codec/utils/mod.rs:28-40);
    // the codec must have been sized by load_synthetic first; throws on an unknown name or a size mismatch
    void set_tensor(const std::string& name, const float* data, size_t n);
    void fsq_code(uint32_t idx, float* code4) const;
    int hop() const { int h = 1; for (int d : downsample) h *= d; for (int r : up_rates) h *= r; return h; }
    std::vector<float> decode(const uint32_t* codes, int T, std::vector<std::vector<float>>* stages = nullptr) const;
};

}  // namespace oracle
