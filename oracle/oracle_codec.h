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

    void init_fish15();
    void init_tiny();
    void load_synthetic(uint64_t seed);
    void fsq_code(uint32_t idx, float* code4) const;
    int hop() const { int h = 1; for (int d : downsample) h *= d; for (int r : up_rates) h *= r; return h; }
    std::vector<float> decode(const uint32_t* codes, int T, std::vector<std::vector<float>>* stages = nullptr) const;
};

}  // namespace oracle
