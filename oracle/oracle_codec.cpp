// TEST INFRASTRUCTURE ONLY (oracle/).  PARITY UNPINNED against the reference binary (see oracle_lm.h).
// CPU f32 restatement of FireflyCodec::decode (Fish 1.4 / 1.5 configuration), op-for-op with:
//   fish_speech_core/lib/codec/firefly.rs:42-48, decoder.rs:37-68, quantizer.rs:126-146,
//   grouped_residual_fsq.rs:95-114,175-185, fsq.rs:40-66,119-159, convnext.rs:110-126,
//   hifi_gan.rs:74-85,114-117,208-216, utils/mod.rs:53-62,110-122, config.rs:98-113,155-167
#include "oracle_codec.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "fsgen.h"

namespace oracle {

// candle conv1d on a causally left-padded input (FishConvNet::forward, utils/mod.rs:53-62):
// pad_left = (k-1)*dil + 1 - stride, no right pad.  x: (Cin, T) -> y: (Cout, Tout), groups in {1, Cin}.
static void fish_conv1d(const float* x, int Cin, int T, const Conv& c, int dil, int groups, std::vector<float>& y,
                        int& Tout) {
    const int k = c.k, Cout = c.cout, stride = 1;
    const int pad = (k - 1) * dil + 1 - stride;
    const int Tp = T + pad;
    Tout = (Tp - ((k - 1) * dil + 1)) / stride + 1;  // == T
    std::vector<float> xp((size_t)Cin * Tp, 0.f);
    for (int i = 0; i < Cin; ++i) std::memcpy(&xp[(size_t)i * Tp + pad], &x[(size_t)i * T], sizeof(float) * T);
    y.assign((size_t)Cout * Tout, 0.f);
    const int cin_g = Cin / groups, cout_g = Cout / groups;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
        float* yo = &y[(size_t)o * Tout];
        const int g = o / cout_g;
        for (int t = 0; t < Tout; ++t) yo[t] = 0.f;
        for (int ii = 0; ii < cin_g; ++ii) {
            const float* xi = &xp[(size_t)(g * cin_g + ii) * Tp];
            for (int kk = 0; kk < k; ++kk) {
                const float w = c.w[((size_t)o * cin_g + ii) * k + kk];
                const float* xs = xi + kk * dil;
#pragma omp simd
                for (int t = 0; t < Tout; ++t) yo[t] += w * xs[t];
            }
        }
        const float b = c.b[o];
        for (int t = 0; t < Tout; ++t) yo[t] += b;
    }
}

// candle conv_transpose1d then trim `k - stride` samples on the right (FishTransConvNet::forward, utils/mod.rs:110-122).
// weight layout [Cin, Cout, k].
static void fish_conv_transpose1d(const float* x, int Cin, int T, const Conv& c, int stride, std::vector<float>& y,
                                  int& Tout) {
    const int k = c.k, Cout = c.cout;
    const int Tfull = (T - 1) * stride + k;
    const int trim = k > stride ? k - stride : 0;
    Tout = Tfull - trim;
    y.assign((size_t)Cout * Tout, 0.f);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
        std::vector<float> full((size_t)Tfull, 0.f);
        for (int i = 0; i < Cin; ++i) {
            const float* xi = &x[(size_t)i * T];
            for (int kk = 0; kk < k; ++kk) {
                const float w = c.w[((size_t)i * Cout + o) * k + kk];
                for (int t = 0; t < T; ++t) full[(size_t)t * stride + kk] += xi[t] * w;
            }
        }
        for (int t = 0; t < Tout; ++t) y[(size_t)o * Tout + t] = full[t] + c.b[o];
    }
}

static inline float silu(float x) { return x / (1.f + std::exp(-x)); }
// candle Tensor::gelu == tanh approximation
static inline float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f;  // sqrt(2/pi)
    return 0.5f * x * (1.f + std::tanh(k0 * x * (1.f + 0.044715f * x * x)));
}

// convnext.rs:110-126
static void convnext_block(const ConvNeXt& b, std::vector<float>& x, int C, int T) {
    std::vector<float> h;
    int To;
    fish_conv1d(x.data(), C, T, b.dwconv, 1, C, h, To);
    const int Hd = (int)b.pw1_b.size();
    std::vector<float> out((size_t)C * T);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        std::vector<float> v(C), u(Hd);
        float mean = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = h[(size_t)c * T + t]; mean += v[c]; }
        mean /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) { float d = v[c] - mean; var += d * d; }
        var /= (float)C;
        const float inv = 1.f / std::sqrt(var + 1e-6f);
        for (int c = 0; c < C; ++c) v[c] = (v[c] - mean) * inv * b.norm_w[c] + b.norm_b[c];
        for (int j = 0; j < Hd; ++j) {
            float acc = 0.f;
            const float* w = &b.pw1_w[(size_t)j * C];
            for (int c = 0; c < C; ++c) acc += v[c] * w[c];
            u[j] = gelu_tanh(acc + b.pw1_b[j]);
        }
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
            const float* w = &b.pw2_w[(size_t)c * Hd];
            for (int j = 0; j < Hd; ++j) acc += u[j] * w[j];
            out[(size_t)c * T + t] = x[(size_t)c * T + t] + b.gamma[c] * (acc + b.pw2_b[c]);
        }
    }
    x.swap(out);
}

void Codec::init_fish15() {
    // codec/config.rs:155-167 (firefly_1_4 == 1_5), :98-113 (HiFiGAN)
    n_groups = 8; levels = {8, 5, 5, 5}; input_dim = 512; downsample = {2, 2};
    up_rates = {8, 8, 2, 2, 2}; up_kernels = {16, 16, 4, 4, 4}; res_kernels = {3, 7, 11}; res_dils = {1, 3, 5};
    init_ch = 512; pre_k = 13; post_k = 13;
}

void Codec::init_tiny() {  // channels / 8, same topology (tests only)
    init_fish15();
    input_dim = 64; init_ch = 64;
}

static void fill_conv(Conv& c, int cout, int cin_per_group, int k, bool transpose_layout, const std::string& name,
                      uint64_t seed) {
    c.cout = cout; c.k = k;
    const size_t n = (size_t)cout * cin_per_group * k;
    c.w.resize(n); c.b.resize(cout);
    // N(0, 1/fan_in), fan_in = cin_per_group * k (transposed conv: effective taps per output = cin * k / stride; keep cin*k)
    (void)transpose_layout;
    fsgen::fill(c.w.data(), n, name + ".conv.weight", seed, 0.f, 1.0 / std::sqrt((double)cin_per_group * k), false);
    fsgen::fill(c.b.data(), cout, name + ".conv.bias", seed, 0.f, 0.02, false);
}

void Codec::load_synthetic(uint64_t seed) {
    const int dg = input_dim / n_groups;
    proj_w.resize(n_groups); proj_b.resize(n_groups);
    for (int g = 0; g < n_groups; ++g) {
        std::string p = "quantizer.residual_fsq.rvqs." + std::to_string(g) + ".project_out";
        proj_w[g].resize((size_t)dg * 4); proj_b[g].resize(dg);
        fsgen::fill(proj_w[g].data(), proj_w[g].size(), p + ".weight", seed, 0.f, 0.5, false);
        fsgen::fill(proj_b[g].data(), dg, p + ".bias", seed, 0.f, 0.02, false);
    }
    const int C = input_dim;
    up_conv.resize(downsample.size()); up_block.resize(downsample.size());
    for (size_t i = 0; i < downsample.size(); ++i) {
        std::string p = "quantizer.upsample." + std::to_string(i);
        // ConvTranspose1d weight [in, out, k]
        up_conv[i].cout = C; up_conv[i].k = downsample[i];
        up_conv[i].w.resize((size_t)C * C * downsample[i]); up_conv[i].b.resize(C);
        fsgen::fill(up_conv[i].w.data(), up_conv[i].w.size(), p + ".0.conv.weight", seed, 0.f, 1.0 / std::sqrt((double)C), false);
        fsgen::fill(up_conv[i].b.data(), C, p + ".0.conv.bias", seed, 0.f, 0.02, false);
        ConvNeXt& b = up_block[i];
        std::string q = p + ".1";
        fill_conv(b.dwconv, C, 1, 7, false, q + ".dwconv", seed);
        b.norm_w.resize(C); b.norm_b.resize(C); b.gamma.resize(C);
        fsgen::fill(b.norm_w.data(), C, q + ".norm.weight", seed, 1.f, 0.1, false);
        fsgen::fill(b.norm_b.data(), C, q + ".norm.bias", seed, 0.f, 0.02, false);
        b.pw1_w.resize((size_t)4 * C * C); b.pw1_b.resize(4 * C); b.pw2_w.resize((size_t)4 * C * C); b.pw2_b.resize(C);
        fsgen::fill(b.pw1_w.data(), b.pw1_w.size(), q + ".pwconv1.weight", seed, 0.f, 1.0 / std::sqrt((double)C), false);
        fsgen::fill(b.pw1_b.data(), 4 * C, q + ".pwconv1.bias", seed, 0.f, 0.02, false);
        fsgen::fill(b.pw2_w.data(), b.pw2_w.size(), q + ".pwconv2.weight", seed, 0.f, 1.0 / std::sqrt(4.0 * C), false);
        fsgen::fill(b.pw2_b.data(), C, q + ".pwconv2.bias", seed, 0.f, 0.02, false);
        fsgen::fill(b.gamma.data(), C, q + ".gamma", seed, 0.1f, 0.02, false);
    }
    fill_conv(conv_pre, init_ch, input_dim, pre_k, false, "head.conv_pre", seed);
    const int ns = (int)up_rates.size();
    ups.resize(ns); res.resize(ns);
    for (int i = 0; i < ns; ++i) {
        const int cin = init_ch >> i, cout = init_ch >> (i + 1);
        std::string p = "head.ups." + std::to_string(i);
        ups[i].cout = cout; ups[i].k = up_kernels[i];
        ups[i].w.resize((size_t)cin * cout * up_kernels[i]); ups[i].b.resize(cout);
        // each output sample sees cin * k / stride taps
        fsgen::fill(ups[i].w.data(), ups[i].w.size(), p + ".conv.weight", seed, 0.f,
                    1.0 / std::sqrt((double)cin * up_kernels[i] / up_rates[i]), false);
        fsgen::fill(ups[i].b.data(), cout, p + ".conv.bias", seed, 0.f, 0.02, false);
        res[i].resize(res_kernels.size());
        for (size_t j = 0; j < res_kernels.size(); ++j) {
            res[i][j].c1.resize(res_dils.size()); res[i][j].c2.resize(res_dils.size());
            for (size_t m = 0; m < res_dils.size(); ++m) {
                std::string q = "head.resblocks." + std::to_string(i) + ".blocks." + std::to_string(j);
                fill_conv(res[i][j].c1[m], cout, cout, res_kernels[j], false, q + ".convs1." + std::to_string(m), seed);
                fill_conv(res[i][j].c2[m], cout, cout, res_kernels[j], false, q + ".convs2." + std::to_string(m), seed);
            }
        }
    }
    fill_conv(conv_post, 1, init_ch >> ns, post_k, false, "head.conv_post", seed);
}

// fsq.rs:137-144 + :119-122 : code[k] = ((floor(idx / basis_k) mod levels_k) - hw_k) / hw_k
void Codec::fsq_code(uint32_t idx, float* code4) const {
    int basis = 1;
    for (size_t k = 0; k < levels.size(); ++k) {
        const int lv = levels[k];
        const float li = (float)((idx / basis) % lv);
        const float hw = std::floor((float)lv / 2.f);
        code4[k] = (li - hw) / hw;
        basis *= lv;
    }
}

// decoder.rs:37-68 for one full-length item (masks are all ones).  codes: (n_groups, T) u32 -> pcm (2048*T) f32.
std::vector<float> Codec::decode(const uint32_t* codes, int T, std::vector<std::vector<float>>* stages) const {
    const int dg = input_dim / n_groups, C = input_dim;
    // quantizer.decode (quantizer.rs:135-146): per group gather + project_out, concat -> (T, C) -> transpose (C, T)
    std::vector<float> z((size_t)C * T);
    for (int g = 0; g < n_groups; ++g)
        for (int t = 0; t < T; ++t) {
            const uint32_t idx = codes[(size_t)g * T + t];
            if (idx >= 1000u) throw std::runtime_error("FSQ index out of range");
            float code[4];
            fsq_code(idx, code);
            for (int o = 0; o < dg; ++o) {
                float acc = 0.f;
                for (int k = 0; k < 4; ++k) acc += code[k] * proj_w[g][(size_t)o * 4 + k];
                z[(size_t)(g * dg + o) * T + t] = acc + proj_b[g][o];
            }
        }
    if (stages) stages->push_back(z);
    // upsample (quantizer.rs:126-133): upsample.0 then upsample.1
    int Tc = T;
    for (size_t i = 0; i < up_conv.size(); ++i) {
        std::vector<float> y;
        int To;
        fish_conv_transpose1d(z.data(), C, Tc, up_conv[i], downsample[i], y, To);
        z.swap(y); Tc = To;
        convnext_block(up_block[i], z, C, Tc);
        if (stages) stages->push_back(z);
    }
    // HiFiGAN::forward (hifi_gan.rs:208-216)
    std::vector<float> x;
    int To;
    fish_conv1d(z.data(), C, Tc, conv_pre, 1, 1, x, To);
    if (stages) stages->push_back(x);
    int ch = init_ch;
    for (size_t i = 0; i < ups.size(); ++i) {
        for (auto& v : x) v = silu(v);
        std::vector<float> y;
        fish_conv_transpose1d(x.data(), ch, Tc, ups[i], up_rates[i], y, To);
        ch >>= 1; Tc = To;
        // ParallelBlock: mean over 3 ResBlock1 (hifi_gan.rs:114-117): stack(...).mean(0) = sum * (1/3)
        std::vector<float> acc((size_t)ch * Tc, 0.f);
        for (size_t j = 0; j < res[i].size(); ++j) {
            std::vector<float> r = y;
            for (size_t m = 0; m < res_dils.size(); ++m) {  // ResBlock1::forward (:74-85); both convs dilated (:58-61)
                std::vector<float> xt = r, t1, t2;
                for (auto& v : xt) v = silu(v);
                int tt;
                fish_conv1d(xt.data(), ch, Tc, res[i][j].c1[m], res_dils[m], 1, t1, tt);
                for (auto& v : t1) v = silu(v);
                fish_conv1d(t1.data(), ch, Tc, res[i][j].c2[m], res_dils[m], 1, t2, tt);
                for (size_t e = 0; e < r.size(); ++e) r[e] = r[e] + t2[e];
            }
            for (size_t e = 0; e < r.size(); ++e) acc[e] += r[e];
        }
        const float third = (float)(1.0 / 3.0);
        for (auto& v : acc) v *= third;
        x.swap(acc);
        if (stages) stages->push_back(x);
    }
    for (auto& v : x) v = silu(v);
    std::vector<float> pcm;
    fish_conv1d(x.data(), ch, Tc, conv_post, 1, 1, pcm, To);
    for (auto& v : pcm) v = std::tanh(v);
    return pcm;
}

}  // namespace oracle
