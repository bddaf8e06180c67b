// TEST INFRASTRUCTURE ONLY (oracle/).  PARITY UNPINNED against the reference binary (see oracle_lm.h).
// CPU f32 restatement of FireflyCodec::decode (Fish 1.4 / 1.5 configuration), op-for-op with:
//   fish_speech_core/lib/codec/firefly.rs:42-48, decoder.rs:37-68, quantizer.rs:126-146,
//   grouped_residual_fsq.rs:95-114,175-185, fsq.rs:40-66,119-159, convnext.rs:110-126,
//   hifi_gan.rs:74-85,114-117,208-216, utils/mod.rs:53-62,110-122, config.rs:98-113,155-167
#include "oracle_codec.h"

#include <algorithm>
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
    enc_dims = {128, 256, 384, 512}; enc_depths = {3, 3, 9, 3};  // BackboneConfig::fish_1_4 (config.rs:47-57)
}

void Codec::init_tiny() {  // channels / 8, same topology (tests only)
    init_fish15();
    input_dim = 64; init_ch = 64;
    enc_dims = {16, 32, 48, 64}; enc_depths = {1, 1, 2, 1};
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

// decode side only (what a "hard checkpoint" test overrides): name -> the vector load_synthetic sized
void Codec::set_tensor(const std::string& name, const float* data, size_t n) {
    std::vector<float>* dst = nullptr;
    auto conv = [&](Conv& c, const std::string& prefix) {
        if (name == prefix + ".conv.weight") dst = &c.w;
        else if (name == prefix + ".conv.bias") dst = &c.b;
    };
    for (int g = 0; g < n_groups && !dst; ++g) {
        const std::string p = "quantizer.residual_fsq.rvqs." + std::to_string(g) + ".project_out";
        if (name == p + ".weight") dst = &proj_w[g];
        else if (name == p + ".bias") dst = &proj_b[g];
    }
    for (size_t i = 0; i < up_conv.size() && !dst; ++i) {
        const std::string p = "quantizer.upsample." + std::to_string(i);
        conv(up_conv[i], p + ".0");
        ConvNeXt& b = up_block[i];
        conv(b.dwconv, p + ".1.dwconv");
        if (name == p + ".1.norm.weight") dst = &b.norm_w;
        else if (name == p + ".1.norm.bias") dst = &b.norm_b;
        else if (name == p + ".1.pwconv1.weight") dst = &b.pw1_w;
        else if (name == p + ".1.pwconv1.bias") dst = &b.pw1_b;
        else if (name == p + ".1.pwconv2.weight") dst = &b.pw2_w;
        else if (name == p + ".1.pwconv2.bias") dst = &b.pw2_b;
        else if (name == p + ".1.gamma") dst = &b.gamma;
    }
    if (!dst) conv(conv_pre, "head.conv_pre");
    if (!dst) conv(conv_post, "head.conv_post");
    for (size_t i = 0; i < ups.size() && !dst; ++i) {
        conv(ups[i], "head.ups." + std::to_string(i));
        for (size_t j = 0; j < res[i].size() && !dst; ++j)
            for (size_t m = 0; m < res[i][j].c1.size() && !dst; ++m) {
                const std::string q = "head.resblocks." + std::to_string(i) + ".blocks." + std::to_string(j);
                conv(res[i][j].c1[m], q + ".convs1." + std::to_string(m));
                if (!dst) conv(res[i][j].c2[m], q + ".convs2." + std::to_string(m));
            }
    }
    if (!dst) throw std::runtime_error("oracle codec: unknown decode-side tensor " + name);
    if (dst->size() != n) throw std::runtime_error("oracle codec: " + name + " has " + std::to_string(dst->size()) + " elements, got " + std::to_string(n));
    std::copy(data, data + n, dst->begin());
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
    // ---- encoder (names of convnext.rs:186-271, quantizer.rs:44-66, grouped_residual_fsq.rs:52-56)
    auto fill_block = [&](ConvNeXt& b, int Cb, const std::string& q) {
        fill_conv(b.dwconv, Cb, 1, 7, false, q + ".dwconv", seed);
        b.norm_w.resize(Cb); b.norm_b.resize(Cb); b.gamma.resize(Cb);
        fsgen::fill(b.norm_w.data(), Cb, q + ".norm.weight", seed, 1.f, 0.1, false);
        fsgen::fill(b.norm_b.data(), Cb, q + ".norm.bias", seed, 0.f, 0.02, false);
        b.pw1_w.resize((size_t)4 * Cb * Cb); b.pw1_b.resize(4 * Cb); b.pw2_w.resize((size_t)4 * Cb * Cb); b.pw2_b.resize(Cb);
        fsgen::fill(b.pw1_w.data(), b.pw1_w.size(), q + ".pwconv1.weight", seed, 0.f, 1.0 / std::sqrt((double)Cb), false);
        fsgen::fill(b.pw1_b.data(), 4 * Cb, q + ".pwconv1.bias", seed, 0.f, 0.02, false);
        fsgen::fill(b.pw2_w.data(), b.pw2_w.size(), q + ".pwconv2.weight", seed, 0.f, 1.0 / std::sqrt(4.0 * Cb), false);
        fsgen::fill(b.pw2_b.data(), Cb, q + ".pwconv2.bias", seed, 0.f, 0.02, false);
        fsgen::fill(b.gamma.data(), Cb, q + ".gamma", seed, 0.1f, 0.02, false);
    };
    auto fill_ln = [&](std::vector<float>& w, std::vector<float>& b, int n, const std::string& q) {
        w.resize(n); b.resize(n);
        fsgen::fill(w.data(), n, q + ".weight", seed, 1.f, 0.1, false);
        fsgen::fill(b.data(), n, q + ".bias", seed, 0.f, 0.02, false);
    };
    const int nst = (int)enc_dims.size();
    fill_conv(stem_conv, enc_dims[0], n_mels, enc_k, false, "backbone.downsample_layers.0.0", seed);
    fill_ln(stem_ln_w, stem_ln_b, enc_dims[0], "backbone.downsample_layers.0.1");
    mid_ln_w.assign(nst, {}); mid_ln_b.assign(nst, {}); mid_conv.assign(nst, Conv());
    stages.assign(nst, {});
    for (int i = 0; i < nst; ++i) {
        if (i > 0) {
            const std::string q = "backbone.downsample_layers." + std::to_string(i);
            fill_ln(mid_ln_w[i], mid_ln_b[i], enc_dims[i - 1], q + ".0");
            Conv& c = mid_conv[i];  // plain Conv1d: tensors `1.weight` / `1.bias` (convnext.rs:229-233)
            c.cout = enc_dims[i]; c.k = 1;
            c.w.resize((size_t)enc_dims[i] * enc_dims[i - 1]); c.b.resize(enc_dims[i]);
            fsgen::fill(c.w.data(), c.w.size(), q + ".1.weight", seed, 0.f, 1.0 / std::sqrt((double)enc_dims[i - 1]), false);
            fsgen::fill(c.b.data(), c.b.size(), q + ".1.bias", seed, 0.f, 0.02, false);
        }
        stages[i].resize(enc_depths[i]);
        for (int j = 0; j < enc_depths[i]; ++j) fill_block(stages[i][j], enc_dims[i], "backbone.stages." + std::to_string(i) + "." + std::to_string(j));
    }
    fill_ln(enc_norm_w, enc_norm_b, enc_dims[nst - 1], "backbone.norm");
    down_conv.resize(downsample.size()); down_block.resize(downsample.size());
    for (size_t i = 0; i < downsample.size(); ++i) {
        const std::string q = "quantizer.downsample." + std::to_string(i);
        fill_conv(down_conv[i], C, C, downsample[i], false, q + ".0", seed);
        fill_block(down_block[i], C, q + ".1");
    }
    pin_w.resize(n_groups); pin_b.resize(n_groups);
    for (int g = 0; g < n_groups; ++g) {
        const std::string q = "quantizer.residual_fsq.rvqs." + std::to_string(g) + ".project_in";
        pin_w[g].resize((size_t)4 * dg); pin_b[g].resize(4);
        fsgen::fill(pin_w[g].data(), pin_w[g].size(), q + ".weight", seed, 0.f, 1.0 / std::sqrt((double)dg), false);
        fsgen::fill(pin_b[g].data(), 4, q + ".bias", seed, 0.f, 0.02, false);
    }
    mel_fb = mel_filterbank(44100, n_fft, n_mels);
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


// ================================================================================================ encoder side
// Slaney mel filterbank (librosa.filters.mel, norm = "slaney", htk = False) for f_min = 0, f_max = sr / 2, evaluated in f64
// and rounded to f32, laid out [n_fft/2+1][n_mels] as load_mel_buffer reads it (spectrogram.rs:90-101).  The reference embeds
// this table as a binary resource (melfilters160.bytes); regenerated here from the published formula -- max |diff| to the
// embedded table 1.8e-7 (tests/test_codec_encode.py checks a committed sample of the table).
std::vector<float> Codec::mel_filterbank(int sr, int n_fft, int n_mels) {
    const int nf = n_fft / 2 + 1;
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
    std::vector<float> fb((size_t)nf * n_mels, 0.f);
    for (int m = 0; m < n_mels; ++m) {
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        for (int f = 0; f < nf; ++f) {
            const double freq = (sr / 2.0) * (double)f / (double)(nf - 1);
            const double lower = (freq - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - freq) / (mel_f[m + 2] - mel_f[m + 1]);
            const double w = std::max(0.0, std::min(lower, upper));
            fb[(size_t)f * n_mels + m] = (float)(w * enorm);
        }
    }
    return fb;
}

// in-place iterative radix-2 FFT (forward, f64) -- the transform rustfft computes for the reference (stft.rs:82-83)
static void fft_f64(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

std::vector<float> Codec::log_mel(const float* pcm, int n, int* frames_out) const {
    const int pad = (n_fft - hop_length) / 2, nf = n_fft / 2 + 1;
    if (n < pad) throw std::runtime_error("input shorter than the reflect padding (the reference slices out of range)");
    // reflect_pad (spectrogram.rs:15-27): the edge samples ARE repeated (signal[0..pad] reversed)
    std::vector<float> x((size_t)n + 2 * pad);
    for (int i = 0; i < pad; ++i) x[i] = pcm[pad - 1 - i];
    std::memcpy(&x[pad], pcm, sizeof(float) * n);
    for (int i = 0; i < pad; ++i) x[(size_t)pad + n + i] = pcm[n - 1 - i];
    const long long Lp = (long long)x.size();
    // streaming STFT (stft.rs:52-90): a frame is emitted once >= n_fft samples were pushed; frame f = padded[f*hop, f*hop + n_fft),
    // the final partial hop zero-filled
    const long long full = Lp / hop_length, rem = Lp % hop_length;
    long long n_frames = std::max(0LL, full - (n_fft / hop_length - 1));
    if (rem > 0 && full * hop_length + rem >= n_fft) n_frames += 1;
    *frames_out = (int)n_frames;
    std::vector<double> win(n_fft);
    for (int i = 0; i < n_fft; ++i) win[i] = 0.5 * (1.0 - std::cos((2.0 * M_PI * (double)i) / (double)n_fft));
    std::vector<float> lin((size_t)nf * n_frames);  // channel-first [nf][frames]
#pragma omp parallel for schedule(static)
    for (long long f = 0; f < n_frames; ++f) {
        std::vector<double> re(n_fft), im(n_fft, 0.0);
        for (int j = 0; j < n_fft; ++j) {
            const long long idx = f * hop_length + j;
            re[j] = (idx < Lp ? (double)x[idx] : 0.0) * win[j];
        }
        fft_f64(re, im);
        for (int k = 0; k < nf; ++k) lin[(size_t)k * n_frames + f] = (float)std::sqrt(re[k] * re[k] + im[k] * im[k]) + 1e-6f;
    }
    // apply_mel_scale + compress (spectrogram.rs:136-151): (frames, nf) . (nf, n_mels) -> transpose; clamp(1e-5, 100).log()
    std::vector<float> mel((size_t)n_mels * n_frames);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < n_mels; ++m)
        for (long long f = 0; f < n_frames; ++f) {
            float acc = 0.f;
            for (int k = 0; k < nf; ++k) acc += lin[(size_t)k * n_frames + f] * mel_fb[(size_t)k * n_mels + m];
            mel[(size_t)m * n_frames + f] = std::log(std::min(std::max(acc, 1e-5f), 100.0f));
        }
    return mel;
}

// LayerNormChannelsFirst (convnext.rs:144-154): per time step over channels, eps 1e-6
static void layernorm_cf(std::vector<float>& x, int C, int T, const std::vector<float>& w, const std::vector<float>& b) {
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        float mean = 0.f;
        for (int c = 0; c < C; ++c) mean += x[(size_t)c * T + t];
        mean /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) { const float d = x[(size_t)c * T + t] - mean; var += d * d; }
        var /= (float)C;
        const float sd = std::sqrt(var + 1e-6f);
        for (int c = 0; c < C; ++c) x[(size_t)c * T + t] = (x[(size_t)c * T + t] - mean) / sd * w[c] + b[c];
    }
}

// FishConvNet with stride (utils/mod.rs:53-62): left pad (k-1)+1-stride zeros, then conv1d(stride)
static void fish_conv1d_strided(const float* x, int Cin, int T, const Conv& c, int stride, std::vector<float>& y, int& Tout) {
    const int k = c.k, Cout = c.cout, pad = k - stride;
    const int Tp = T + pad;
    Tout = (Tp - k) / stride + 1;
    y.assign((size_t)Cout * Tout, 0.f);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o)
        for (int t = 0; t < Tout; ++t) {
            float acc = 0.f;
            for (int i = 0; i < Cin; ++i)
                for (int kk = 0; kk < k; ++kk) {
                    const int src = t * stride + kk - pad;
                    if (src >= 0) acc += c.w[((size_t)o * Cin + i) * k + kk] * x[(size_t)i * T + src];
                }
            y[(size_t)o * Tout + t] = acc + c.b[o];
        }
}

// plain candle Conv1d k = 1 (MidLayer, convnext.rs:229-238)
static void conv1x1(const std::vector<float>& x, int Cin, int T, const Conv& c, std::vector<float>& y) {
    y.assign((size_t)c.cout * T, 0.f);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < c.cout; ++o) {
        float* yo = &y[(size_t)o * T];
        for (int i = 0; i < Cin; ++i) {
            const float w = c.w[(size_t)o * Cin + i];
            const float* xi = &x[(size_t)i * T];
            for (int t = 0; t < T; ++t) yo[t] += w * xi[t];
        }
        for (int t = 0; t < T; ++t) yo[t] += c.b[o];
    }
}

// FSQ::bound (fsq.rs:68-84) for one coordinate of level count lv, all f32 as candle evaluates it
static inline float fsq_bound(float z, int lv) {
    const float half_l = ((float)lv - 1.0f) * 1.001f / 2.0f;
    const float offset = (lv % 2 == 0) ? 0.5f : 0.0f;
    const float q = offset / half_l;
    const float shift = std::log((1.0f + q) / (1.0f - q)) * 0.5f;  // atanh as the reference spells it (fsq.rs:20-25)
    return std::tanh(z + shift) * half_l - offset;
}

std::vector<uint32_t> Codec::encode_mel(const std::vector<float>& mel, int frames, int* L_out, std::vector<std::vector<float>>* so) const {
    int T = frames, To;
    // ConvNeXtEncoder (convnext.rs:319-331): stem conv + LN + blocks, then (LN + 1x1 conv + blocks) x 3, final LN
    std::vector<float> x;
    fish_conv1d(mel.data(), n_mels, T, stem_conv, 1, 1, x, To);
    layernorm_cf(x, enc_dims[0], T, stem_ln_w, stem_ln_b);
    for (const auto& b : stages[0]) convnext_block(b, x, enc_dims[0], T);
    if (so) so->push_back(x);
    for (size_t i = 1; i < enc_dims.size(); ++i) {
        layernorm_cf(x, enc_dims[i - 1], T, mid_ln_w[i], mid_ln_b[i]);
        std::vector<float> y;
        conv1x1(x, enc_dims[i - 1], T, mid_conv[i], y);
        x.swap(y);
        for (const auto& b : stages[i]) convnext_block(b, x, enc_dims[i], T);
        if (so) so->push_back(x);
    }
    layernorm_cf(x, enc_dims.back(), T, enc_norm_w, enc_norm_b);
    if (so) so->push_back(x);
    // quantizer.encode (quantizer.rs:104-124): downsample convs + blocks, then grouped residual FSQ
    const int C = input_dim;
    for (size_t i = 0; i < down_conv.size(); ++i) {
        std::vector<float> y;
        fish_conv1d_strided(x.data(), C, T, down_conv[i], downsample[i], y, To);
        x.swap(y); T = To;
        convnext_block(down_block[i], x, C, T);
        if (so) so->push_back(x);
    }
    // ResidualFSQ::forward with num_quantizers == 1 (grouped_residual_fsq.rs:75-93): project_in, bound, then the layer's own
    // quantize = round(bound(residual / scale)) / half_width (fsq.rs:86-91) -- bound is applied TWICE, as in the reference
    const int dg = C / n_groups;
    std::vector<uint32_t> idx((size_t)n_groups * T);
    std::vector<float> margin((size_t)n_groups * T, 1.f);  // distance of the pre-round value from the nearest x.5 boundary
    for (int g = 0; g < n_groups; ++g)
        for (int t = 0; t < T; ++t) {
            float sum = 0.f;
            int basis = 1;
            for (size_t k = 0; k < levels.size(); ++k) {
                float acc = 0.f;
                for (int c = 0; c < dg; ++c) acc += x[(size_t)(g * dg + c) * T + t] * pin_w[g][k * dg + c];
                acc += pin_b[g][k];
                const float residual = fsq_bound(acc, levels[k]);
                const float hw = std::floor((float)levels[k] / 2.0f);
                const float pre = fsq_bound(residual / 1.0f, levels[k]);
                margin[(size_t)g * T + t] = std::min(margin[(size_t)g * T + t], std::fabs(std::fabs(pre - std::floor(pre)) - 0.5f));
                const float code = std::round(pre) / hw;                                         // quantize
                const float zhat = code * hw + hw;                                               // _scale_and_shift
                sum += zhat * (float)basis;                                                      // codes_to_indices: sum then -> i64
                basis *= levels[k];
            }
            idx[(size_t)g * T + t] = (uint32_t)(long long)sum;
        }
    if (so) so->push_back(margin);  // test aid (last stage): a GPU index may differ only where this is ~0
    *L_out = T;
    return idx;
}

std::vector<uint32_t> Codec::encode(const float* pcm, int n, int* L) const {
    int frames = 0;
    const std::vector<float> mel = log_mel(pcm, n, &frames);
    return encode_mel(mel, frames, L);
}

}  // namespace oracle
