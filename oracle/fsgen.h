// TEST INFRASTRUCTURE ONLY (oracle/): deterministic synthetic-weight generator, CPU side.
//
// There are no Fish-Speech checkpoints in the build container or on the GPU box
// ("no model weights ... anywhere on disk"), so parity is established on
// synthetic weights at the true tensor names/shapes of the reference loader
// (fish_speech_core/lib/lm/dual_ar.rs:460-529, codec/*.rs load fns).
//
// The generator is specified so that it is BIT-reproducible on any IEEE-754 machine
// (no libm calls): element i of tensor `name` is
//     h   = splitmix64_mix(fnv1a64(name) ^ seed  +  (i+1) * 0x9E3779B97F4A7C15)
//     s   = (h & 0xFFFF) + ((h>>16)&0xFFFF) + ((h>>32)&0xFFFF) + (h>>48) - 131070   (Irwin-Hall n=4)
//     val = mean + (float)s * (float)(std / 37837.2272)          [one f32 multiply, one f32 add]
// The product library has its own copy of this spec (csrc/fs_synth.h); the two are
// cross-checked by tests/test_oracle_known_answers.py::test_synth_generator_spec (spec values) and tests/test_lm_gpu.py::test_synthetic_weights_match_oracle_spec (vs the product).  The oracle never ships in the product path.
#pragma once
#include <cstdint>
#include <cmath>
#include <cstring>
#include <string>

namespace fsgen {

static inline uint64_t fnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ull; }
    return h;
}
static inline uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline float elem(uint64_t key, uint64_t i, float mean, float scale) {
    uint64_t h = mix(key + (i + 1) * 0x9E3779B97F4A7C15ull);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) +
                (int32_t)(h >> 48) - 131070;
    float v = (float)s * scale;
    return mean + v;
}
static inline float scale_for(double stdv) { return (float)(stdv / 37837.2272); }

// round-to-nearest-even f32 -> bf16 -> f32 (what a bf16 checkpoint tensor holds)
static inline float round_bf16(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    r &= 0xFFFF0000u;
    float o; std::memcpy(&o, &r, 4);
    return o;
}

// f32 -> OCP e4m3fn (RNE, saturating) -> f32: what an fp8 checkpoint tensor holds after the per-row scale.  Same integer
// algorithm as the product's quantiser (csrc/fs_common.h); cross-checked on the GPU by tests/test_fp8_gpu.py.
static inline uint8_t f32_to_e4m3(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint8_t)(sign | 0x7F);
    float af; std::memcpy(&af, &a, 4);
    if (af >= 464.0f) return (uint8_t)(sign | 0x7E);
    if (af < 0.0009765625f) return sign;
    int e = (int)(a >> 23) - 127;
    if (e < -6) return (uint8_t)(sign | (uint32_t)__builtin_rintf(af * 512.0f));
    uint32_t mant = a & 0x7FFFFFu, keep = mant >> 20, rest = mant & 0xFFFFFu;
    if (rest > 0x80000u || (rest == 0x80000u && (keep & 1u))) keep += 1;
    if (keep == 8) { keep = 0; e += 1; }
    if (e > 8) return (uint8_t)(sign | 0x7E);
    uint8_t out = (uint8_t)(sign | ((uint32_t)(e + 7) << 3) | keep);
    if ((out & 0x7F) == 0x7F) out = (uint8_t)(sign | 0x7E);
    return out;
}
static inline float e4m3_to_f32(uint8_t b) {
    const uint32_t sign = (uint32_t)(b & 0x80) << 24, e = (b >> 3) & 0xF, m = b & 7;
    float f;
    if (e == 0) f = (float)m * 0.001953125f;
    else { const uint32_t u = ((e + 120) << 23) | (m << 20); std::memcpy(&f, &u, 4); }
    uint32_t u; std::memcpy(&u, &f, 4); u |= sign; std::memcpy(&f, &u, 4);
    return f;
}
// per-row absmax scaling: scale = amax / 448 (1 when the row is all zero); w -> e4m3(w / scale) * scale
static inline void quant_rows_fp8(float* w, size_t rows, size_t cols) {
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long)rows; ++r) {
        float* row = w + (size_t)r * cols;
        float amax = 0.f;
        for (size_t c = 0; c < cols; ++c) amax = std::fmax(amax, std::fabs(row[c]));
        const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
        for (size_t c = 0; c < cols; ++c) row[c] = e4m3_to_f32(f32_to_e4m3(row[c] / scale)) * scale;
    }
}

// Fill `n` floats of tensor `name`.
static inline void fill(float* dst, size_t n, const std::string& name, uint64_t seed, float mean, double stdv,
                        bool bf16) {
    const uint64_t key = fnv1a64(name.c_str()) ^ seed;
    const float sc = scale_for(stdv);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        float v = elem(key, (uint64_t)i, mean, sc);
        dst[i] = bf16 ? round_bf16(v) : v;
    }
}

}  // namespace fsgen
