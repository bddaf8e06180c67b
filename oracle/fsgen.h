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
// cross-checked by tests/test_synth.py.  The oracle never ships in the product path.
#pragma once
#include <cstdint>
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
