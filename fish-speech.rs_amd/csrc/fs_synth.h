// Deterministic synthetic-weight generator of the product library (device + host).
//
// No Fish-Speech checkpoint exists offline ("no model weights ... anywhere on disk"), so
// fs_lm_load_synthetic / fs_codec_load_synthetic fill tensors, addressed by the reference loader's tensor
// names (fish_speech_core/lib/lm/dual_ar.rs:125-156,219-223,415-419,466-511), from a counter-based hash that
// is bit-reproducible on any IEEE-754 machine (integer hash -> Irwin-Hall(4) -> one f32 multiply, one f32 add;
// no libm):
//     h   = mix(fnv1a64(name) ^ seed  +  (i+1) * 0x9E3779B97F4A7C15)
//     s   = sum of the four 16-bit fields of h  - 131070
//     val = mean + (float)s * (float)(std / 37837.2272)
// tests/test_lm_gpu.py::test_synthetic_weights_match_oracle_spec and tests/test_safetensors_gpu.py cross-check this implementation against
// the test oracle's independent copy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fs {

__host__ __device__ inline uint64_t synth_fnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) {
        h ^= (uint8_t)*s;
        h *= 0x100000001B3ull;
    }
    return h;
}
__host__ __device__ inline uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline float synth_elem(uint64_t key, uint64_t i, float mean, float scale) {
    uint64_t h = synth_mix(key + (i + 1) * 0x9E3779B97F4A7C15ull);
    int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)(h >> 48) - 131070;
    return __fadd_rn(mean, __fmul_rn((float)s, scale));  // no FMA contraction: two roundings, like the spec
}
inline float synth_scale(double stdv) { return (float)(stdv / 37837.2272); }

}  // namespace fs
