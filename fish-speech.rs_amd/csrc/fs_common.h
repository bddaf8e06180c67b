// Shared host-side helpers of libfishrt: error plumbing + HIP checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace fs {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define FS_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            throw fs::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                            std::to_string(__LINE__) + ")");                                           \
    } while (0)

#define FS_REQUIRE(cond, msg)                     \
    do {                                          \
        if (!(cond)) throw fs::Error(std::string(msg)); \
    } while (0)

typedef uint16_t bf16_t;  // raw bf16 storage
struct fp8_t { uint8_t v; };  // raw OCP e4m3fn storage (gfx950's FP8 format; NOT the MI300 fnuz variant)

// f32 -> e4m3fn, round-to-nearest-even, saturating to +-448 (no inf; NaN -> 0x7F).  Pure integer/float arithmetic so the
// host, the device quantiser and the test oracle produce identical bytes.
__host__ __device__ inline uint8_t f32_to_e4m3(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint8_t)(sign | 0x7F);  // NaN
    float af;
    __builtin_memcpy(&af, &a, 4);
    if (af >= 464.0f) return (uint8_t)(sign | 0x7E);      // >= halfway between 448 and 480 saturates to 448 (0x7E)
    if (af < 0.0009765625f) return sign;                  // < 2^-10: rounds to zero (half of the smallest subnormal 2^-9)
    int e = (int)(a >> 23) - 127;
    if (e < -6) {                                         // subnormal: multiples of 2^-9
        const float q = af * 512.0f;                      // exact scaling
        float r = __builtin_rintf(q);                     // RNE (default rounding mode)
        uint32_t m = (uint32_t)r;                         // 0..8 (8 => smallest normal)
        return (uint8_t)(sign | m);
    }
    // normal: 3 mantissa bits
    uint32_t mant = a & 0x7FFFFFu;
    uint32_t keep = mant >> 20, rest = mant & 0xFFFFFu;
    if (rest > 0x80000u || (rest == 0x80000u && (keep & 1u))) keep += 1;
    if (keep == 8) { keep = 0; e += 1; }
    if (e > 8) return (uint8_t)(sign | 0x7E);
    uint8_t out = (uint8_t)(sign | ((uint32_t)(e + 7) << 3) | keep);
    if ((out & 0x7F) == 0x7F) out = (uint8_t)(sign | 0x7E);  // 0x7F is NaN in e4m3fn: saturate
    return out;
}
__host__ __device__ inline float e4m3_to_f32(uint8_t b) {
    const uint32_t sign = (uint32_t)(b & 0x80) << 24, e = (b >> 3) & 0xF, m = b & 7;
    float f;
    if (e == 0) f = (float)m * 0.001953125f;                               // m * 2^-9
    else if (e == 15 && m == 7) { const uint32_t n = 0x7FC00000u; __builtin_memcpy(&f, &n, 4); return f; }
    else { const uint32_t u = ((e + 120) << 23) | (m << 20); __builtin_memcpy(&f, &u, 4); }
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u |= sign;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(r >> 16);
}
static inline float bf16_to_f32_host(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace fs
