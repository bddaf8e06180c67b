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
