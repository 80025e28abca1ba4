// Internal helpers shared by the translation units of libldn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ldn_hip.h"

namespace ldn {

void set_error(const char* fmt, ...);
bool allow_dynamic_lds(const void* kernel, size_t bytes);

#define LDN_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            ::ldn::set_error(__VA_ARGS__);      \
            return LDN_EINVAL;                  \
        }                                       \
    } while (0)

#define LDN_CHECK_LAUNCH(what)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            ::ldn::set_error("%s: %s", what, hipGetErrorString(e__));                   \
            return LDN_EHIP;                                                            \
        }                                                                               \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kXcds = 8;    // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ldn
