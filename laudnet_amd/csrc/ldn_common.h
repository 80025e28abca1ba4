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

// ---- LDN_DEBUG build (python -m laudnet_amd.build --debug -> libldn_hip_debug.so): device-side checks of every index list a
// kernel consumes (bounds, alignment of channel pairs, counts).  A violation never traps: it is COUNTED (per translation unit)
// and the first one's code is kept; ldn_debug_violations() sums the counters.  Release builds compile the checks away.
#ifdef LDN_DEBUG
static __device__ unsigned g_ldn_viol[2] = {0u, 0u};   // {count, code of the first violation}
#define LDN_DCHECK(cond, code)                                                   \
    do {                                                                         \
        if (!(cond)) {                                                           \
            if (atomicAdd(&g_ldn_viol[0], 1u) == 0u) g_ldn_viol[1] = (code);     \
        }                                                                        \
    } while (0)
#define LDN_DEFINE_TU_VIOLATIONS(fn)                                                                  \
    int fn(unsigned* count, unsigned* code, int reset) {                                              \
        unsigned v[2] = {0u, 0u};                                                                     \
        if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_ldn_viol), sizeof(v)) != hipSuccess) return LDN_EHIP; \
        *count += v[0];                                                                               \
        if (v[0] && !*code) *code = v[1];                                                             \
        if (reset) {                                                                                  \
            unsigned z[2] = {0u, 0u};                                                                 \
            if (hipMemcpyToSymbol(HIP_SYMBOL(g_ldn_viol), z, sizeof(z)) != hipSuccess) return LDN_EHIP; \
        }                                                                                             \
        return LDN_OK;                                                                                \
    }
#else
#define LDN_DCHECK(cond, code) do { } while (0)
#define LDN_DEFINE_TU_VIOLATIONS(fn) int fn(unsigned*, unsigned*, int) { return LDN_OK; }
#endif
int tu_violations_conv(unsigned* count, unsigned* code, int reset);
int tu_violations_index(unsigned* count, unsigned* code, int reset);
int tu_violations_regnet(unsigned* count, unsigned* code, int reset);
int tu_violations_tail(unsigned* count, unsigned* code, int reset);
int tu_violations_dense(unsigned* count, unsigned* code, int reset);
int tu_violations_small(unsigned* count, unsigned* code, int reset);
int tu_violations_rows3(unsigned* count, unsigned* code, int reset);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// lane ^ 1 exchange (DPP quad_perm [1,0,3,2]).  Inline asm on purpose: hipcc 7.2 merges several __builtin_amdgcn_update_dpp calls
// of one unrolled loop into ONE v_mov_b32_dpp of the first operand and uses its result for all of them (seen in the pre-split
// epilogues of round 5: every lane got its partner's element 0 four times).  The s_nop covers the VALU-write -> DPP-read wait states.
__device__ __forceinline__ float dpp_swap_pair(float v) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kXcds = 8;    // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ldn
