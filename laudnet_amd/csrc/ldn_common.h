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
int tu_chain_stalls(unsigned* count, int reset);      // csrc/ldn_tail.hip
// The process's fault word: pinned host memory that a kernel sets when one of its bounded waits runs into its bound (ldn_fault_flag; csrc/ldn_index.hip).
// fault_word_dev(): its device-side address (nullptr if the allocation failed).
int* fault_word_host();
int* fault_word_dev();

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

// One lane's 16 bytes of a PRE-SPLIT octet ([8 hi | 8 lo] bf16): lanes 2q / 2q + 1 hold channels 0-3 / 4-7 of the octet as fp32 quads.
// Each lane splits its OWN quad (hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even: the in-loop split of the un-split kernels),
// then the pair exchanges what the other one stores: the even lane stores the octet's 8 hi (its own 4 + the partner's), the odd lane the
// 8 lo (the partner's 4 + its own).  Two DPP moves of packed pairs per lane.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t presplit_store_quad(const f32x4 x, bool odd) {
#ifdef LDN_OF_NOP     // tuning only (wrong values): what the conversion itself costs
    return __builtin_bit_cast(u32x4_t, x);
#endif
    unsigned hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const __bf16 h0 = (__bf16)x[2 * q], h1 = (__bf16)x[2 * q + 1];
        const bf16x2_t hp = {h0, h1};
        const bf16x2_t lp = {(__bf16)(x[2 * q] - (float)h0), (__bf16)(x[2 * q + 1] - (float)h1)};
        hi[q] = __builtin_bit_cast(unsigned, hp);
        lo[q] = __builtin_bit_cast(unsigned, lp);
    }
    const float r0 = dpp_swap_pair(__builtin_bit_cast(float, odd ? hi[0] : lo[0]));     // even sends its lo, odd sends its hi
    const float r1 = dpp_swap_pair(__builtin_bit_cast(float, odd ? hi[1] : lo[1]));
    const unsigned p0 = __builtin_bit_cast(unsigned, r0), p1 = __builtin_bit_cast(unsigned, r1);
    return odd ? u32x4_t{p0, p1, lo[0], lo[1]} : u32x4_t{hi[0], hi[1], p0, p1};
}

// x[l] + x[l ^ 8] + x[l ^ 16] + ... : the sum over lane bits 3, 4, 5 (the eight row groups of a (row = lane >> 3, quad = lane & 7) epilogue layout), in every
// lane, WITHOUT the LDS: the tree of three __shfl_xor steps (8, 16, 32 = three dependent ds_bpermute round trips) as one DPP add inside the 16-lane row
// and the two gfx950 half / row exchanges.  Operand for operand the same additions (a + b vs b + a): bit-identical to the shuffle form.
__device__ __forceinline__ float sum_lane_bits_345(float x) {
    float t;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=&v"(t) : "v"(x));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// 16-byte global store, optionally WRITE-THROUGH (sc0 sc1: the line leaves the XCD's L2 at once instead of staying dirty until the
// end-of-kernel write-back).  LDN_WT_STORES is a tuning switch (the ~5.6 us behind every large row kernel).
template <typename V>
__device__ __forceinline__ void store16(void* ptr, const V v) {
#ifdef LDN_WT_STORES
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(ptr), "v"(v) : "memory");
#else
    *reinterpret_cast<V*>(ptr) = v;
#endif
}

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kXcds = 8;    // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ldn
