// k_grouped16_mfma -- grouped 3x3 convolution (group width 16) + BN (+ReLU) over packed pixel rows on the matrix cores
// (LAD-RegNet conv b, laud_regnet.py:184-189 on the kept rows; gfx950, bf16x3 arithmetic):
//     out[r, 16 g + n] = act(scale * sum_{t < 9} sum_{i < 16} a[nbr[r, t], 16 g + i] * w[16 g + n, t, i] + shift)
// Per group it is a GEMM with M = 16 output channels, N = rows, K = 9 taps x 16 channels: v_mfma_f32_16x16x32_bf16 takes 16 output
// channels x 16 rows x (two taps x 16 channels) per instruction, so a K step is a PAIR of taps and a lane's eight k-values are eight
// consecutive channels of one neighbour row (two 16-byte loads; the im2col is a per-lane address through the neighbour table).
// The VALU kernel it replaces (k_grouped3x3_lds: one dependent load per tap, 144 LDS weight reads per output quad) ran at ~30 % of
// the fp32 VALU peak and was 32 % of the RegNet layer-skip step.
// One 512-thread workgroup = a chunk of <= 12 groups (their pre-split weight fragments sit in LDS for the workgroup's lifetime)
// x a strided set of 16-row tiles, one tile per wave at a time; the loads of the next group fly during the MFMAs of the current one.
#include "ldn_common.h"

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GM_STEPS = 5;                 // nine taps in pairs: the second half of step 4 has zero weights
constexpr int GM_FRAG = GM_STEPS * 64 * 32; // bytes of one group's fragments: [5][64 lanes][8 hi | 8 lo] bf16
constexpr int GM_MAXG = 12;                 // groups per workgroup (120 KB of LDS)

struct GmArgs {
    const float* a; int lda;
    const int32_t* nbr; const int32_t* m_count; int m_cap;
    const unsigned char* wf;                // [C / 16][5][64][32 B]
    int C;
    const float* scale; const float* shift; int relu;
    float* out; int ldo;
    int gchunk;                             // groups per workgroup
};

__device__ __attribute__((aligned(16))) float g_gm_zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

__global__ __launch_bounds__(512, 2) void k_grouped16_mfma(const GmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = p.C / 16;
    const int g0 = blockIdx.y * p.gchunk;
    const int ng = min(p.gchunk, G - g0);
    const int M = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap;
    for (int i = tid; i < ng * (GM_FRAG / 16); i += 512)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(p.wf + (size_t)g0 * GM_FRAG)[i];
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;            // B operand / output column = row n of the tile; k-group kg of a step
    const int ntiles = (M + 15) / 16;
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
        const int r = tile * 16 + n;
        // this lane's neighbour row per K step: tap 2 s + (kg >> 1), channels 8 (kg & 1) .. + 7 of the group
        long aoff[GM_STEPS];
#pragma unroll
        for (int s = 0; s < GM_STEPS; ++s) {
            const int t = 2 * s + (kg >> 1);
            const int ar = (t < 9 && r < M) ? p.nbr[(size_t)r * 9 + t] : -1;
            aoff[s] = ar >= 0 ? (long)ar * p.lda + 8 * (kg & 1) : -1;
        }
        f32x4 x0[2][GM_STEPS], x1[2][GM_STEPS];
        auto request = [&](int gl, int buf) {
#pragma unroll
            for (int s = 0; s < GM_STEPS; ++s) {
                const float* src = aoff[s] >= 0 ? p.a + aoff[s] + (g0 + gl) * 16 : g_gm_zero;
                x0[buf][s] = *reinterpret_cast<const f32x4*>(src);
                x1[buf][s] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        };
        request(0, 0);
#pragma unroll 1
        for (int gl = 0; gl < ng; gl += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                // (two groups per trip so that the double buffer index is a constant)
                const int gc = gl + u;
                if (gc >= ng) break;
                if (gc + 1 < ng) request(gc + 1, (u + 1) & 1);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const unsigned char* wfr = smem + (size_t)gc * GM_FRAG + lane * 32;
#pragma unroll
                for (int s = 0; s < GM_STEPS; ++s) {
                    bf16x8 bh, bl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = e < 4 ? x0[u][s][e] : x1[u][s][e - 4];
                        const __bf16 hb = (__bf16)v;
                        bh[e] = hb;
                        bl[e] = (__bf16)(v - (float)hb);
                    }
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wfr + s * 64 * 32);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(wfr + s * 64 * 32 + 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                }
                // C layout: lane (column n = row of the tile, kg) holds output channels 4 kg .. 4 kg + 3 of the group
                const int c = (g0 + gc) * 16 + 4 * kg;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c), sh = *reinterpret_cast<const f32x4*>(p.shift + c);
                f32x4 v = acc * sc + sh;
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (r < M) *reinterpret_cast<f32x4*>(p.out + (size_t)r * p.ldo + c) = v;
            }
        }
    }
}

}  // namespace ldn

using namespace ldn;

extern "C" size_t ldn_grouped16_weight_bytes(int C) { return C > 0 && C % 16 == 0 ? (size_t)(C / 16) * GM_FRAG : 0; }

extern "C" int ldn_grouped16_conv3x3_rows(const float* a, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap,
                                          const void* w_frag, int C, const float* scale, const float* shift, int relu, float* out,
                                          int ldo, void* stream) {
    LDN_REQUIRE(a && nbr && w_frag && scale && shift && out, "ldn_grouped16_conv3x3_rows: null pointer");
    LDN_REQUIRE(C > 0 && C % 16 == 0, "ldn_grouped16_conv3x3_rows: channels must be a multiple of the group width 16 (got %d)", C);
    LDN_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && lda >= C && ldo >= C, "ldn_grouped16_conv3x3_rows: strides must be multiples of 4");
    LDN_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)w_frag % 16 == 0 && (uintptr_t)scale % 16 == 0 &&
                (uintptr_t)shift % 16 == 0, "ldn_grouped16_conv3x3_rows: pointers must be 16-byte aligned");
    if (m_cap <= 0) return LDN_OK;
    GmArgs g{};
    g.a = a; g.lda = lda; g.nbr = nbr; g.m_count = m_count; g.m_cap = m_cap; g.wf = static_cast<const unsigned char*>(w_frag); g.C = C;
    g.scale = scale; g.shift = shift; g.relu = relu; g.out = out; g.ldo = ldo;
    const int G = C / 16;
    const int nchunks = ceil_div(G, GM_MAXG);
    g.gchunk = ceil_div(G, nchunks);
    const size_t lds = (size_t)g.gchunk * GM_FRAG;
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_grouped16_mfma), lds), "k_grouped16_mfma: cannot reserve %zu B of LDS", lds);
    const int tiles = ceil_div(m_cap, 16);
    int bx = ceil_div(tiles, 8);
    int cus = 256;
    (void)ldn_device_cus(&cus);
    const int cap = max(1, (cus * (lds <= 80 * 1024 ? 2 : 1) * 4) / nchunks);      // a few waves of workgroups: the fragments are staged per workgroup
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(k_grouped16_mfma, dim3((unsigned)bx, (unsigned)nchunks), dim3(512), lds, static_cast<hipStream_t>(stream), g);
    LDN_CHECK_LAUNCH("k_grouped16_mfma");
    return LDN_OK;
}
