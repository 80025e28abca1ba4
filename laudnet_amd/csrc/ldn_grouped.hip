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


// ------------------------------------------------------------------------------------------------------------------------
// k_grouped16_img -- the same convolution when the packed rows are WHOLE IMAGES (layer skip: image k of the kept ones owns the
// input rows [k Hi Wi, (k + 1) Hi Wi) and the output rows [k Ho Wo, (k + 1) Ho Wo)).  The neighbour-table kernel above reads every
// input row nine times (once per tap) through the L2 -- on the 14 x 14 maps of RegNetY-800MF's stage 3 that is what bounds it
// (59 us for 64 MB in + 64 MB out).  Here a workgroup = (kept image, chunk of groups): the image's channels of the chunk are
// staged in LDS ONCE ([pixel][16 ng + 4] floats, + one zero pixel for the taps outside the image) and the im2col is a per-lane
// LDS address computed from the geometry, as in k_tail.  Work items = (16-pixel tile, group), dealt to the 8 waves.
struct GiArgs {
    const float* a; int lda;
    const int32_t* m_count;                 // device-side number of OUTPUT rows (kept images x Ho Wo)
    const unsigned char* wf; int C;
    const float* scale; const float* shift; int relu;
    float* out; int ldo;
    int Hi, Wi, Ho, Wo, stride;
    int gchunk, in_ld;                      // groups per workgroup; floats per staged pixel (16 gchunk + 4)
    int R, nbands;                          // output rows per workgroup (Ho: the whole image), bands per image
    int images_cap;                         // images the launch covers (workgroups = (image, band) pairs x group chunks, see the kernel)
    float* gap;                             // optional [kept image][band][C]: channel sums of this launch's FINAL output over the band's pixels
                                            // (the squeeze of the SE block that follows conv b, laud_regnet.py:194: no second pass over h_b)
};

// sum over the 16 lanes of a DPP row (inline asm: hipcc 7.2 merges neighbouring update_dpp calls, see dpp_swap_pair); fixed order
__device__ __forceinline__ float row16_sum(float v) {
    float t;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(t) : "v"(v));
    return t;
}

__global__ __launch_bounds__(512, 2) void k_grouped16_img(const GiArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = p.C / 16;
    // Workgroup order (round 6): the group chunks of ONE (image, band) pair take linear ids L, L + 8, L + 16, ... -- consecutive workgroups of ONE XCD
    // (block b runs on XCD b % 8).  With one group per workgroup (the banded form: 64 bytes of every 128-byte line of a staged pixel) the chunks of a
    // pair were nbands ids apart, on unrelated XCDs, and every line came over the fabric once per chunk: 926 MB fetched for 394 MB of input rows on
    // the first block of stage 1 (profiles/r06_pmc_all_regnet_before.txt), 463 MB in this order (r06_pmc_all_regnet.txt).  The launch's time did not
    // move (191 -> 198 us): it is not bound by the fabric; the bytes are simply no longer wasted.
    const int nch = (G + p.gchunk - 1) / p.gchunk;
    const int per = 8 * nch, jid = (int)(blockIdx.x % (unsigned)per);
    const long pair = (long)(blockIdx.x / (unsigned)per) * 8 + (jid & 7);
    const int k = (int)(pair / p.nbands), band = (int)(pair - (long)k * p.nbands);
    const int g0 = (jid >> 3) * p.gchunk;
    const int ng = min(p.gchunk, G - g0);
    const int HWo = p.Ho * p.Wo;
    if (k >= p.images_cap || (long)k * HWo >= (long)p.m_count[0]) return;   // beyond the launch's pairs / beyond the kept images
    // this workgroup's band of output rows and the input rows it reads (pad 1: one halo row on each side, inside the image)
    const int y0 = band * p.R, rows_out = min(p.R, p.Ho - y0);
    const int iy0 = max(y0 * p.stride - 1, 0), iy1 = min((y0 + rows_out - 1) * p.stride + 1, p.Hi - 1);
    const int HWi = (iy1 - iy0 + 1) * p.Wi;                                // staged pixels
    const size_t in_row0 = (size_t)k * p.Hi * p.Wi + (size_t)iy0 * p.Wi;   // first staged input row (flat)
    float* const s_in = reinterpret_cast<float*>(smem + (size_t)p.gchunk * GM_FRAG);
    for (int i = tid; i < ng * (GM_FRAG / 16); i += 512)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(p.wf + (size_t)g0 * GM_FRAG)[i];
    const int qpp = ng * 4;                                                // 16-byte pieces per pixel
    // (round 6) eight pieces per thread in flight: one piece at a time the staging was a chain of up to ten memory latencies per workgroup -- what the
    // banded launches of the first blocks spent part of their time in (k_grouped16_img on RegNetY-800MF's first block, 63 KB staged per workgroup:
    // 198 -> 172 us; the rest is 18 rounds of 9 us workgroups with ~1 us of matrix work each -- launch, zero-fill, barrier)
    const int npiece = HWi * qpp;
    const float inv_qpp = 1.f / (float)qpp;
    for (int i0 = tid; i0 < npiece; i0 += 8 * 512) {
        f32x4 v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * 512, npiece - 1);
            int px = (int)(((float)i + 0.5f) * inv_qpp);                   // i / qpp through the reciprocal, corrected (i < 2^22)
            int q = i - px * qpp;
            if (q < 0) { --px; q += qpp; } else if (q >= qpp) { ++px; q -= qpp; }
            dst[u] = px * p.in_ld + q * 4;
            v[u] = *reinterpret_cast<const f32x4*>(p.a + (in_row0 + px) * p.lda + g0 * 16 + q * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 512 < npiece) *reinterpret_cast<f32x4*>(s_in + dst[u]) = v[u];
    }
    for (int i = tid; i < p.in_ld; i += 512) s_in[(size_t)HWi * p.in_ld + i] = 0.f;   // the zero pixel
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;
    const int npix = rows_out * p.Wo;                                      // output pixels of the band
    const int ntile = (npix + 15) / 16;
    float* const s_gap = s_in + ((size_t)HWi + 1) * p.in_ld;              // (gap) [group][tile][16 channels]: per-tile channel sums
    for (int w = wave; w < ntile * ng; w += 8) {
        const int gl = w / ntile, tile = w - gl * ntile;
        const int q = tile * 16 + n;
        const bool valid = q < npix;
        const int ly = (valid ? q : 0) / p.Wo, ox = (valid ? q : 0) - ly * p.Wo;
        const int oy = y0 + ly;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* wfr = smem + (size_t)gl * GM_FRAG + lane * 32;
#pragma unroll
        for (int s = 0; s < GM_STEPS; ++s) {
            const int t = 2 * s + (kg >> 1);
            const int iy = oy * p.stride - 1 + t / 3, ix = ox * p.stride - 1 + t % 3;
            const bool inb = valid && t < 9 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            const float* src = s_in + (size_t)(inb ? (iy - iy0) * p.Wi + ix : HWi) * p.in_ld + gl * 16 + 8 * (kg & 1);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
            bf16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = e < 4 ? x0[e] : x1[e - 4];
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
        const int c = (g0 + gl) * 16 + 4 * kg;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c), sh = *reinterpret_cast<const f32x4*>(p.shift + c);
        f32x4 v = acc * sc + sh;
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (valid) *reinterpret_cast<f32x4*>(p.out + ((size_t)k * HWo + (size_t)y0 * p.Wo + q) * p.ldo + c) = v;
        if (p.gap) {      // (uniform) channel sums of the tile: a fixed tree over its 16 pixels, tiles added in order below
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = row16_sum(valid ? v[e] : 0.f);
            if (n == 0) *reinterpret_cast<f32x4*>(s_gap + (size_t)w * 16 + 4 * kg) = t;
        }
    }
    if (p.gap) {
        __syncthreads();
        if (tid < ng * 16) {
            const int gl = tid >> 4, ch = tid & 15;
            float s = 0.f;
            for (int t = 0; t < ntile; ++t) s += s_gap[((size_t)gl * ntile + t) * 16 + ch];
            p.gap[((size_t)k * p.nbands + band) * p.C + (g0 + gl) * 16 + ch] = s;
        }
    }
}

// groups per workgroup and output rows per workgroup for an Hi x Wi input map (false: not even three input rows of one group fit)
// gap: the launch also leaves channel sums -- [group][16-pixel tile][16] floats of scratch behind the staged rows count against the same budgets
// (they did not, and a map near the whole-image limit made the gap form's launch fail instead of answering "does not fit")
static bool gi_plan(int Hi, int Wi, int Ho, int stride, int G, int* ng_out, int* R_out, bool gap = false) {
    const int Wo = (Wi - 1) / stride + 1;
    auto bytes = [&](int ng, int in_rows, int out_rows) {
        return (size_t)ng * GM_FRAG + ((size_t)in_rows * Wi + 1) * (16 * ng + 4) * 4 + (gap ? (size_t)ng * ceil_div(out_rows * Wo, 16) * 64 : 0);
    };
    if (bytes(1, Hi, Ho) <= 150 * 1024) {          // whole images: as many groups per workgroup as leave room for two workgroups per CU
        int ng = 1;
        while (ng < G && ng < GM_MAXG && bytes(ng + 1, Hi, Ho) <= 80 * 1024) ++ng;
        const int nchunks = ceil_div(G, ng);
        *ng_out = ceil_div(G, nchunks);
        *R_out = Ho;
        return true;
    }
    int R = 0;                                 // bands of output rows, one group per workgroup
    while (R < Ho && bytes(1, R * stride + 3, R + 1) <= 80 * 1024) ++R;          // (R + 1 - 1) stride + 3 input rows for R + 1 output rows
    if (R < 1) return false;
    const int nb = ceil_div(Ho, R);
    *ng_out = 1;
    *R_out = ceil_div(Ho, nb);
    return true;
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

extern "C" int ldn_grouped16_images_fit(int Hi, int Wi, int C) {
    int ng = 0, R = 0;   // (stride 2 is the worst case for the banded form: three input rows for one output row)
    // (with the channel-sum scratch of the gap form: the module gates its fused SE path on this answer)
    return (Hi > 0 && Wi > 0 && C > 0 && C % 16 == 0 && gi_plan(Hi, Wi, (Hi - 1) / 2 + 1, 2, C / 16, &ng, &R, true) &&
            gi_plan(Hi, Wi, Hi, 1, C / 16, &ng, &R, true)) ? ng : 0;
}

static int grouped16_images(const float* a, int lda, const int32_t* m_count, int images_cap, int Hi, int Wi, int Ho,
                            int Wo, int stride, const void* w_frag, int C, const float* scale, const float* shift,
                            int relu, float* out, int ldo, float* gap, void* stream) {
    LDN_REQUIRE(a && m_count && w_frag && scale && shift && out, "ldn_grouped16_conv3x3_images: null pointer");
    LDN_REQUIRE(C > 0 && C % 16 == 0, "ldn_grouped16_conv3x3_images: channels must be a multiple of the group width 16 (got %d)", C);
    LDN_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && lda >= C && ldo >= C, "ldn_grouped16_conv3x3_images: strides must be multiples of 4");
    LDN_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)w_frag % 16 == 0 && (uintptr_t)scale % 16 == 0 &&
                (uintptr_t)shift % 16 == 0, "ldn_grouped16_conv3x3_images: pointers must be 16-byte aligned");
    LDN_REQUIRE(stride >= 1 && Hi > 0 && Wi > 0 && Ho == (Hi - 1) / stride + 1 && Wo == (Wi - 1) / stride + 1,
                "ldn_grouped16_conv3x3_images: %dx%d -> %dx%d is not a 3x3 / pad 1 / stride %d geometry", Hi, Wi, Ho, Wo, stride);
    if (images_cap <= 0) return LDN_OK;
    GiArgs g{};
    g.a = a; g.lda = lda; g.m_count = m_count; g.wf = static_cast<const unsigned char*>(w_frag); g.C = C;
    g.scale = scale; g.shift = shift; g.relu = relu; g.out = out; g.ldo = ldo;
    g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.gap = gap;
    const int G = C / 16;
    LDN_REQUIRE(gi_plan(Hi, Wi, Ho, stride, G, &g.gchunk, &g.R, gap != nullptr),
                "ldn_grouped16_conv3x3_images: three rows of a %d-wide map do not fit the LDS (ldn_grouped16_images_fit)", Wi);
    g.nbands = ceil_div(Ho, g.R);
    g.in_ld = 16 * g.gchunk + 4;
    const int in_rows = g.R == Ho ? Hi : min(Hi, (g.R - 1) * stride + 3);
    const size_t lds = (size_t)g.gchunk * GM_FRAG + ((size_t)in_rows * Wi + 1) * g.in_ld * 4 +
                       (gap ? (size_t)g.gchunk * ceil_div(g.R * Wo, 16) * 64 : 0);        // (gap: per-tile channel sums)
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_grouped16_img), lds), "k_grouped16_img: cannot reserve %zu B of LDS", lds);
    g.images_cap = images_cap;
    const long pairs = (long)images_cap * g.nbands;
    const long nwg = (pairs + 7) / 8 * 8 * ceil_div(G, g.gchunk);
    LDN_REQUIRE(nwg < (1l << 31), "ldn_grouped16_conv3x3_images: too many workgroups (%ld)", nwg);
    hipLaunchKernelGGL(k_grouped16_img, dim3((unsigned)nwg), dim3(512), lds, static_cast<hipStream_t>(stream), g);
    LDN_CHECK_LAUNCH("k_grouped16_img");
    return LDN_OK;
}

extern "C" int ldn_grouped16_conv3x3_images(const float* a, int lda, const int32_t* m_count, int images_cap, int Hi, int Wi, int Ho,
                                            int Wo, int stride, const void* w_frag, int C, const float* scale, const float* shift,
                                            int relu, float* out, int ldo, void* stream) {
    return grouped16_images(a, lda, m_count, images_cap, Hi, Wi, Ho, Wo, stride, w_frag, C, scale, shift, relu, out, ldo, nullptr, stream);
}

extern "C" int ldn_grouped16_images_bands(int Hi, int Wi, int Ho, int stride, int C) {
    int ng = 0, R = 0;
    if (!(Hi > 0 && Wi > 0 && Ho > 0 && stride >= 1 && C > 0 && C % 16 == 0 && gi_plan(Hi, Wi, Ho, stride, C / 16, &ng, &R, true))) return 0;      // (the gap form's bands: it sizes that form's buffer)
    return ceil_div(Ho, R);
}

extern "C" int ldn_grouped16_conv3x3_images_gap(const float* a, int lda, const int32_t* m_count, int images_cap, int Hi, int Wi, int Ho,
                                                int Wo, int stride, const void* w_frag, int C, const float* scale, const float* shift,
                                                int relu, float* out, int ldo, float* gap_partial, void* stream) {
    LDN_REQUIRE(gap_partial, "ldn_grouped16_conv3x3_images_gap: null pointer");
    return grouped16_images(a, lda, m_count, images_cap, Hi, Wi, Ho, Wo, stride, w_frag, C, scale, shift, relu, out, ldo, gap_partial, stream);
}
