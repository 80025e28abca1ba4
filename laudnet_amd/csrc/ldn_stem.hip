// k_stem -- the static stem of the LAUD-ResNets as ONE kernel (gfx950 / CDNA4, bf16x3 arithmetic):
//
//     conv 7x7 stride 2 pad 3 (3 -> C channels, bn1's scale folded into the weights)  ->  max-pool 3x3 stride 2 pad 1
//     ->  + bn1's shift  ->  ReLU                                  (imagenet_classification/models/laud_resnet.py:316-326)
//
// relu(maxpool(s * conv + t)) == relu(maxpool(s * conv) + t): adding a per-channel constant and the ReLU are monotone and the
// pool's padding is -inf, so the shift and the ReLU are applied to the POOLED map.  The full-resolution conv output (822 MB at
// bs256 / 224^2, written once and read once by the library's conv -> max-pool pair) never exists in memory: the kernel reads
// the image (154 MB) and writes the pooled map (205 MB).
//
// One persistent 512-thread workgroup per CU walks tiles of 8 x 7 pooled pixels = 17 x 15 conv pixels (255: one 32-pixel MFMA
// column tile per wave; the conv rows / columns shared with the neighbouring tiles are recomputed, 1.14x):
//   * the 39 x 35 x 3 input patch of the tile is fetched into registers while the previous tile is computed and then written to
//     LDS as raw fp32 (zero outside the image);
//   * transposed MFMA formulation (A operand = weights, B operand = activations, lane = conv pixel): K = (ky, kx, c) is laid out
//     as 7 rows of 24 (21 real + 3 zero-weight slots), so that the 8 k-values of a lane are 8 CONSECUTIVE floats of the patch
//     (an im2col that is a per-lane LDS address); they are split into bf16 hi / lo by the wave that owns the pixels; the weights
//     are pre-split once per module into MFMA fragment order (two ds_read_b128 per fragment, no VALU);
//   * the conv tile goes to LDS ([pixel][C] fp32, 16-byte slots XOR-swizzled with the pixel) and is pooled from there:
//     thread = (pooled pixel, 4 channels), nine 16-byte reads, 16-byte coalesced stores of the NHWC output.
#include "ldn_common.h"

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef LDN_STEM_ABLATE
#define LDN_STEM_ABLATE 0   // tuning only (results are wrong): 1 = no MFMA loop, 2 = no pooling phase, 4 = no patch fetch, 8 = no conv-tile write
#endif
constexpr int S_PH = 8, S_PW = 7;                     // pooled pixels per tile
constexpr int S_CH = 2 * S_PH + 1, S_CW = 2 * S_PW + 1;   // conv pixels per tile: 17 x 15 = 255
constexpr int S_IH = 2 * (S_CH - 1) + 7, S_IW = 2 * (S_CW - 1) + 7;   // input patch: 39 x 35
constexpr int S_ROWF = S_IW * 3;                      // values per patch row (105)
constexpr int S_ROWP = S_ROWF + 1;                    // ... padded to an even count: every k-octet of a lane then starts on a 4-byte boundary of the bf16 planes
constexpr int S_PATCH = (S_IH + 1) * S_ROWP + 32;     // + one row and a tail: the zero-weight k slots read past the window
constexpr int S_KSTEPS = 11;                          // 7 rows x 24 = 168 -> 11 steps of 16 (the last half step has zero weights)
constexpr int S_LOADS = (S_IH * S_ROWF + 511) / 512;  // patch floats per thread (8)

struct StemArgs {
    const float* x; int B, H, W;                      // NHWC fp32, 3 channels
    const unsigned char* wf;                          // [C / 32][11][64 lanes][8 hi | 8 lo] bf16
    const float* shift;                               // [C]
    float* out; int C, Hc, Wc, Hp, Wp;                // NHWC [B, Hp, Wp, C]
    int tiles_y, tiles_x, ntiles;
    float* gap;                                       // optional [B][tiles_y * tiles_x][C]: per-tile channel sums of out (the first block's channel masker)
};

template <int NSUB>   // C = 32 * NSUB
__global__ __launch_bounds__(512, 2) void k_stem(const StemArgs p) {
    constexpr int C = 32 * NSUB;
    constexpr int WF_BYTES = NSUB * S_KSTEPS * 64 * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const s_w = smem;                                           // weight fragments
    // the input patch PRE-SPLIT into two bf16 planes (hi = bf16(x), lo = bf16(x - hi)), [40][106] (+ tail) each: an input value takes
    // part in ~12 conv outputs, so splitting it once when the patch is staged instead of once per use removes ~300 VALU instructions
    // per lane and tile from the MFMA loop (round 4; the products and their order are unchanged: bit-identical results)
    __bf16* const s_ph = reinterpret_cast<__bf16*>(smem + WF_BYTES);
    __bf16* const s_pl = s_ph + round_up(S_PATCH, 8);
    float* const s_conv = reinterpret_cast<float*>(s_pl + round_up(S_PATCH, 8));   // [256][C]
    float* const s_gs = s_conv + 256 * C;                                      // [2][8 waves][C] channel sums of the tile's pooled pixels (gap only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    for (int i = tid; i < WF_BYTES / 16; i += 512)
        reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(p.wf)[i];
    for (int i = tid; i < 2 * round_up(S_PATCH, 8); i += 512) s_ph[i] = (__bf16)0.f;   // both planes: the pad column / row / tail stay zero

    // this lane's conv pixel of the tile and its patch offset (floats)
    const int pm = wave * 32 + l31;
    const int pmc = min(pm, S_CH * S_CW - 1);
    const int coy = pmc / S_CW, cox = pmc - coy * S_CW;
    const int pbase = 2 * coy * S_ROWP + 2 * cox * 3;

    auto tile_origin = [&](int t, int& b, int& py0, int& px0) {
        const int per_img = p.tiles_y * p.tiles_x;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        py0 = ty * S_PH;
        px0 = (r - ty * p.tiles_x) * S_PW;
    };
    float pre[S_LOADS];
    auto fetch_one = [&](int b, int iy0, int ix0, int i) {      // patch value i of this thread for the tile whose window starts at (iy0, ix0)
        const int e = i * 512 + tid;
        const int r = e / S_ROWF, cc = e - r * S_ROWF;
        const int iy = iy0 + r, ix = ix0 + cc / 3;
        const bool ok = e < S_IH * S_ROWF && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const long off = (((long)b * p.H + iy) * p.W + ix0) * 3 + cc;       // >= 0 whenever ok
#if LDN_STEM_ABLATE & 4
        pre[i] = 0.f;
#else
        pre[i] = ok ? p.x[off] : 0.f;
#endif
    };
    auto fetch = [&](int t) {
        int b, py0, px0;
        tile_origin(t, b, py0, px0);
#pragma unroll
        for (int i = 0; i < S_LOADS; ++i) fetch_one(b, 4 * py0 - 5, 4 * px0 - 5, i);
    };
    // The pooled values of a tile are STORED during the MFMA loop of the next tile, and the patch of the next tile is FETCHED one value
    // per K step: issued in bursts between the phases, the 14 KB of stores and the 16 KB of loads of a tile cost the phase they sit in
    // their issue time (a CU retires ~8 B/clk of stores: ablations of round 4, 103 + 72 us of a 398 us launch) while the matrix pipe idles.
    constexpr int QUADS = C / 4;
    constexpr int PI = (S_PH * S_PW * QUADS + 511) / 512;      // pooled (pixel, channel quad) items per thread and tile
    static_assert(S_LOADS + PI <= S_KSTEPS, "one deferred load / store per K step");
    f32x4 pv[PI];
    float* pdst[PI];
#pragma unroll
    for (int k = 0; k < PI; ++k) pdst[k] = nullptr;

    // the per-wave channel sums of tile t are combined (in wave order) and stored behind the NEXT barrier the loop has anyway
    auto flush_gap = [&](int tprev, int slot) {
        if (tid < C) {
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += s_gs[(slot * 8 + w8) * C + tid];
            p.gap[(size_t)tprev * C + tid] = v;     // tile index t = b * tiles_per_image + r: exactly the [B][tiles][C] layout
        }
    };
    int t = blockIdx.x, it = 0;
    if (t < p.ntiles) fetch(t);
    for (; t < p.ntiles; t += gridDim.x, ++it) {
        __syncthreads();           // every wave has left the previous tile's patch and conv tile (and, first, the weights are staged)
        if (p.gap && it > 0) flush_gap(t - (int)gridDim.x, (it - 1) & 1);
#pragma unroll
        for (int i = 0; i < S_LOADS; ++i) {
            const int e = i * 512 + tid;
            if (e < S_IH * S_ROWF) {
                const int r = e / S_ROWF, o = e + r;           // row r, column e - r * S_ROWF of the padded planes
                const __bf16 hb = (__bf16)pre[i];
                s_ph[o] = hb;
                s_pl[o] = (__bf16)(pre[i] - (float)hb);
            }
        }
        __syncthreads();
        const bool have_next = t + (int)gridDim.x < p.ntiles;
        int nb = 0, npy0 = 0, npx0 = 0;
        if (have_next) tile_origin(t + gridDim.x, nb, npy0, npx0);

        f32x16 acc[NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // software-pipelined over the 11 K16 steps (schedule pinned): the 8 patch floats and the weight fragments of step s + 1 are
        // requested before the MFMAs of step s and split into bf16 hi / lo after them
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
        u32x4s rawh[2], rawl[2];
        bf16x8 fh[2][NSUB], fl[2][NSUB];
        auto request = [&](int s, int buf) {
            const int q = 2 * s + h;                   // this lane's k8 group: patch row q / 3, values 8 (q % 3) .. + 7 of the row window
            const int off = pbase + (q / 3) * S_ROWP + 8 * (q % 3);        // even: four aligned dwords per plane
            const unsigned* sh = reinterpret_cast<const unsigned*>(s_ph + off);
            const unsigned* sl = reinterpret_cast<const unsigned*>(s_pl + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) { rawh[buf][e] = sh[e]; rawl[buf][e] = sl[e]; }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                const unsigned char* wp = s_w + ((j * S_KSTEPS + s) * 64 + lane) * 32;
                fh[buf][j] = *reinterpret_cast<const bf16x8*>(wp);
                fl[buf][j] = *reinterpret_cast<const bf16x8*>(wp + 16);
            }
        };
        auto split = [&](int buf, bf16x8& bh, bf16x8& bl) {     // (nothing left to split: the planes hold the operands)
            bh = __builtin_bit_cast(bf16x8, rawh[buf]);
            bl = __builtin_bit_cast(bf16x8, rawl[buf]);
        };
        bf16x8 bh, bl;
        request(0, 0);
        split(0, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < ((LDN_STEM_ABLATE & 1) ? 1 : S_KSTEPS); ++s) {
            if (s + 1 < S_KSTEPS) request(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s & 1][j], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s & 1][j], bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s & 1][j], bh, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < S_KSTEPS) split((s + 1) & 1, bh, bl);
            if (s < S_LOADS) { if (have_next) fetch_one(nb, 4 * npy0 - 5, 4 * npx0 - 5, s); }
            else if (s - S_LOADS < PI) {
                if (pdst[s - S_LOADS]) __builtin_nontemporal_store(pv[s - S_LOADS], reinterpret_cast<f32x4*>(pdst[s - S_LOADS]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // C layout (lane = pixel l31, register r = channel (r & 3) + 8 (r >> 2) + 4 h of the subtile) -> s_conv[pixel][C]
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int q4 = 0; q4 < ((LDN_STEM_ABLATE & 8) ? 1 : 4); ++q4) {
                const f32x4 v = {acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
                const int slot = 8 * j + 2 * q4 + h;
                *reinterpret_cast<f32x4*>(s_conv + pm * C + ((slot ^ (pm & 7)) << 2)) = v;
            }
        __syncthreads();

        // ---- max-pool 3x3 stride 2 pad 1 over the conv tile, + shift, ReLU
        int b, py0, px0;
        tile_origin(t, b, py0, px0);
        f32x4 gs = {0.f, 0.f, 0.f, 0.f};           // this thread's channel quad (tid % QUADS: 512 % QUADS == 0) summed over its pooled pixels
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int w = tid + 512 * k;
            pdst[k] = nullptr;
            if ((LDN_STEM_ABLATE & 2) || w >= S_PH * S_PW * QUADS) continue;
            const int cq = w % QUADS, pp = w / QUADS;
            const int ly = pp / S_PW, lx = pp - ly * S_PW;
            const int py = py0 + ly, px = px0 + lx;
            if (py >= p.Hp || px >= p.Wp) continue;
            // branch-free: all nine 16-byte reads are issued together (the tile-local window always exists in s_conv); a tap outside the
            // conv map (the pool's -inf padding) is replaced by the centre value, which is always inside (max unchanged)
            f32x4 v9[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cy = 2 * py - 1 + dy, cx = 2 * px - 1 + dx;      // conv pixel (global); tile-local = (2 ly + dy, 2 lx + dx)
                    const bool ok = cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc;
                    const int cp = ok ? (2 * ly + dy) * S_CW + 2 * lx + dx : (2 * ly + 1) * S_CW + 2 * lx + 1;
                    v9[dy * 3 + dx] = *reinterpret_cast<const f32x4*>(s_conv + cp * C + ((cq ^ (cp & 7)) << 2));
                }
            f32x4 m = v9[4];
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v9[k][e]);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + cq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e] + sh[e], 0.f);
            pv[k] = m;                             // stored during the next tile's MFMA loop (or after the last tile)
            pdst[k] = p.out + (((size_t)b * p.Hp + py) * p.Wp + px) * C + cq * 4;
            gs += m;
        }
        if (p.gap) {   // lanes with the same lane % QUADS hold the same channel quad: fold them, one LDS row per wave
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = gs[e];
                if constexpr (QUADS <= 32) v += __shfl_xor(v, 32, 64);
                if constexpr (QUADS <= 16) v += __shfl_xor(v, 16, 64);
                if constexpr (QUADS <= 8) v += __shfl_xor(v, 8, 64);
                gs[e] = v;
            }
            if (lane < QUADS) *reinterpret_cast<f32x4*>(s_gs + (((it & 1) * 8 + wave) * C) + lane * 4) = gs;
        }
    }
#pragma unroll
    for (int k = 0; k < PI; ++k)
        if (pdst[k]) __builtin_nontemporal_store(pv[k], reinterpret_cast<f32x4*>(pdst[k]));
    if (p.gap && it > 0) {
        __syncthreads();
        flush_gap(t - (int)gridDim.x, (it - 1) & 1);
    }
}


template <int NSUB>
static int launch_stem(const StemArgs& a, int cus, hipStream_t st) {
    constexpr int C = 32 * NSUB;
    const size_t lds = (size_t)NSUB * S_KSTEPS * 64 * 32 + (size_t)round_up(S_PATCH, 8) * 4 + (size_t)256 * C * 4 + (size_t)2 * 8 * C * 4;
    LDN_REQUIRE(lds <= 160 * 1024, "ldn_stem_conv_pool: %zu B of LDS exceed 160 KiB", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_stem<NSUB>), lds), "k_stem: cannot reserve %zu B of LDS", lds);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int grid = min(a.ntiles, cus * per_cu);
    hipLaunchKernelGGL((k_stem<NSUB>), dim3((unsigned)grid), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_stem");
    return LDN_OK;
}


// ------------------------------------------------------------------------------------------------------------------------
// k_stem3 -- the static stem of the LAD-RegNets as one kernel: conv 3x3 stride 2 pad 1 (3 -> C channels, the BN's scale folded into
// the weights) -> + shift -> ReLU                     (imagenet_classification/models/laud_regnet.py:59-71, SimpleStemIN).
// The library ran it as a conv, a batch-norm pass and a ReLU pass over the 411 MB output (bs256 / 224^2); here the image is read
// once and the output written once.  Same transposed MFMA formulation as k_stem: K = (ky, kx, c) is laid out as 3 rows of 16
// (9 real + 7 zero-weight slots), one K16 step per kernel row, so that a lane's 8 k-values are 8 consecutive floats of the staged
// input row window.  One 512-thread workgroup = a band of R output rows of one image (R * Wo <= 256 pixels, wave w owns pixels
// 32 w .. 32 w + 31 of the band); the 2 R + 1 input rows are staged in LDS with a zero pixel left and right.
struct Stem3Args {
    const float* x; int B, H, W;                      // NHWC fp32, 3 channels
    const unsigned char* wf;                          // [C / 32][3][64 lanes][8 hi | 8 lo] bf16
    const float* shift; int relu;
    float* out; int C, Ho, Wo;                        // NHWC [B, Ho, Wo, C]
    int R, bands;                                     // output rows per workgroup, workgroups per image
    int RS;                                           // floats per staged input row: 3 + 3 W + 3, rounded up to a multiple of 4
    int scr_off;                                      // floats in front of the 8 x 4 KiB transpose scratch
};

template <int NSUB>
__global__ __launch_bounds__(512, 2) void k_stem3(const Stem3Args p) {
    constexpr int C = 32 * NSUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const s_in = reinterpret_cast<float*>(smem);                        // [2 R + 1][RS] + 16 floats of tail
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.bands, band = blockIdx.x - b * p.bands;
    const int y0 = band * p.R;
    const int rows = min(p.R, p.Ho - y0);
    const int npix = rows * p.Wo;
    const int nin = 2 * rows + 1;                                              // input rows 2 y0 - 1 .. 2 y0 + 2 rows - 1

    // weight fragments of this lane (L2-resident: 6 KB per 32 channels), requested before the staging
    bf16x8 fh[NSUB][3], fl[NSUB][3];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const unsigned char* wp = p.wf + ((size_t)(j * 3 + s) * 64 + lane) * 32;
            fh[j][s] = *reinterpret_cast<const bf16x8*>(wp);
            fl[j][s] = *reinterpret_cast<const bf16x8*>(wp + 16);
        }
    // stage: row r of the band = image row 2 y0 - 1 + r (zero outside the image), floats [3, 3 + 3 W) of the slot; the rest zero
    const int rowf = 3 * p.W;
    for (int i = tid; i < nin * p.RS + 16; i += 512) {
        const int r = i / p.RS, c = i - r * p.RS - 3;
        const int iy = 2 * y0 - 1 + r;
        const bool ok = r < nin && c >= 0 && c < rowf && iy >= 0 && iy < p.H;
        s_in[i] = ok ? p.x[((size_t)b * p.H + iy) * rowf + c] : 0.f;
    }
    __syncthreads();
    if (wave * 32 >= npix) return;

    const int pm = wave * 32 + l31;
    const int pmc = min(pm, npix - 1);
    const int ly = pmc / p.Wo, ox = pmc - ly * p.Wo;
    // window of the pixel in staged row 2 ly + ky: floats 6 ox .. 6 ox + 8 (slot 0 = image column 2 ox - 1); this lane reads + 8 h
    const float* src = s_in + (2 * ly) * p.RS + 6 * ox + 8 * h;
    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        bf16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = src[s * p.RS + e];
            const __bf16 hb = (__bf16)v;
            bh[e] = hb;
            bl[e] = (__bf16)(v - (float)hb);
        }
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[j][s], bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[j][s], bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[j][s], bh, acc[j], 0, 0, 0);
        }
    }
    // C layout (lane = pixel l31, register r = channel (r & 3) + 8 (r >> 2) + 4 h of the subtile) -> this wave's 32 x 32 scratch
    // (16-byte slots XOR-swizzled with the pixel) -> rows of 32 channels: a store instruction writes 8 whole 128-byte pixel rows
    // (partial-line stores straight from the C layout ran at a quarter of the rate)
    float* const scr = s_in + p.scr_off + wave * 1024;
    const int trw = lane >> 3, tc = lane & 7;
    float* const dst = p.out + (((size_t)b * p.Ho + y0) * p.Wo + wave * 32) * C;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = {acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
            *reinterpret_cast<f32x4*>(scr + l31 * 32 + (((2 * q4 + h) ^ (l31 & 7)) << 2)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + 32 * j + 4 * tc);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = trw + 8 * it;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tc ^ (row & 7)) << 2));
            v = v + sh;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (wave * 32 + row < npix)
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + (size_t)row * C + 32 * j + 4 * tc));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NSUB>
static int launch_stem3(const Stem3Args& a, hipStream_t st) {
    const size_t lds = (size_t)a.scr_off * 4 + 8 * 4096;
    LDN_REQUIRE(lds <= 96 * 1024, "ldn_stem3_conv: %zu B of LDS (image too wide)", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_stem3<NSUB>), lds), "k_stem3: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL((k_stem3<NSUB>), dim3((unsigned)(a.B * a.bands)), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_stem3");
    return LDN_OK;
}

}  // namespace ldn

using namespace ldn;

extern "C" size_t ldn_stem_weight_bytes(int cout) { return cout > 0 && cout % 32 == 0 ? (size_t)(cout / 32) * S_KSTEPS * 64 * 32 : 0; }

static int stem_conv_pool_impl(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout,
                               float* out, int Hp, int Wp, float* gap, void* stream);
extern "C" int ldn_stem_conv_pool(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout,
                                  float* out, int Hp, int Wp, void* stream) {
    return stem_conv_pool_impl(x, B, H, W, w_frag, shift, cout, out, Hp, Wp, nullptr, stream);
}
extern "C" int ldn_stem_gap_splits(int H, int W) {
    if (H < 1 || W < 1) return 0;
    const int Hp = (((H - 1) / 2 + 1) - 1) / 2 + 1, Wp = (((W - 1) / 2 + 1) - 1) / 2 + 1;
    return ceil_div(Hp, S_PH) * ceil_div(Wp, S_PW);
}
extern "C" int ldn_stem_conv_pool_gap(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout,
                                      float* out, int Hp, int Wp, float* gap, void* stream) {
    LDN_REQUIRE(gap && (uintptr_t)gap % 16 == 0, "ldn_stem_conv_pool_gap: gap must be a 16-byte aligned [B][ldn_stem_gap_splits(H, W)][cout] buffer");
    return stem_conv_pool_impl(x, B, H, W, w_frag, shift, cout, out, Hp, Wp, gap, stream);
}
static int stem_conv_pool_impl(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout,
                               float* out, int Hp, int Wp, float* gap, void* stream) {
    LDN_REQUIRE(x && w_frag && shift && out, "ldn_stem_conv_pool: null pointer");
    LDN_REQUIRE(B > 0 && H > 0 && W > 0, "ldn_stem_conv_pool: bad geometry");
    LDN_REQUIRE(cout == 32 || cout == 64, "ldn_stem_conv_pool: cout must be 32 or 64 (got %d)", cout);
    LDN_REQUIRE((uintptr_t)w_frag % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)shift % 16 == 0 && (uintptr_t)x % 4 == 0,
                "ldn_stem_conv_pool: w_frag / shift / out must be 16-byte aligned");
    StemArgs a{};
    a.x = x; a.B = B; a.H = H; a.W = W;
    a.wf = static_cast<const unsigned char*>(w_frag); a.shift = shift; a.out = out; a.C = cout;
    a.Hc = (H - 1) / 2 + 1; a.Wc = (W - 1) / 2 + 1;            // conv 7x7 stride 2 pad 3
    a.Hp = (a.Hc - 1) / 2 + 1; a.Wp = (a.Wc - 1) / 2 + 1;      // max-pool 3x3 stride 2 pad 1
    LDN_REQUIRE(Hp == a.Hp && Wp == a.Wp, "ldn_stem_conv_pool: output must be %dx%d for a %dx%d input (got %dx%d)", a.Hp, a.Wp, H, W, Hp, Wp);
    a.tiles_y = ceil_div(a.Hp, S_PH); a.tiles_x = ceil_div(a.Wp, S_PW);
    LDN_REQUIRE((long)B * a.tiles_y * a.tiles_x < (1L << 31), "ldn_stem_conv_pool: too many tiles");
    a.ntiles = B * a.tiles_y * a.tiles_x;
    a.gap = gap;
    int cus = 0;
    if (ldn_device_cus(&cus) != LDN_OK || cus <= 0) cus = 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return cout == 64 ? launch_stem<2>(a, cus, st) : launch_stem<1>(a, cus, st);
}

extern "C" size_t ldn_stem3_weight_bytes(int cout) { return cout > 0 && cout % 32 == 0 ? (size_t)(cout / 32) * 3 * 64 * 32 : 0; }

extern "C" int ldn_stem3_conv(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout, int relu,
                              float* out, int Ho, int Wo, void* stream) {
    LDN_REQUIRE(x && w_frag && shift && out, "ldn_stem3_conv: null pointer");
    LDN_REQUIRE(B > 0 && H > 0 && W > 0, "ldn_stem3_conv: bad geometry");
    LDN_REQUIRE(cout == 32 || cout == 64, "ldn_stem3_conv: cout must be 32 or 64 (got %d)", cout);
    LDN_REQUIRE((uintptr_t)w_frag % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)shift % 16 == 0 && (uintptr_t)x % 4 == 0,
                "ldn_stem3_conv: w_frag / shift / out must be 16-byte aligned");
    Stem3Args a{};
    a.x = x; a.B = B; a.H = H; a.W = W;
    a.wf = static_cast<const unsigned char*>(w_frag); a.shift = shift; a.relu = relu; a.out = out; a.C = cout;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;            // conv 3x3 stride 2 pad 1
    LDN_REQUIRE(Ho == a.Ho && Wo == a.Wo, "ldn_stem3_conv: output must be %dx%d for a %dx%d input (got %dx%d)", a.Ho, a.Wo, H, W, Ho, Wo);
    LDN_REQUIRE(a.Wo <= 256, "ldn_stem3_conv: images wider than 512 pixels are not built (got %d)", W);
    a.R = min(a.Ho, 256 / a.Wo);
    a.bands = ceil_div(a.Ho, a.R);
    a.RS = round_up(3 * W + 6, 4);
    a.scr_off = round_up((2 * a.R + 1) * a.RS + 16, 4);
    LDN_REQUIRE((long)B * a.bands < (1L << 31), "ldn_stem3_conv: too many workgroups");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return cout == 64 ? launch_stem3<2>(a, st) : launch_stem3<1>(a, st);
}
