// k_smallmap -- a WHOLE channel-mode bottleneck on a small map (<= 64 pixels: the 7x7 maps of stage 4) in ONE launch, one
// workgroup per image (gfx950 / CDNA4, bf16x3 arithmetic):
//
//     conv1 (1x1, per-image output subset) -> bn1 + ReLU -> conv2 (3x3, per-image input AND output subsets) -> bn2 + ReLU ->
//     conv3 (1x1, per-image input subset) -> bn3 + residual + ReLU [+ GAP partials for the next block's channel masker]
//
// (reference: imagenet_classification/models/laud_resnet.py:115-144 restricted to the image's active channels, stride 1, identity
// shortcut).  Why its own kernel: at width 512 the "wave = 32 pixels x ALL channels" tiling of k_head / k_tail needs 256 accumulator
// registers, and a 49-pixel image fills two of a workgroup's eight 32-pixel tiles.  Here the eight waves tile the image as
// 2 pixel tiles x 4 channel quarters: wave (t, q) owns pixels [32 t, 32 t + 32) and the n-subtiles j = q + 4 i (i < 4) of the
// image's packed channel list, i.e. at most 4 x 16 accumulator registers.  h1 and h2 (49 x K x 4 B pre-split: 63 KB at K = 320) never
// leave the CU: they live in LDS in the slice-major pre-split layout of k_tail's h1 slices, so the 3x3's tap shift stays a per-lane
// LDS row address and conv3's B operand is read like conv2's.  All weights stream through LDS-DMA rings (counted vmcnt) that
// share what the activations leave of the 160 KiB: the launch is bounded by the weight stream of the image's private subset
// (8.9 MB per image at K = 320, from L2) and the matrix pipe, not by HBM (x in, out once).
// Same arithmetic as ldn_bottleneck_head + ldn_bottleneck_tail (same pre-split weight layouts, products and channel algebra); the
// order of the K sums is the same too, except that conv3 walks K in steps of 16 for groups of 512 output channels.
#include <type_traits>

#include "ldn_common.h"

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SmallArgs {
    const float* x; int ldx;
    int B, H, Wd, HW, cin, W, cout;
    const unsigned char* w1s;                         // [W][cin/8][32 B]
    const unsigned char* w2p;                         // [9][W/2][W/2][16 B]
    const unsigned char* w3p;                         // [W/2][cout][8 B]
    const int32_t* k_idx; const int32_t* k_cnt;       // [B][W], [B]
    const float* sc1; const float* sh1; const float* ps1;   // [W]
    const float* sc2; const float* sh2; const float* ps2;   // [W], [16][W], [W]
    const float* sh3;                                 // [cout]
    const float* residual; int ldr; float* out; int ldo;
    float* colsum;                                    // optional [B][2][cout]
};

__device__ __attribute__((aligned(16))) float g_small_zero[4] = {0.f, 0.f, 0.f, 0.f};

#ifndef SM_ABLATE
#define SM_ABLATE 0     // tuning only (conv3 fast loop): 1 = no MFMA, 2 = no fragment reads, 4 = no DMA (results are wrong, timings are the point)
#endif
#ifdef LDN_TRACE   // tuning only: per-workgroup phase timestamps + wave 0's wait / compute split (tools/trace_small.py)
__device__ unsigned long long* g_small_trace = nullptr;
#define ST(x) x = __builtin_amdgcn_s_memtime();
#define ST_ADD(acc, a, b) acc += (b) - (a);
#else
#define ST(x)
#define ST_ADD(acc, a, b)
#endif

namespace {

#ifndef SM_FORCE_K16
#define SM_FORCE_K16 0
#endif
constexpr int S_LDS = 160 * 1024;
constexpr int S_KIDX_BYTES = 2304;                    // int[W + 32], W <= 512
constexpr int S_ZERO_OFF = S_KIDX_BYTES;              // one all-zero row of 128 B (out-of-image taps)
constexpr int S_ACT_OFF = S_KIDX_BYTES + 128;         // h1, then h2: [K slice][pixel][128 B]
constexpr int S_TAB1_BYTES = 3 * 512 * 4;             // conv1's epilogue tables at the END of the LDS

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {   // 16 B per lane, LDS = lds_base + lane * 16
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);    // wave-uniform by construction; a copy when it already is a scalar register
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// four consecutive 1 KB pieces of one row: source = sbase (wave-uniform) + voff (per-lane byte offset) + f * 1024, LDS = lds_base + lane * 16
// + f * 1024 -- the instruction offset moves BOTH addresses (measured: tools/experiments/dma_offset.hip), so one M0 set-up and one address
// serve the row
__device__ __forceinline__ void dma16x4_row(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    {
        const unsigned long long u = reinterpret_cast<unsigned long long>(sbase);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        sbase = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    }
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
// NF gathered 1 KB pieces of one staged row: piece f = sbase (wave-uniform) + vo[f] (per-lane byte offset, biased by the caller so that the
// instruction offset f * 1024 is part of it) -> LDS lds_base + f * 1024 + lane * 16; one M0 set-up for the row
template <int NF> __device__ __forceinline__ void dma16_gather_row(const unsigned (&vo)[4], const void* sbase, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    if constexpr (NF == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo[0]), "s"(sbase), "s"(lds_base) : "memory");
    else if constexpr (NF == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo[0]), "v"(vo[1]), "s"(sbase), "s"(lds_base) : "memory");
    else if constexpr (NF == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "s"(sbase), "s"(lds_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "s"(sbase), "s"(lds_base) : "memory");
}
// piece F (1 KB) of a row whose scalar base / lane offset / LDS row base were prepared by row_prepare (same addressing as dma16x4_row)
template <int F> __device__ __forceinline__ void dma16_row_piece(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base), "n"(F * 1024) : "memory");
}
__device__ __forceinline__ const void* uniform_ptr(const void* v) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
template <int N> __device__ __forceinline__ void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {   // counted wait with a run-time (wave-uniform) count; more than 15: 15 (over-waits)
    switch (n) {
        case 0: wait_vm_n<0>(); break;   case 1: wait_vm_n<1>(); break;   case 2: wait_vm_n<2>(); break;
        case 3: wait_vm_n<3>(); break;   case 4: wait_vm_n<4>(); break;   case 5: wait_vm_n<5>(); break;
        case 6: wait_vm_n<6>(); break;   case 7: wait_vm_n<7>(); break;   case 8: wait_vm_n<8>(); break;
        case 9: wait_vm_n<9>(); break;   case 10: wait_vm_n<10>(); break; case 11: wait_vm_n<11>(); break;
        case 12: wait_vm_n<12>(); break; case 13: wait_vm_n<13>(); break; case 14: wait_vm_n<14>(); break;
        default: wait_vm_n<15>(); break;
    }
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned lds_off(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}

// LDS geometry of an image with nsub K slices on a map of HW pixels (shared by the kernel and the host-side fit check)
struct SmallGeom {
    int act_bytes, ring_off, space, w2row, nks2, slot2, d2, d3;
};
__host__ __device__ inline SmallGeom small_geom(int HW, int nsub) {
    SmallGeom g;
    g.act_bytes = nsub * HW * 128;
    g.ring_off = S_ACT_OFF + g.act_bytes;
    g.space = S_LDS - g.ring_off;
    g.w2row = (nsub > 0 ? (nsub + 3) / 4 : 1) * 1024;                // bytes of one staged k-pair row of W2: nsub * 32 entries of 8 B, whole DMA instructions
    const int slot32 = 16 * g.w2row;
    // K32 chunks when three of them fit, else K16 chunks in a deeper ring (measured: two K32 slots are behind four K16 slots)
    if (SM_FORCE_K16 == 0 && 3 * slot32 <= g.space) { g.nks2 = 2; g.slot2 = slot32; g.d2 = g.space / slot32 < 3 ? g.space / slot32 : 3; }
    else { g.nks2 = 1; g.slot2 = 8 * g.w2row; g.d2 = g.space / g.slot2 < 4 ? g.space / g.slot2 : 4; }
    g.d3 = g.space / 32768 < 3 ? g.space / 32768 : 3;
    return g;
}

// acc (C layout of the transposed MFMA: lane = pixel, register r = channel (r & 3) + 8 (r >> 2) + 4 h of the subtile) -> u = relu(acc *
// sc + sh) - ps, split, into the pre-split slice-major activation buffer: slice j, pixel row pm, octet qd: [8 hi | 8 lo]
__device__ __forceinline__ void store_presplit(unsigned char* act, int HW, int j, int pm, int h, int qd, const f32x4 v) {
    const bf16x2 h0 = {(__bf16)v[0], (__bf16)v[1]}, h1 = {(__bf16)v[2], (__bf16)v[3]};
    const bf16x2 l0 = {(__bf16)(v[0] - (float)h0[0]), (__bf16)(v[1] - (float)h0[1])};
    const bf16x2 l1 = {(__bf16)(v[2] - (float)h1[0]), (__bf16)(v[3] - (float)h1[1])};
    const u32x2 hi = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    const u32x2 lo = {__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
    const unsigned sw = ((unsigned)pm >> 1) & 7u;
    unsigned char* row = act + (size_t)j * HW * 128 + pm * 128;
    *reinterpret_cast<u32x2*>(row + (((2u * qd) ^ sw) << 4) + 8 * h) = hi;
    *reinterpret_cast<u32x2*>(row + (((2u * qd + 1u) ^ sw) << 4) + 8 * h) = lo;
}

// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63), valid in the LAST lane of the half: four DPP adds inside the rows of 16
// lanes (xor 1, xor 2, half mirror, mirror), then row_bcast15 carries a row's total into the next row (VALU only: __shfl_xor is an LDS
// instruction with an address computation per step)
__device__ __forceinline__ float half_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});    // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});    // row_mirror
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});    // row_bcast15 into rows 1 and 3
    return v;
}

}  // namespace

__global__ __launch_bounds__(512, 2) void k_smallmap(const SmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int t = wave & 1, q = wave >> 1;                     // pixel tile, channel quarter
    const int b = blockIdx.x;
    const int W = p.W, HW = p.HW;
#ifdef LDN_TRACE
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, ta = 0, tb = 0, wt[3] = {0, 0, 0}, ct[3] = {0, 0, 0}, x3[3] = {0, 0, 0};
    ST(ts[0])
#endif
    int* const s_kidx = reinterpret_cast<int*>(smem);
    unsigned char* const s_act = smem + S_ACT_OFF;

    const int Kb = min(p.k_cnt[b], W);
    const int nsub = __builtin_amdgcn_readfirstlane(ceil_div(Kb, 32));
    const int Kp = nsub * 32;
    for (int i = tid; i < W + 32; i += 512) s_kidx[i] = i < Kb ? p.k_idx[(size_t)b * W + i] : -1;
    if (tid < 32) reinterpret_cast<float*>(smem + S_ZERO_OFF)[tid] = 0.f;
    LDN_DCHECK(p.k_cnt[b] >= 0 && p.k_cnt[b] <= W && (p.k_cnt[b] & 1) == 0, 701);
    if (tid < Kb) {
        const int ch = p.k_idx[(size_t)b * W + tid];
        LDN_DCHECK(ch >= 0 && ch < W, 702);
        LDN_DCHECK((tid & 1) ? (ch == p.k_idx[(size_t)b * W + tid - 1] + 1) : ((ch & 1) == 0), 703);
        LDN_DCHECK(tid == 0 || ch > p.k_idx[(size_t)b * W + tid - 1], 704);
    }
    __syncthreads();

    SmallGeom G = small_geom(HW, nsub);
    // (integer divisions by run-time values run on the vector unit: their wave-uniform results are pinned to scalar registers here, the
    // LDS-DMA base addresses derived from them must be scalar operands)
    G.d2 = __builtin_amdgcn_readfirstlane(G.d2); G.d3 = __builtin_amdgcn_readfirstlane(G.d3);
    G.nks2 = __builtin_amdgcn_readfirstlane(G.nks2); G.slot2 = __builtin_amdgcn_readfirstlane(G.slot2);
    G.w2row = __builtin_amdgcn_readfirstlane(G.w2row); G.ring_off = __builtin_amdgcn_readfirstlane(G.ring_off);
    const int pm = 32 * t + l31;
    const bool pvalid = pm < HW;
    const bool active = 32 * t < HW;                           // this wave's pixel tile holds pixels
    const int prow = pvalid ? pm : 0;                          // a row that exists (lanes beyond the map compute a copy, never stored)
    const long img_row0 = (long)b * HW;
    const unsigned lds_ring = lds_off(smem + G.ring_off);

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (nsub > 0) {
        // ==================================================================================================== conv1 (1x1)
        // ring slot = [64 x rows | wrows weight rows] x 128 B over everything behind the channel list (h1 does not exist yet).  The weight
        // rows are staged WAVE-MAJOR: piece i (rows 64 i + 8 w .. + 7 of the image's list) of wave w lies at 1 KB block w * nw + i, so that a
        // wave's pieces are consecutive kilobytes and up to four of them share one address set-up (dma16_gather_row).  Software pipeline
        // over the K16 steps as in conv2 / conv3: the raw fragments of step u + 1 are read while the MFMAs of step u run.
        {
            const int wrows = round_up(Kb, 64);
            const int slot1 = (64 + wrows) * 128;
            const int D = __builtin_amdgcn_readfirstlane(min(4, (S_LDS - S_ACT_OFF) / slot1));            // >= 2 (W <= 512)
            const int nw = wrows / 64;
            const bool xw = 8 * wave < HW;                                 // this wave stages x rows 8 w .. 8 w + 7
            const int ipc = (xw ? 1 : 0) + nw;
            const int nchunks = p.cin / 32;
            const unsigned ring1 = lds_off(s_act), ring_bytes = (unsigned)(D * slot1);
            // per-lane source offsets, fixed for the whole K loop (rows beyond the map / the list fetch existing ones: never multiplied / zero tables)
            unsigned xo[4] = {0, 0, 0, 0};
            {
                const int r = 8 * wave + (lane >> 3);
                const int ls = (lane & 7) ^ ((r >> 1) & 7);
                xo[0] = (unsigned)(((img_row0 + min(r, HW - 1)) * p.ldx + ls * 4) * 4);
            }
            unsigned wo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r8 = lane >> 3;
                const int r = i * 64 + wave * 8 + r8;
                const int ls = (lane & 7) ^ (r8 >> 1);
                const int ch = s_kidx[min(r, Kb - 1)];
                wo[i] = (unsigned)(ch * p.cin * 4 + ls * 16 + (3 - (i & 3)) * 1024);
            }
            unsigned foff = 0; int fc = 0;
            auto gather = [&](int nf, const unsigned (&vo)[4], const void* sbase, unsigned lds) {
                if (nf >= 4) dma16_gather_row<4>(vo, sbase, lds);
                else if (nf == 3) dma16_gather_row<3>(vo, sbase, lds);
                else if (nf == 2) dma16_gather_row<2>(vo, sbase, lds);
                else if (nf == 1) dma16_gather_row<1>(vo, sbase, lds);
            };
            const unsigned wa4[4] = {wo[0], wo[1], wo[2], wo[3]}, wb4[4] = {wo[4], wo[5], wo[6], wo[7]};
            auto issue = [&]() {
                const int c = min(fc, nchunks - 1);             // chunks beyond the K loop: the last one again (keeps the per-chunk count)
                const unsigned slot = ring1 + foff;
                if (xw) dma16_gather_row<1>(xo, uniform_ptr(p.x + c * 32), slot + 8 * wave * 128);
                const void* wb = uniform_ptr(p.w1s + (long)c * 128 - 3072);
                gather(min(nw, 4), wa4, wb, slot + (64 * 128) + (wave * nw) * 1024);
                if (nw > 4) gather(nw - 4, wb4, wb, slot + (64 * 128) + (wave * nw + 4) * 1024);
                foff += slot1; if (foff == ring_bytes) foff = 0;
                ++fc;
            };
            for (int c = 0; c < D; ++c) issue();
            const int nv = nsub > q ? (nsub - q + 3) / 4 : 0;             // this wave's n-subtiles: q, q + 4, ...
            const unsigned xsw = ((unsigned)pm >> 1) & 7u;
            const unsigned wsw = ((unsigned)l31 >> 1) & 3u;              // weight rows: swizzled by the row within its 8-row piece
            // weight fragment row of this lane for subtile j = q + 4 i: list row n = 32 j + l31 = piece n / 64 of wave (n % 64) / 8
            const unsigned w_rd = (unsigned)((64 + ((4 * (q & 1) + (l31 >> 3)) * nw + (q >> 1)) * 8 + (l31 & 7)) * 128);   // + i * 2048
            unsigned coff = 0;
            // the loop is instantiated per subtile count of the wave (0: a wave without pixels or channels only stages and synchronises):
            // predicated fragment reads cost copies of the whole register set at every branch join
            auto run = [&](auto NVc) {
                constexpr int NV = decltype(NVc)::value;
                struct Frag { f32x4 x0, x1; bf16x8 ah[NV > 0 ? NV : 1], al[NV > 0 ? NV : 1]; };
                auto load_frag = [&](Frag& f, int kk) {
                    if constexpr (NV > 0) {
                        const unsigned char* xs = smem + S_ACT_OFF + coff;
                        const unsigned sl = 4u * kk + 2u * h;           // x: fp32 k .. k+3 | k+4 .. k+7 of this lane's half of the K16 step
                        f.x0 = *reinterpret_cast<const f32x4*>(xs + pm * 128 + ((sl ^ xsw) << 4));
                        f.x1 = *reinterpret_cast<const f32x4*>(xs + pm * 128 + (((sl + 1) ^ xsw) << 4));
                        const unsigned char* wh = xs + w_rd + ((sl ^ wsw) << 4);          // weights: octet 2 kk + h = slots (hi, lo)
                        const unsigned char* wl = xs + w_rd + (((sl + 1) ^ wsw) << 4);
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            f.ah[i] = *reinterpret_cast<const bf16x8*>(wh + i * 2048);
                            f.al[i] = *reinterpret_cast<const bf16x8*>(wl + i * 2048);
                        }
                    }
                };
                auto mfma_frag = [&](const Frag& f) {
                    if constexpr (NV > 0) {
                        bf16x8 bh, bl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = pvalid ? (e < 4 ? f.x0[e] : f.x1[e - 4]) : 0.f;
                            const __bf16 hb = (__bf16)v;
                            bh[e] = hb;
                            bl[e] = (__bf16)(v - (float)hb);
                        }
#pragma unroll
                        for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], bh, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], bl, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], bh, acc[i], 0, 0, 0);
                    }
                };
                Frag f0, f1;
                wait_vm_rt(ipc * (D - 1));                       // chunk 0
                lds_barrier();
                load_frag(f0, 0);
#pragma unroll 1
                for (int c = 0; c < nchunks; ++c) {
                    load_frag(f1, 1);
                    mfma_frag(f0);
                    ST(ta)
                    wait_vm_rt(ipc * (D - 2));                   // chunk c + 1 has landed ...
                    lds_barrier();                               // ... for every wave; every wave has read chunk c
                    ST(tb)
                    ST_ADD(wt[0], ta, tb)
                    issue();
                    coff += slot1; if (coff == ring_bytes) coff = 0;
                    load_frag(f0, 0);
                    mfma_frag(f1);
                }
            };
            switch (active ? nv : 0) {
                case 0: run(std::integral_constant<int, 0>{}); break;
                case 1: run(std::integral_constant<int, 1>{}); break;
                case 2: run(std::integral_constant<int, 2>{}); break;
                case 3: run(std::integral_constant<int, 3>{}); break;
                default: run(std::integral_constant<int, 4>{}); break;
            }
            wait_vm_n<0>();
            lds_barrier();                                     // the ring is dead: its place is h1's
            ST(ts[1])
        }
        // ---- epilogue 1: h1 = relu(sc1 * conv1 + sh1) - c1 into the activation buffer
        {
            float* const s_tab = reinterpret_cast<float*>(smem + S_LDS - S_TAB1_BYTES);     // [3][512]
            for (int i = tid; i < 3 * Kp; i += 512) {
                const int k = i / Kp, n = i - k * Kp;
                const int ch = s_kidx[n];
                const float* src = k == 0 ? p.sc1 : (k == 1 ? p.sh1 : p.ps1);
                s_tab[k * 512 + n] = ch >= 0 ? src[ch] : 0.f;
            }
            lds_barrier();
            if (pvalid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = q + 4 * i;
                    if (j < nsub) {
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const int n0 = 32 * j + 8 * qd + 4 * h;
                            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
                            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + 512 + n0);
                            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + 1024 + n0);
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][4 * qd + e] * sc[e] + sh[e], 0.f) - ps[e];
                            store_presplit(s_act, HW, j, pm, h, qd, v);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            lds_barrier();                                     // h1 complete; the tables are free (they overlap the W2 ring)
            ST(ts[2])
        }

        // ==================================================================================================== conv2 (3x3)
        {
            const int oy = pm / p.Wd, ox = pm - oy * p.Wd;
            unsigned tmask = 0;                                // bit tp: tap tp of this lane's pixel lies inside the map (else: the zero row)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int iy = oy + tp / 3 - 1, ix = ox + tp % 3 - 1;
                const bool ok = pvalid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
                tmask |= ok ? (1u << tp) : 0u;
            }
            const int cls = (((oy - 1 < 0) | ((oy + 1 >= p.H) << 1)) * 4 + ((ox - 1 < 0) | ((ox + 1 >= p.Wd) << 1)));
            const int nks = G.nks2, D = G.d2, W2ROW = G.w2row, slotB = G.slot2;
            const int nf = W2ROW / 1024;                       // DMA instructions per k-pair row
            const int per = nks == 2 ? 1 : 2;                  // chunks per (slice, tap)
            const int nchunks = nsub * 9 * per;
            const int ipc = nks * nf;
            int npo[4];                                        // per-lane n-pair source offsets (bytes within a k-pair row of w2p)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int v = 64 * f + lane;
                const int ch = 2 * v < Kb ? s_kidx[2 * v] : -1;
                npo[f] = ch >= 0 ? (ch >> 1) * 16 : -1;
            }
            if (D > 1) {
                // ---- the common case (two or more ring slots): a software pipeline over the K16 steps, as in conv3 below -- while the MFMAs
                // of a step run, the RAW fragments of the next step are read, and the rows of chunk c + D are issued (one address set-up per
                // row) into the slot the current chunk's fragments were read from.  A step = 8 k-pair rows of (slice, tap); a chunk holds
                // two steps (nks 2) or one.  Rows / n-pairs beyond the image's list fetch listed ones: their products are multiplied by exact
                // zeros (h1 columns beyond the list) or land in accumulator columns whose epilogue tables are zero.
                unsigned vo[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int v = 64 * f + lane;
                    vo[f] = (unsigned)((s_kidx[2 * v < Kb ? 2 * v : 0] >> 1) * 16 + (3 - f) * 1024);
                }
                const long tapstride = (long)(W / 2) * (W / 2) * 16;
                const unsigned ring_bytes = (unsigned)(D * slotB);
                int is = 0, itp = 0, ihf = 0; unsigned foff = 0;
                int rbv[2];                                     // row byte offsets (within a tap) of this wave's rows of slice `is`
                auto load_rb = [&]() {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        rbv[e] = (s_kidx[min(32 * min(is, nsub - 1) + (nks == 2 ? 4 * wave + 2 * e : 16 * e + 2 * wave), Kb - 2)] >> 1) * (W / 2) * 16;
                };
                load_rb();
                auto row_dma = [&](const void* sbase, unsigned lds) {
                    if (nf == 3) dma16_gather_row<3>(vo, sbase, lds);
                    else if (nf == 4) dma16_gather_row<4>(vo, sbase, lds);
                    else if (nf == 2) dma16_gather_row<2>(vo, sbase, lds);
                    else dma16_gather_row<1>(vo, sbase, lds);
                };
                auto issue = [&]() {
                    const unsigned char* tapbase = p.w2p + (long)itp * tapstride - 3072;
                    const unsigned slot = lds_ring + foff;
                    if (nks == 2) {
                        row_dma(uniform_ptr(tapbase + __builtin_amdgcn_readfirstlane(rbv[0])), slot + (2 * wave) * W2ROW);
                        row_dma(uniform_ptr(tapbase + __builtin_amdgcn_readfirstlane(rbv[1])), slot + (2 * wave + 1) * W2ROW);
                    } else {
                        row_dma(uniform_ptr(tapbase + __builtin_amdgcn_readfirstlane(ihf ? rbv[1] : rbv[0])), slot + wave * W2ROW);
                    }
                    foff += slotB; if (foff == ring_bytes) foff = 0;
                    if (++ihf == per) { ihf = 0; if (++itp == 9) { itp = 0; ++is; load_rb(); } }
                };
                for (int c = 0; c < D; ++c) issue();
                const int nv = nsub > q ? (nsub - q + 3) / 4 : 0;             // this wave's n-subtiles: q, q + 4, ...
                const unsigned a_rd = (unsigned)(4 * h * W2ROW + l31 * 8 + q * 256);
                unsigned coff = 0;                               // slot offset of the chunk the next fragments are read from
                const unsigned half_rows = (unsigned)(8 * W2ROW);
                auto run = [&](auto NVc) {
                    constexpr int NV = decltype(NVc)::value;    // the wave's subtile count as a constant (see conv1)
                    struct Frag { bf16x8 bh, bl; u32x2 e[NV > 0 ? NV : 1][4]; };
                    // fragments of step (slice s, tap tp, K16 step kk); `rows` = byte offset of the step's 8 rows within the chunk's slot
                    auto load_frag = [&](Frag& f, int s, int tp, int kk, unsigned rows) {
                        if constexpr (NV > 0) {
                            const int ty = (tp * 11) >> 5;                                   // tp / 3 for tp < 9
                            const int r = ((tmask >> tp) & 1u) ? pm + (ty - 1) * p.Wd + (tp - 3 * ty - 1) : -1;
                            const unsigned char* hrow = r >= 0 ? s_act + (size_t)s * HW * 128 + r * 128 : smem + S_ZERO_OFF;
                            const unsigned rx = r >= 0 ? ((unsigned)r >> 1) & 7u : 0u;
                            const unsigned sl = 2u * (2u * kk + h);
                            f.bh = *reinterpret_cast<const bf16x8*>(hrow + ((sl ^ rx) << 4));
                            f.bl = *reinterpret_cast<const bf16x8*>(hrow + (((sl + 1) ^ rx) << 4));
                            const unsigned char* w0 = smem + G.ring_off + coff + rows + a_rd;      // the four k-pair rows of this lane's half
                            const unsigned char* w1 = w0 + W2ROW;
                            const unsigned char* w2 = w1 + W2ROW;
                            const unsigned char* w3 = w2 + W2ROW;
#pragma unroll
                            for (int i = 0; i < NV; ++i) {
                                f.e[i][0] = *reinterpret_cast<const u32x2*>(w0 + i * 1024);
                                f.e[i][1] = *reinterpret_cast<const u32x2*>(w1 + i * 1024);
                                f.e[i][2] = *reinterpret_cast<const u32x2*>(w2 + i * 1024);
                                f.e[i][3] = *reinterpret_cast<const u32x2*>(w3 + i * 1024);
                            }
                        }
                    };
                    auto mfma_frag = [&](const Frag& f) {
                        if constexpr (NV > 0) {
                            bf16x8 ah[NV], al[NV];
#pragma unroll
                            for (int i = 0; i < NV; ++i) {
                                const u32x4 ahu = {f.e[i][0][0], f.e[i][1][0], f.e[i][2][0], f.e[i][3][0]};
                                const u32x4 alu = {f.e[i][0][1], f.e[i][1][1], f.e[i][2][1], f.e[i][3][1]};
                                ah[i] = __builtin_bit_cast(bf16x8, ahu); al[i] = __builtin_bit_cast(bf16x8, alu);
                            }
#pragma unroll
                            for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], f.bh, acc[i], 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], f.bl, acc[i], 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], f.bh, acc[i], 0, 0, 0);
                        }
                    };
                    auto sync_issue = [&]() {          // the next chunk has landed for every wave; every wave has read the current one
                        ST(ta)
                        wait_vm_rt(ipc * (D - 2));
                        lds_barrier();
                        ST(tb)
                        ST_ADD(wt[1], ta, tb)
                        issue();
                        coff += slotB; if (coff == ring_bytes) coff = 0;
                    };
                    Frag f0, f1;
                    wait_vm_rt(ipc * (D - 1));                   // chunk 0
                    lds_barrier();
                    load_frag(f0, 0, 0, 0, 0);
#pragma unroll 1
                    for (int s = 0; s < nsub; ++s) {
#pragma unroll 1
                        for (int tp = 0; tp < 9; ++tp) {
                            const int tn = tp == 8 ? 0 : tp + 1, sn = tp == 8 ? (s + 1 < nsub ? s + 1 : s) : s;   // the (slice, tap) after this one
                            if (nks == 1) sync_issue();
                            load_frag(f1, s, tp, 1, nks == 2 ? half_rows : 0u);
                            mfma_frag(f0);
                            sync_issue();
                            load_frag(f0, sn, tn, 0, 0u);
                            mfma_frag(f1);
                        }
                    }
                };
                ST(ta)
                switch (active ? nv : 0) {
                    case 0: run(std::integral_constant<int, 0>{}); break;
                    case 1: run(std::integral_constant<int, 1>{}); break;
                    case 2: run(std::integral_constant<int, 2>{}); break;
                    case 3: run(std::integral_constant<int, 3>{}); break;
                    default: run(std::integral_constant<int, 4>{}); break;
                }
                ST(tb)
                ST_ADD(ct[1], ta, tb)
            } else {
            // the issue stream walks (slice, tap, half) one chunk at a time; the k-pair rows this wave stages change with the slice only
            int is = 0, itp = 0, ihf = 0, ioff = 0, coff = 0;  // coordinates / slot offset of the next chunk to issue; slot offset of the chunk consumed
            int kv[2] = {-1, -1};                               // nks 2: channels of k-pair rows 2 w, 2 w + 1; nks 1: of row w of half 0 / 1
            auto load_kv = [&]() {
#pragma unroll
                for (int e = 0; e < 2; ++e) kv[e] = is < nsub ? s_kidx[32 * is + (nks == 2 ? 4 * wave + 2 * e : 16 * e + 2 * wave)] : -1;
            };
            load_kv();
            // DMA instruction k = 4 e + f of the next chunk: piece f of this wave's row e; chunks beyond the loop: zero lines
            auto issue_part = [&](int k) {
                const int e = k >> 2, f = k & 3;
                if (e < nks && f < nf) {
                    const unsigned slot = lds_ring + ioff;
                    const int u = nks == 2 ? 2 * wave + e : wave;
                    const int kch = is < nsub ? (nks == 2 ? (e ? kv[1] : kv[0]) : (ihf ? kv[1] : kv[0])) : -1;
                    const int rowoff = (itp * (W / 2) + (kch >> 1)) * (W / 2) * 16;
                    const int np = f == 0 ? npo[0] : f == 1 ? npo[1] : f == 2 ? npo[2] : npo[3];
                    const unsigned char* src = (kch >= 0 && np >= 0) ? p.w2p + rowoff + np : reinterpret_cast<const unsigned char*>(g_small_zero);
                    dma16(src, slot + u * W2ROW + f * 1024);
                }
            };
            auto issue_advance = [&]() {
                ioff += slotB; if (ioff == D * slotB) ioff = 0;
                if (++ihf == per) { ihf = 0; if (++itp == 9) { itp = 0; ++is; load_kv(); } }
            };
            auto issue = [&]() {
#pragma unroll
                for (int k = 0; k < 8; ++k) issue_part(k);
                issue_advance();
            };
            if (D > 1) for (int c = 0; c < D - 1; ++c) issue();
            const unsigned a_lane = (unsigned)(4 * h * W2ROW + l31 * 8);
            unsigned jo[4];                                     // byte offset of the subtile's entries within a staged row
#pragma unroll
            for (int i = 0; i < 4; ++i) jo[i] = (unsigned)(min(q + 4 * i, nsub - 1) * 256);
            int c = 0;
#pragma unroll 1
            for (int s = 0; s < nsub; ++s) {
                const unsigned char* hs = s_act + (size_t)s * HW * 128;
                // the tap loop stays ROLLED (the tap's row is computed, not looked up): unrolled nine times with both chunk shapes the kernel
                // outgrows the instruction cache it shares with the neighbouring CU
#pragma unroll 1
                for (int tp = 0; tp < 9; ++tp) {
                    const int ty = (tp * 11) >> 5;                                   // tp / 3 for tp < 9
                    const int r = ((tmask >> tp) & 1u) ? pm + (ty - 1) * p.Wd + (tp - 3 * ty - 1) : -1;
                    const unsigned char* hrow = r >= 0 ? hs + r * 128 : smem + S_ZERO_OFF;
                    const unsigned rx = r >= 0 ? ((unsigned)r >> 1) & 7u : 0u;
#pragma unroll 1
                    for (int hf = 0; hf < per; ++hf, ++c) {
                        ST(ta)
                        if (D == 1) { lds_barrier(); issue(); wait_vm_n<0>(); }
                        else wait_vm_rt(ipc * (D - 2));
                        lds_barrier();
                        ST(tb)
                        ST_ADD(wt[1], ta, tb)
                        const unsigned char* ws = smem + G.ring_off + coff;
                        coff += slotB; if (coff == D * slotB) coff = 0;
                        if (!active) { if (D > 1) issue(); continue; }
                        ST(ta)
                        if (nks == 2) {
                            bf16x8 bh[2], bl[2];
                            u32x2 e[2][4][4];
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) {
                                const unsigned sl = 2u * (2u * ks + h);
                                bh[ks] = *reinterpret_cast<const bf16x8*>(hrow + ((sl ^ rx) << 4));
                                bl[ks] = *reinterpret_cast<const bf16x8*>(hrow + (((sl + 1) ^ rx) << 4));
                            }
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
#pragma unroll
                                    for (int qq = 0; qq < 4; ++qq)
                                        e[ks][i][qq] = *reinterpret_cast<const u32x2*>(ws + a_lane + (8 * ks + qq) * W2ROW + jo[i]);
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (q + 4 * i < nsub) {
                                        const u32x4 ahu = {e[ks][i][0][0], e[ks][i][1][0], e[ks][i][2][0], e[ks][i][3][0]};
                                        const u32x4 alu = {e[ks][i][0][1], e[ks][i][1][1], e[ks][i][2][1], e[ks][i][3][1]};
                                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc[i], 0, 0, 0);
                                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc[i], 0, 0, 0);
                                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc[i], 0, 0, 0);
                                    }
                                    if (D > 1) issue_part(4 * ks + i);     // the next chunk's DMA between the MFMA groups (see conv1)
                                }
                            if (D > 1) issue_advance();
                        } else {
                            const unsigned sl = 2u * (2u * (unsigned)hf + h);
                            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hrow + ((sl ^ rx) << 4));
                            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hrow + (((sl + 1) ^ rx) << 4));
                            u32x2 e[4][4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int qq = 0; qq < 4; ++qq) e[i][qq] = *reinterpret_cast<const u32x2*>(ws + a_lane + qq * W2ROW + jo[i]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (q + 4 * i < nsub) {
                                    const u32x4 ahu = {e[i][0][0], e[i][1][0], e[i][2][0], e[i][3][0]};
                                    const u32x4 alu = {e[i][0][1], e[i][1][1], e[i][2][1], e[i][3][1]};
                                    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
                                }
                                if (D > 1) issue_part(i);
                            }
                            if (D > 1) issue_advance();
                        }
                        ST(tb)
                        ST_ADD(ct[1], ta, tb)
                    }
                }
            }
            }
            wait_vm_n<0>();
            lds_barrier();                                     // every wave has left h1 and the W2 ring
            ST(ts[3])
            // ---- epilogue 2: h2 = relu(sc2 * conv2 + sh2[class]) - c2 over h1's place; tables in the dead ring
            float* const s_tab = reinterpret_cast<float*>(smem + G.ring_off);        // sc2 [Kp] | ps2 [Kp] | sh2 [16][Kp]
            for (int i = tid; i < 18 * Kp; i += 512) {
                const int k = i / Kp, n = i - k * Kp;
                const int ch = s_kidx[n];
                const float* src = k == 0 ? p.sc2 : (k == 1 ? p.ps2 : p.sh2 + (size_t)(k - 2) * W);
                s_tab[i] = ch >= 0 ? src[ch] : 0.f;
            }
            lds_barrier();
            if (pvalid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = q + 4 * i;
                    if (j < nsub) {
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const int n0 = 32 * j + 8 * qd + 4 * h;
                            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
                            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + Kp + n0);
                            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + (2 + cls) * Kp + n0);
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][4 * qd + e] * sc[e] + sh[e], 0.f) - ps[e];
                            store_presplit(s_act, HW, j, pm, h, qd, v);
                        }
                    }
                }
            }
            lds_barrier();                                     // h2 complete; the tables are free (the W3 ring takes their place)
            ST(ts[4])
        }
    }

    // ======================================================================================================== conv3 (1x1)
    // groups of 512 output channels (wave (t, q): the subtiles 512 g + 32 (q + 4 i)), K walked in steps of 16: chunk (g, ks) = 8 k-pair
    // rows x 512 entries of 8 B = 32 KiB; every wave stages one row (4 instructions)
    {
        const int D = G.d3;
        const int nk16 = Kp / 16;
        const int ngroups = ceil_div(p.cout, 512);
        const int nchunks = ngroups * nk16;
        int ig = 0, iks = 0, ioff = 0, coff = 0;              // coordinates / slot offset of the next chunk to issue; slot offset of the chunk consumed
        const unsigned char* i3base = nullptr;                // source of the next chunk's row for this lane (nullptr: zero lines)
        auto issue_begin = [&]() {
            i3base = nullptr;
            if (ig < ngroups && nk16 > 0) {
                const int kch = s_kidx[16 * iks + 2 * wave];
                if (kch >= 0) i3base = p.w3p + ((long)(kch >> 1) * p.cout + 512 * ig + 2 * lane) * 8;
            }
        };
        auto issue_part = [&](int f) {
            const unsigned slot = lds_ring + ioff;
            const unsigned char* src = (i3base && 512 * ig + 128 * f < p.cout) ? i3base + f * 1024 : reinterpret_cast<const unsigned char*>(g_small_zero);
            dma16(src, slot + wave * 4096 + f * 1024);
        };
        auto issue_advance = [&]() {
            ioff += 32768; if (ioff == D * 32768) ioff = 0;
            if (ig < ngroups && nk16 > 0) { if (++iks == nk16) { iks = 0; ++ig; } }
        };
        auto issue = [&]() {
            issue_begin();
#pragma unroll
            for (int f = 0; f < 4; ++f) issue_part(f);
            issue_advance();
        };
        const bool fast3 = D > 1 && (p.cout & 511) == 0 && nk16 > 0;
        if (D > 1 && !fast3) for (int c = 0; c < D - 1; ++c) issue();
        const unsigned rx = ((unsigned)prow >> 1) & 7u;
        const unsigned a_lane = (unsigned)(4 * h * 4096 + l31 * 8);
        // ---- epilogue of a group: bn3 shift + residual + ReLU straight from the C layout (lane = pixel: 16 B of 4 channels per
        // register quad; the two half-waves and the four quads of a pixel fill its 128-byte line), GAP partials per pixel tile
        auto group_epilogue = [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = 512 * g + 32 * (q + 4 * i);
                if (c0 >= p.cout) continue;
                f32x4 res[4], sh[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int cch = c0 + 8 * qd + 4 * h;
                    const float* rsrc = (p.residual && pvalid) ? p.residual + (img_row0 + pm) * p.ldr + cch : g_small_zero;
                    res[qd] = *reinterpret_cast<const f32x4*>(rsrc);
                    sh[qd] = *reinterpret_cast<const f32x4*>(p.sh3 + cch);
                }
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int cch = c0 + 8 * qd + 4 * h;
                    f32x4 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = fmaxf(acc[i][4 * qd + e] + sh[qd][e] + res[qd][e], 0.f);
                    if (pvalid) *reinterpret_cast<f32x4*>(p.out + (img_row0 + pm) * p.ldo + cch) = x;
                    if (p.colsum) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = half_sum_dpp(pvalid ? x[e] : 0.f);
                        if (l31 == 31) *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * 2 + t) * p.cout + cch) = x;
                    }
                }
            }
        };
        if (fast3) {
            // ---- the common case (two or more ring slots, whole groups): a K loop without predicates.  Every wave stages one 4 KB row per
            // chunk with ONE address set-up (dma16x4_row); rows of the K padding fetch the last listed pair (their h2 operand is exactly 0).
            const unsigned voff = (unsigned)lane * 16u, lds_row = lds_ring + wave * 4096, ring_bytes = (unsigned)D * 32768u;
            unsigned foff = 0;                                  // slot offset of the next chunk to issue
            int fg = 0, fks = 0;
            int kch_n = s_kidx[min(2 * wave, Kb - 2)];          // channel of this wave's k-pair row of the next chunk to issue (read one step ahead)
            const void* frow = nullptr; unsigned flds = 0;      // scalar source row / LDS row of the chunk being issued
            auto fprepare = [&]() {
                const int kch = __builtin_amdgcn_readfirstlane(kch_n);
                frow = uniform_ptr(p.w3p + ((size_t)(kch >> 1) * p.cout + 512 * min(fg, ngroups - 1)) * 8);
                flds = __builtin_amdgcn_readfirstlane(lds_row + foff);
                foff += 32768; if (foff == ring_bytes) foff = 0;
                if (++fks == nk16) { fks = 0; ++fg; }
                kch_n = s_kidx[min(16 * fks + 2 * wave, Kb - 2)];
            };
            auto fissue = [&]() { fprepare(); dma16x4_row(voff, frow, flds); };
            for (int c = 0; c < D; ++c) fissue();             // ALL slots: the fragments of a chunk are in registers one step before its MFMAs
            // B fragment offsets of this lane within a K slice for the two K16 steps (hi; lo = the neighbouring 16-byte slot)
            unsigned bo[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bo[kk] = (unsigned)prow * 128u + (((2u * (2u * kk + h)) ^ rx) << 4);
            const unsigned a_rd = a_lane + 32 * q * 8;
            // Software pipeline over the K16 steps: while the MFMAs of step c run, the fragments of step c + 1 are read (its chunk landed
            // before the barrier of this step) and the DMA of chunk c + D is issued into the slot step c's fragments were read from one
            // step earlier.  Two register sets, even / odd steps of a K slice.
            // (a Frag holds the RAW k-pair entries: they are rearranged into MFMA operands at the start of the step that multiplies them --
            // rearranged right after the reads, the wave would wait for the reads before it issues the previous step's MFMAs)
            struct Frag { bf16x8 bh, bl; u32x2 e[4][4]; };
            auto load_frag = [&](Frag& f, const unsigned char* hs, unsigned bofs) {
                const unsigned char* wa = smem + G.ring_off + coff + a_rd;
                coff += 32768; if ((unsigned)coff == ring_bytes) coff = 0;
                if (!active) return;
#if SM_ABLATE & 2
                return;
#endif
                f.bh = *reinterpret_cast<const bf16x8*>(hs + bofs);
                f.bl = *reinterpret_cast<const bf16x8*>(hs + (bofs ^ 16u));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) f.e[i][qq] = *reinterpret_cast<const u32x2*>(wa + qq * 4096 + i * 1024);
            };
            auto mfma_frag = [&](const Frag& f) {
                if (!active) return;
                bf16x8 ah[4], al[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 ahu = {f.e[i][0][0], f.e[i][1][0], f.e[i][2][0], f.e[i][3][0]};
                    const u32x4 alu = {f.e[i][0][1], f.e[i][1][1], f.e[i][2][1], f.e[i][3][1]};
                    ah[i] = __builtin_bit_cast(bf16x8, ahu); al[i] = __builtin_bit_cast(bf16x8, alu);
                }
#if !(SM_ABLATE & 1)
                // per accumulator the order of the three products is the usual one (lo*hi, hi*lo, hi*hi); the four accumulators are
                // interleaved so that no MFMA waits for the one issued just before it
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], f.bh, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], f.bl, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], f.bh, acc[i], 0, 0, 0);
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(al[i]), "v"(ah[i]), "v"(f.bh), "v"(f.bl));
#endif
            };
            // (the four DMA instructions as ONE burst right after the barrier: spread between the MFMAs of the step they measured 10 % slower)
            auto sync_issue = [&]() {          // the next chunk has landed for every wave; every wave has read the current one
                ST(tb)
                if (D == 2) wait_vm_n<0>(); else wait_vm_n<4>();
                ST(ta)
                ST_ADD(x3[0], tb, ta)
                lds_barrier();
                ST(tb)
                ST_ADD(x3[1], ta, tb)
#if !(SM_ABLATE & 4)
                fissue();
#else
                fprepare();
#endif
                ST(ta)
                ST_ADD(x3[2], tb, ta)
            };
            Frag f0, f1;
            if (D == 2) wait_vm_n<4>(); else wait_vm_n<8>();   // chunk 0
            lds_barrier();
            load_frag(f0, s_act, bo[0]);
#pragma unroll 1
            for (int g = 0; g < ngroups; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
                for (int sl2 = 0; sl2 < nsub; ++sl2) {
                    const unsigned char* hs = s_act + (size_t)sl2 * HW * 128;
                    const unsigned char* hn = sl2 + 1 < nsub ? hs + (size_t)HW * 128 : s_act;      // the next step's slice (next group: slice 0)
                    sync_issue();
                    load_frag(f1, hs, bo[1]);
                    mfma_frag(f0);
                    sync_issue();
                    load_frag(f0, hn, bo[0]);
                    mfma_frag(f1);
                }
                if (active) group_epilogue(g);
            }
        } else {
        int c = 0;
#pragma unroll 1
        for (int g = 0; g < ngroups; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
            for (int ks = 0; ks < nk16; ++ks, ++c) {
                if (D == 1) { lds_barrier(); issue(); wait_vm_n<0>(); }
                else wait_vm_rt(4 * (D - 2));
                lds_barrier();
                const unsigned char* ws = smem + G.ring_off + coff;
                coff += 32768; if (coff == D * 32768) coff = 0;
                if (!active) { if (D > 1) issue(); continue; }
                if (D > 1) issue_begin();
                const unsigned char* hrow = s_act + (size_t)(ks >> 1) * HW * 128 + prow * 128;
                const unsigned sl = 2u * (2u * (ks & 1) + h);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hrow + ((sl ^ rx) << 4));
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hrow + (((sl + 1) ^ rx) << 4));
                u32x2 e[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) e[i][qq] = *reinterpret_cast<const u32x2*>(ws + a_lane + qq * 4096 + 32 * (q + 4 * i) * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (512 * g + 32 * (q + 4 * i) < p.cout) {
                        const u32x4 ahu = {e[i][0][0], e[i][1][0], e[i][2][0], e[i][3][0]};
                        const u32x4 alu = {e[i][0][1], e[i][1][1], e[i][2][1], e[i][3][1]};
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
                    }
                    if (D > 1) issue_part(i);
                }
                if (D > 1) issue_advance();
            }
            if (active) group_epilogue(g);
        }
        }
        wait_vm_n<0>();      // no LDS-DMA may be in flight when the workgroup's LDS is released
#ifdef LDN_TRACE
        ST(ts[5])
        if (g_small_trace && lane == 0) {
            unsigned long long* r = g_small_trace + ((size_t)b * 8 + wave) * 16;
            for (int i = 0; i < 6; ++i) r[i] = ts[i];
            for (int i = 0; i < 3; ++i) { r[6 + i] = wt[i]; r[9 + i] = ct[i]; }
            r[12] = nsub; r[13] = x3[0]; r[14] = x3[1]; r[15] = x3[2];
        }
#endif
        if (p.colsum && !active) {   // an empty pixel tile (maps of at most 32 pixels): its partial is zero
            for (int cch = 128 * q + lane * 2; cch < p.cout; cch += 512) {
                p.colsum[((size_t)b * 2 + t) * p.cout + cch] = 0.f;
                p.colsum[((size_t)b * 2 + t) * p.cout + cch + 1] = 0.f;
            }
        }
    }
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_small)

static bool small_fits(int HW, int cin, int width, int cout) {
    if (HW < 1 || HW > 64 || width < 64 || width > 512 || width % 64 || cin < 32 || cin % 32 || cout < 128 || cout % 128) return false;
    const int nsub = width / 32;
    const SmallGeom g = small_geom(HW, nsub);
    if (S_ACT_OFF + g.act_bytes > S_LDS - S_TAB1_BYTES) return false;          // h1 next to conv1's epilogue tables
    if (g.space < 18 * width * 4) return false;                                // conv2's epilogue tables in the dead ring
    if (g.d2 < 1 || g.d3 < 1) return false;                                    // one slot of each weight ring
    if (2 * (64 + width) * 128 > S_LDS - S_ACT_OFF) return false;              // two slots of conv1's ring
    return true;
}

}  // namespace ldn

using namespace ldn;

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_small_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_small_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int ldn_bottleneck_smallmap_fits(int H, int Wd, int cin, int width, int cout) {
    if (H < 1 || Wd < 1) return 0;
    return small_fits(H * Wd, cin, width, cout) ? 1 : 0;
}

extern "C" int ldn_bottleneck_smallmap(const float* x, int ldx, int B, int H, int Wd, int cin, int width, const void* w1_split,
                                       const void* w2_pairs, const void* w3_pairs, int cout, const int32_t* ch_idx,
                                       const int32_t* ch_cnt, const float* scale1, const float* shift1, const float* post_sub1,
                                       const float* scale2, const float* shift2_tab, const float* post_sub2, const float* shift3,
                                       const float* residual, int ldr, float* out, int ldo, float* colsum, void* stream) {
    LDN_REQUIRE(x && w1_split && w2_pairs && w3_pairs && ch_idx && ch_cnt && scale1 && shift1 && post_sub1 && scale2 && shift2_tab &&
                post_sub2 && shift3 && out, "ldn_bottleneck_smallmap: null pointer");
    LDN_REQUIRE(B > 0 && H > 0 && Wd > 0, "ldn_bottleneck_smallmap: bad geometry");
    LDN_REQUIRE(small_fits(H * Wd, cin, width, cout),
                "ldn_bottleneck_smallmap: a %dx%d map with cin %d, width %d, cout %d does not fit the workgroup (ldn_bottleneck_smallmap_fits == 0)",
                H, Wd, cin, width, cout);
    LDN_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldo >= cout && ldo % 4 == 0 && (!residual || (ldr >= cout && ldr % 4 == 0)),
                "ldn_bottleneck_smallmap: bad ldx / ldo / ldr");
    LDN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w1_split % 16 == 0 && (uintptr_t)w2_pairs % 16 == 0 && (uintptr_t)w3_pairs % 16 == 0 &&
                (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 && (uintptr_t)shift3 % 16 == 0 && (uintptr_t)colsum % 16 == 0,
                "ldn_bottleneck_smallmap: pointers must be 16-byte aligned");
    SmallArgs a{};
    a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.Wd = Wd; a.HW = H * Wd; a.cin = cin; a.W = width; a.cout = cout;
    a.w1s = static_cast<const unsigned char*>(w1_split);
    a.w2p = static_cast<const unsigned char*>(w2_pairs);
    a.w3p = static_cast<const unsigned char*>(w3_pairs);
    a.k_idx = ch_idx; a.k_cnt = ch_cnt;
    a.sc1 = scale1; a.sh1 = shift1; a.ps1 = post_sub1; a.sc2 = scale2; a.sh2 = shift2_tab; a.ps2 = post_sub2; a.sh3 = shift3;
    a.residual = residual; a.ldr = ldr; a.out = out; a.ldo = ldo; a.colsum = colsum;
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_smallmap), S_LDS), "k_smallmap: cannot reserve %d B of LDS", S_LDS);
    hipLaunchKernelGGL(k_smallmap, dim3((unsigned)B), dim3(512), S_LDS, static_cast<hipStream_t>(stream), a);
    LDN_CHECK_LAUNCH("k_smallmap");
    return LDN_OK;
}
