// k_tail -- the fused tail of a channel-mode bottleneck (gfx950 / CDNA4, bf16x3 arithmetic):
//
//     conv2 (3x3, per-image input AND output channel subsets) -> bn2 + ReLU (+ the pre-BN mask constants of the channel
//     algebra) -> conv3 (1x1, per-image input subset, all output channels) -> bn3 + residual + ReLU [+ fused GAP partials]
//
// in ONE launch (reference: imagenet_classification/models/laud_resnet.py:123-144 restricted to the active channels).
// What it removes relative to the two launches it replaces (ldn_conv_image 3x3 + 1x1):
//   * the 3x3 no longer re-stages its input map once per tap: a K slice of h1 (32 packed channels of every pixel of the
//     block's halo'd input region) is DMA'd into LDS ONCE and stays there for all nine taps -- the tap shift is a per-lane LDS
//     row address, out-of-image taps read a zero row;
//   * h2 never exists in memory: the 3x3's accumulators are turned into the next GEMM's operand IN REGISTERS.  Everything is
//     computed TRANSPOSED (out^T[channel][pixel] = W^T x act^T: MFMA A operand = weights, B operand = activations, lane =
//     pixel), so the C layout of conv2 (lane = pixel, registers = channels) IS the B layout of conv3 up to a fixed permutation
//     of the K order, which is applied to the weight side for free (the K order of a gathered GEMM is a list);
//   * no operand is split in the K loops: h1 arrives pre-split from conv1's epilogue ([pixel][octet][8 hi | 8 lo] bf16), the
//     weights are pre-split once per module into "pair-interleaved" k-major layouts whose gather unit (an aligned channel
//     PAIR, channel_dyn_granularity % 2 == 0) is one 16-byte DMA piece, so that a weight fragment is four ds_read_b64 and no VALU.
// One 512-thread workgroup = 8 waves, wave w owns the 32 output pixels [32w, 32w+32) of the block (<= 256 pixels: whole
// 14x14 images, 7 rows of a 28x28 map, 4 rows of a 56x56 map) for ALL channels; there are no producer waves: every wave
// issues its share of the LDS-DMA (inline asm, counted s_waitcnt vmcnt(N): the DMA of chunk c+2 and of the next h1 slice fly
// across the barriers of chunks c and c+1).
#include "ldn_common.h"
#include "ldn_mlp.h"
#include <type_traits>
#include <stdlib.h>

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TailArgs {
    const unsigned char* h1; long h1_row_bytes;       // pre-split h1: row of pixel q at h1 + q * h1_row_bytes
    int B, Hi, Wi, Ho, Wo, W, cout;                   // stride 1: Hi == Ho, Wi == Wo
    const unsigned char* w2p;                         // [9][W/2][W/2] pieces of 16 B (see ldn_hip.h)
    const unsigned char* w3p;                         // [W/2][cout] entries of 8 B
    const int32_t* k_idx; const int32_t* k_cnt;       // [B][W], [B]
    const float* sc2; const float* sh2; const float* ps2;   // [W], [16][W], [W]
    const float* sh3;                                 // [cout]
    const float* residual; int ldr; float* out; int ldo;
    float* colsum;                                    // optional [B][mblocks*8][cout]
    int rows_per_blk, mblocks;                        // output rows per workgroup, workgroups per image
    int slice_bytes;                                  // bytes of one h1 slice slot (multiple of 1024)
    // PROJ (k_tail<2, 1, true>): the block's projection shortcut (laud_resnet.py:138-141, stride 1) as 64 more K values of conv3 --
    // out = W3' h2 + Wd' x + (shift3 + shift_d): pxs = the block INPUT pre-split by conv1's launch (32-pixel tiles,
    // ldn_bottleneck_head_split), pw = Wd with its BN scale folded in, in W3's pair layout [cin / 2][cout] x 8 B.  No residual tensor is read.
    const unsigned char* pxs; const unsigned char* pw;
};
constexpr int T_PROJ_CIN = 64;            // input channels of a folded projection (stage 1 of the ResNets: the stem's 64 channels)

__device__ __attribute__((aligned(16))) float g_tail_zero[4] = {0.f, 0.f, 0.f, 0.f};

#ifndef LDN_TAIL_PRIO
#define LDN_TAIL_PRIO 0     // tuning: 1 = s_setprio(1) around the MFMA section of a conv2 chunk, 2 = s_setprio(2) around its DMA issue
#endif
#ifdef LDN_TRACE   // tuning only: per-workgroup phase timestamps of every wave (tools/trace_tail.py)
__device__ unsigned long long* g_tail_trace = nullptr;
__device__ unsigned long long* g_chain_trace = nullptr;   // k_chain: [B][4] = masker, conv1, conv2+conv3, fences (cycles summed over the run)
__device__ unsigned long long* g_head_trace = nullptr;    // k_head: [workgroup][8 waves][8] = vmcnt wait, barrier, DMA issue, x split, MFMA, total, chunks
#define TT(x) x = __builtin_amdgcn_s_memtime();
#define TT_ADD(acc, a, b) acc += (b) - (a);
#else
#define TT(x)
#define TT_ADD(acc, a, b)
#endif

#ifndef LDN_TAIL_ABLATE
#define LDN_TAIL_ABLATE 0   // tuning only (results are wrong; tools/trace_chain.py / trace_tail.py with -DLDN_TRACE -DLDN_MASK_HASH=607): 1 = no LDS-DMA at all, 4 = no MFMA (operands kept alive)
#endif
#if LDN_TAIL_ABLATE & 4
template <typename A, typename B, typename C> __device__ __forceinline__ C t_mfma_bf16(A a, B b, C c) { asm volatile("" ::"v"(a), "v"(b)); return c; }
#else
#define t_mfma_bf16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// LDS-DMA of 16 bytes per lane: LDS destination = lds_base (wave-uniform byte address) + lane * 16, source per lane.
// Inline asm: the compiler neither counts it nor waits for it (cdna_hip_programming.md 5.7) -- every wait is explicit below.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
#if LDN_TAIL_ABLATE & 1
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// L2 prefetch (round 5): 4 bytes per lane through the LDS-DMA path into a scratch word -- no VGPR result to keep alive, counted in vmcnt like
// every other DMA instruction.  What it buys is the LINE in the XCD's L2 ahead of the 16-byte DMA that will fetch it for real.
__device__ __forceinline__ void dma4_touch(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N == 0 || N == 1 || N == 2 || N == 4, "add the immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}
__device__ __forceinline__ void lds_barrier() {   // LDS traffic of this wave retired, then the workgroup barrier (no vmcnt)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned lds_off(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}

__device__ __forceinline__ void split2(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// One K16 step of a 32 x 32 tile.  bf16x3: three v_mfma_f32_32x32x16_bf16 of the hi / lo halves (AH_ / AL_ = 8 hi | 8 lo of the A rows,
// BH_ / BL_ likewise).  F32 (true-fp32 arithmetic, the library's `fp32` math mode): an element takes the same 4 bytes as a bf16 hi + lo
// pair, so every pre-split layout of this file read as plain floats IS the fp32 layout -- AH_ / AL_ then hold the lane's k-slots 0-3 /
// 4-7 as raw floats (BH_ / BL_ likewise) and the step is eight v_mfma_f32_32x32x2_f32 (instruction i pairs k-slot i of lane half 0 with
// k-slot i of lane half 1 on both operands).
#define LDN_K16(F32_, ACC_, AH_, AL_, BH_, BL_) \
    if constexpr (F32_) { \
        const f32x4 ka0_ = __builtin_bit_cast(f32x4, AH_), ka1_ = __builtin_bit_cast(f32x4, AL_); \
        const f32x4 kb0_ = __builtin_bit_cast(f32x4, BH_), kb1_ = __builtin_bit_cast(f32x4, BL_); \
        _Pragma("unroll") for (int ki_ = 0; ki_ < 4; ++ki_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ka0_[ki_], kb0_[ki_], ACC_, 0, 0, 0); \
        _Pragma("unroll") for (int ki_ = 0; ki_ < 4; ++ki_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ka1_[ki_], kb1_[ki_], ACC_, 0, 0, 0); \
    } else { \
        ACC_ = t_mfma_bf16(AL_, BH_, ACC_); \
        ACC_ = t_mfma_bf16(AH_, BL_, ACC_); \
        ACC_ = t_mfma_bf16(AH_, BH_, ACC_); \
    }

constexpr int T_KIDX_BYTES = 1280;        // int[W + 32] channel list (W <= 256)
constexpr int T_W2_SLOTS = 3;

// The MFMA section of one conv2 chunk (tap TR_'s h1 rows of slice slot HS_ x the staged W2 tile WS_): B fragment = h1 row of the tap
// (per-lane LDS address), A = staged W2 rows; per n-subtile two K16 steps, the weight fragment of the second requested before the
// MFMAs of the first (the schedule is pinned: left alone, hipcc hoists every fragment read of the chunk and spills).
#define LDN_TAIL_CHUNK_MFMA(WS_, HS_, TR_) {                                                                  \
            const unsigned char* ws = (WS_); \
            const unsigned rbase = (unsigned)(TR_) * 128u, rx = ((unsigned)(TR_) >> 1) & 7u; \
            bf16x8 bh[2], bl[2]; \
    _Pragma("unroll") \
            for (int half = 0; half < 2; ++half) { \
                const unsigned sl = 2u * (2u * half + h); \
                bh[half] = *reinterpret_cast<const bf16x8*>((HS_) + rbase + ((sl ^ rx) << 4)); \
                bl[half] = *reinterpret_cast<const bf16x8*>((HS_) + rbase + (((sl + 1) ^ rx) << 4)); \
            } \
    _Pragma("unroll") \
            for (int j = 0; j < NS; ++j) { \
                if (j < nsub) { \
                    u32x2 e0[4], e1[4]; \
    _Pragma("unroll") \
                    for (int q = 0; q < 4; ++q) e0[q] = *reinterpret_cast<const u32x2*>(ws + a_lane + q * W2_ROW + j * 256); \
    _Pragma("unroll") \
                    for (int q = 0; q < 4; ++q) e1[q] = *reinterpret_cast<const u32x2*>(ws + a_lane + (8 + q) * W2_ROW + j * 256); \
                    /* bf16x3: an entry = {hi k0, hi k1 | lo k0, lo k1}: dword 0 of the four entries = the 8 hi halves, dword 1 = the 8 lo halves. */ \
                    /* F32: an entry = {w[k0], w[k1]} as floats: entries 0, 1 = k-slots 0-3, entries 2, 3 = k-slots 4-7.                         */ \
                    { \
                        const u32x4 ahu = F32 ? u32x4{e0[0][0], e0[0][1], e0[1][0], e0[1][1]} : u32x4{e0[0][0], e0[1][0], e0[2][0], e0[3][0]}; \
                        const u32x4 alu = F32 ? u32x4{e0[2][0], e0[2][1], e0[3][0], e0[3][1]} : u32x4{e0[0][1], e0[1][1], e0[2][1], e0[3][1]}; \
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu); \
                        LDN_K16(F32, acc[j], ah, al, bh[0], bl[0]) \
                    } \
                    { \
                        const u32x4 ahu = F32 ? u32x4{e1[0][0], e1[0][1], e1[1][0], e1[1][1]} : u32x4{e1[0][0], e1[1][0], e1[2][0], e1[3][0]}; \
                        const u32x4 alu = F32 ? u32x4{e1[2][0], e1[2][1], e1[3][0], e1[3][1]} : u32x4{e1[0][1], e1[1][1], e1[2][1], e1[3][1]}; \
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu); \
                        LDN_K16(F32, acc[j], ah, al, bh[1], bl[1]) \
                    } \
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); \
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); \
                } \
            } \
}

// NS = W / 32 (maximum n-subtiles / K slices of an image): 2, 4 or 8;  ST = stride of the 3x3 (1 or 2: the first block of a stage,
// laud_resnet.py:123 with stride 2 -- the block's halo'd input region then holds 2 R + 1 input rows for R output rows)
template <int NS, int ST = 1, bool PROJ = false, bool F32 = false>
__device__ __forceinline__ void tail_body(const TailArgs& p, const int b, const int mb, unsigned char* const smem, const int tid) {
    static_assert(!PROJ || (NS == 2 && ST == 1), "the folded projection exists for the 64-wide stride-1 block (stage 1's first block)");
    static_assert(!(PROJ && F32), "the folded projection is a bf16x3 form");
    constexpr int W = NS * 32;
    // NS == 2 (stage 1: 14 short blocks per image, all of them bound by the CU's memory pipe in their conv3 phase and idle on it in
    // their conv2 phase): ONE h1 slice slot and 128 registers, so that TWO workgroups fit a CU and overlap each other's phases.
    // ST == 2: the input region is four times the output block, one slice slot is all that fits.
    // ST == 2: the input region is four times the output block -- it is staged one PARITY PLANE at a time (the taps of a stride-2 3x3
    // read one plane each: (even | odd input rows) x (even | odd input columns)), two plane slots.
    constexpr int SLICE_BUFS = ST == 2 ? 2 : (NS == 2 ? 1 : 2);
    constexpr int W2_ROW = NS * 256;                  // bytes of one k-pair row of the staged W2 tile: W entries of 8 B
    constexpr int W2_SLOT = 16 * W2_ROW;              // 16 k-pairs = one K slice of 32
    constexpr int CW = (NS == 8 || PROJ) ? 32 : 64;   // output channels per conv3 chunk (PROJ: the staged chunk also holds Wd's 32 k-pair rows)
    constexpr int NCS = CW / 32;
    constexpr int W3_ROW = CW * 8;                    // bytes of one k-pair row of the staged W3 chunk
    constexpr int RPI = 1024 / W3_ROW;                // rows per DMA instruction
    constexpr int PROWS = PROJ ? T_PROJ_CIN / 2 : 0;  // k-pair rows of the folded projection, staged behind the image's W3 rows
    constexpr int W3_SLOT = (W / 2 + PROWS) * W3_ROW;
    constexpr int NP = W;                             // table width
    int* const s_kidx = reinterpret_cast<int*>(smem);
    unsigned char* const s_h1 = smem + T_KIDX_BYTES;                      // 2 slice slots (conv2 phase)
    unsigned char* const s_w2 = s_h1 + SLICE_BUFS * p.slice_bytes;        // 3 W2 slots   (conv2 phase)
    unsigned char* const s_w3 = smem + T_KIDX_BYTES;                      // 2 W3 slots   (conv3 phase, over the above)
    float* const s_tab = reinterpret_cast<float*>(s_w3 + 2 * W3_SLOT);    // sc2[NP], ps2[NP], sh2[16][NP] (conversion)
    unsigned char* const s_scr = reinterpret_cast<unsigned char*>(s_tab + 18 * NP);   // 8 x 4 KiB transpose scratch

    int lane = tid & 63;                    // (not const: made opaque again in front of the conv3 phase, see there)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int l31 = lane & 31, h = lane >> 5;
#ifdef LDN_TRACE
    unsigned long long tr0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0, ta = 0, tb = 0, w2wait = 0, w3wait = 0, w3bar = 0, w3epi = 0, w3k = 0;
    TT(tr0)
#endif
    // ---- geometry of this workgroup's block of output rows and of its halo'd input region (stride ST, pad 1)
    const int y0 = mb * p.rows_per_blk;
    const int rows = min(p.rows_per_blk, p.Ho - y0);
    const int npix = rows * p.Wo;                              // <= 256
    const int yin0 = max(ST * y0 - 1, 0), yin1 = min(ST * (y0 + rows - 1) + 1, p.Hi - 1);
    const int NR = (yin1 - yin0 + 1) * p.Wi;                   // input pixels resident per slice (stride 1)
    const int NRp = round_up(NR, 8);
    const int ZR = ST == 2 ? p.slice_bytes / 128 - 1 : NRp;    // index of the all-zero row of each slice slot (never touched by the DMA)
    // stride 2: the four parity planes g = 2 (input row odd) + (input column odd) of the block's input region, in PLANE coordinates
    // (input pixel (iy, ix) = plane pixel (iy >> 1, ix >> 1)); rows / columns that exist in the map and are read by some tap
    int plo[4] = {0, 0, 0, 0}, pw[4] = {1, 1, 1, 1}, pn[4] = {0, 0, 0, 0}, clo[4] = {0, 0, 0, 0};
    if constexpr (ST == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int py = g >> 1, px = g & 1;
            const int rlo = py ? max(y0 - 1, 0) : y0;
            const int rhi = min(y0 + rows - 1, (p.Hi - 1 - py) >> 1);           // 2 pr + py <= Hi - 1
            const int chi = min(p.Wo - 1, (p.Wi - 1 - px) >> 1);
            plo[g] = rlo; clo[g] = 0;
            pw[g] = max(chi + 1, 1);
            pn[g] = (rhi >= rlo && chi >= 0 && p.Hi - 1 - py >= 0 && p.Wi - 1 - px >= 0) ? (rhi - rlo + 1) * (chi + 1) : 0;
        }
    }
    const long in_row0 = (long)b * p.Hi * p.Wi + (long)yin0 * p.Wi;     // first input pixel (flat) of the region
    const long out_row0 = (long)b * p.Ho * p.Wo + (long)y0 * p.Wo;      // first output pixel (flat) of the block

    const int Kb = min(p.k_cnt[b], W);
    const int nsub = ceil_div(Kb, 32);                         // n-subtiles == K slices (conv2 is square: same list)
    const int Kp = nsub * 32;
    if (tid < W + 32) s_kidx[tid] = tid < Kb ? p.k_idx[(size_t)b * W + tid] : -1;
    LDN_DCHECK(p.k_cnt[b] >= 0 && p.k_cnt[b] <= W && (p.k_cnt[b] & 1) == 0, 301);   // count within the list, whole channel pairs
    if (tid < Kb) {
        const int ch = p.k_idx[(size_t)b * W + tid];
        LDN_DCHECK(ch >= 0 && ch < W, 302);                                        // list entries are channels of this layer
        LDN_DCHECK((tid & 1) ? (ch == p.k_idx[(size_t)b * W + tid - 1] + 1) : ((ch & 1) == 0), 303);   // aligned pairs (gran % 2 == 0)
        LDN_DCHECK(tid == 0 || ch > p.k_idx[(size_t)b * W + tid - 1], 304);        // ascending
    }
    // the zero rows of the two slice slots (never touched by the DMA, which covers rows 0 .. NRp-1)
    if (tid < 32 * SLICE_BUFS) reinterpret_cast<float*>(s_h1 + (tid >> 5) * p.slice_bytes + ZR * 128)[tid & 31] = 0.f;
    __syncthreads();

    // ---- this lane's output pixel and its nine tap rows in the slice (ZR = zero row)
    const int pm = wave * 32 + l31;
    const bool pvalid = pm < npix;
    const int oy = y0 + pm / p.Wo, ox = pm % p.Wo;
    int trow[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = ST * oy + t / 3 - 1, ix = ST * ox + t % 3 - 1;
        const bool ok = pvalid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        if constexpr (ST == 2) {
            const int g = (iy & 1) * 2 + (ix & 1);
            trow[t] = ok ? ((iy >> 1) - plo[g]) * pw[g] + ((ix >> 1) - clo[g]) : ZR;   // row of the tap's plane slot
        } else {
            trow[t] = ok ? (iy - yin0) * p.Wi + ix : ZR;
        }
    }
    // border class of the pixel for the shift table of the channel algebra: (top | bottom << 1) * 4 + (left | right << 1)
    // = which tap rows / columns fall outside the INPUT map
    const int cls = (((ST * oy - 1 < 0) | ((ST * oy + 1 >= p.Hi) << 1)) * 4 + ((ST * ox - 1 < 0) | ((ST * ox + 1 >= p.Wi) << 1)));

    // ---- DMA helpers -------------------------------------------------------------------------------------------------------
    const unsigned lds_h1 = lds_off(s_h1), lds_w2 = lds_off(s_w2), lds_w3 = lds_off(s_w3);
    // h1 slice piece q (rows 8q .. 8q+7 of the region, 128 B each): lane = (row 8q + (lane >> 3), physical slot lane & 7);
    // the XOR swizzle (slot ^ ((row >> 1) & 7)) is applied to the SOURCE address (the LDS image of a DMA is lane-linear)
    auto dma_h1 = [&](int slice, int q) {
        const int r = q * 8 + (lane >> 3);
        const int lslot = (lane & 7) ^ ((r >> 1) & 7);
        const unsigned char* src = r < NR ? p.h1 + (in_row0 + r) * p.h1_row_bytes + slice * 128 + lslot * 16
                                          : reinterpret_cast<const unsigned char*>(g_tail_zero);
        dma16(src, lds_h1 + (slice % SLICE_BUFS) * p.slice_bytes + q * 1024);
    };
    // W2 chunk (slice s, tap t): 16 k-pair rows x (Kp / 2) n-pair pieces of 16 B.  Wave w stages rows 2w and 2w + 1.
    //   NS == 2: one instruction covers both rows (lanes 0-31 / 32-63); NS == 4: one instruction per row;
    //   NS == 8: two per row (n-pairs 0-63 / 64-127), the second only when the image has more than 128 active channels.
    const int n_w2 = NS == 2 ? 1 : (NS == 4 ? 2 : (Kp > 128 ? 4 : 2));       // DMA instructions per wave and chunk
    // per-lane n-pair source offsets (fixed for the whole kernel)
    long npo[NS == 8 ? 2 : 1];
#pragma unroll
    for (int e = 0; e < (NS == 8 ? 2 : 1); ++e) {
        const int v = (NS == 2 ? (lane & 31) : lane) + 64 * e;               // packed n-pair
        const int ch = 2 * v < Kb ? s_kidx[2 * v] : -1;
        npo[e] = ch >= 0 ? (long)(ch >> 1) * 16 : -1;
    }
    // stride 2: piece q (8 plane pixels) of parity plane g of K slice `slice` into plane slot `buf`
    auto dma_plane = [&](int slice, int g, int buf, int q) {
        const int r = q * 8 + (lane >> 3);
        const int lslot = (lane & 7) ^ ((r >> 1) & 7);
        const int pr = r / pw[g], pc = r - pr * pw[g];
        const long pix = (long)b * p.Hi * p.Wi + (long)(2 * (plo[g] + pr) + (g >> 1)) * p.Wi + 2 * (clo[g] + pc) + (g & 1);
        const unsigned char* src = r < pn[g] ? p.h1 + pix * p.h1_row_bytes + slice * 128 + lslot * 16
                                             : reinterpret_cast<const unsigned char*>(g_tail_zero);
        dma16(src, lds_h1 + buf * p.slice_bytes + q * 1024);
    };
    // W2 chunk c (W2 slot c % 3) = tap t of K slice s; e = 0 / 1: this wave's first / second k-pair row (the two halves of its share)
    auto dma_w2e = [&](int c, int s, int t, int e) {
        const unsigned slot = lds_w2 + (c % T_W2_SLOTS) * W2_SLOT;
        const int u = 2 * wave + (NS == 2 ? (lane >> 5) : e);            // k-pair row of the slice
        const int kch = s_kidx[32 * s + 2 * u];                          // -1 beyond the image's list
        const long rowoff = ((long)t * (W / 2) + (kch >> 1)) * (W / 2) * 16;
        if (NS == 2) {
            if (e == 1) return;
            const unsigned char* src = (kch >= 0 && npo[0] >= 0) ? p.w2p + rowoff + npo[0] : reinterpret_cast<const unsigned char*>(g_tail_zero);
            dma16(src, slot + 2 * wave * W2_ROW);
        } else if (NS == 4) {
            const unsigned char* src = (kch >= 0 && npo[0] >= 0) ? p.w2p + rowoff + npo[0] : reinterpret_cast<const unsigned char*>(g_tail_zero);
            dma16(src, slot + u * W2_ROW);
        } else {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (f == 1 && Kp <= 128) break;
                const unsigned char* src = (kch >= 0 && npo[f] >= 0) ? p.w2p + rowoff + npo[f] : reinterpret_cast<const unsigned char*>(g_tail_zero);
                dma16(src, slot + u * W2_ROW + f * 1024);
            }
        }
    };
    auto dma_w2x = [&](int c, int s, int t) { dma_w2e(c, s, t, 0); dma_w2e(c, s, t, 1); };
    // the dummies that stand in for half e of a W2 tile beyond the K range (they keep the counted wait's arithmetic constant)
    auto dma_w2_dummy = [&](int c, int e) {
        const int n = NS == 2 ? (e == 0 ? 1 : 0) : (NS == 4 ? 1 : (Kp > 128 ? 2 : 1));
        for (int i = 0; i < n; ++i) dma16(g_tail_zero, lds_w2 + (c % T_W2_SLOTS) * W2_SLOT + (2 * wave) * W2_ROW);
    };
    auto dma_w2 = [&](int c) { dma_w2x(c, c / 9, c % 9); };   // stride 1: taps in order
    auto wait_chunk = [&]() {   // everything but this wave's last n_w2 DMA instructions (= the W2 pieces of the NEXT chunk) has landed
        if (n_w2 == 1) wait_vm<1>();
        else if (n_w2 == 2) wait_vm<2>();
        else wait_vm<4>();
    };

    // ======================================================================================================== conv2 (3x3)
    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int nchunks = nsub * 9;
    const int nq = NRp / 8;                                     // DMA pieces per h1 slice (<= 72)
    const bool active = wave * 32 < npix;                       // waves beyond the block's pixels only stage and synchronise
    // fragment addressing
    const unsigned a_lane = (unsigned)(4 * h * W2_ROW + l31 * 8);          // A (weights): k-pair rows 4h .. 4h+3 of a K16 step
    if constexpr (ST == 2) {
        // ---- stride 2: per K slice the nine taps are walked PLANE BY PLANE -- (odd row, odd column): taps 0 2 6 8; (even, odd): 3 5;
        // (odd, even): 1 7; (even, even): 4 -- so that only one parity plane of the slice (a quarter of the input region) is in LDS
        // at a time: plane u + 1 is fetched into the other slot, spread over the chunks of plane u.  The sum over the taps of an
        // output is the same set of products in a different order of the K axis (fp32 accumulation: results differ from the
        // tap-ordered stride-1 form in the last bits only, like any change of the K order).
        constexpr int ORD[4] = {3, 1, 2, 0};                    // plane g of position o in the walk
        constexpr int NTP[4] = {4, 2, 2, 1};                    // taps per position
        constexpr int TAPS[4][4] = {{0, 2, 6, 8}, {3, 5, 3, 3}, {1, 7, 1, 1}, {4, 4, 4, 4}};
        auto tap_of = [&](int c) {                              // chunk c -> tap (c % 9 walks 0 2 6 8 | 3 5 | 1 7 | 4)
            constexpr int SEQ[9] = {0, 2, 6, 8, 3, 5, 1, 7, 4};
            return SEQ[c % 9];
        };
        if (nchunks > 0) {
            for (int q = wave; q < (pn[3] + 7) / 8; q += 8) dma_plane(0, 3, 0, q);   // plane (slice 0, odd / odd) -> slot 0
            dma_w2x(0, 0, tap_of(0));
            dma_w2x(1, 0, tap_of(1));
        }
        TT(tr1)
        int c = 0;
        for (int s = 0; s < nsub; ++s) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int u = 4 * s + o;                        // plane counter: slot u & 1
                const unsigned char* hs = s_h1 + (u & 1) * p.slice_bytes;
                const int gn = ORD[(o + 1) & 3], sn = o == 3 ? s + 1 : s;      // the plane after this one
                const int npn = sn < nsub ? (pn[gn] + 7) / 8 : 0;              // its DMA pieces
#pragma unroll
                for (int i = 0; i < NTP[o]; ++i) {
                    TT(ta)
                    wait_chunk();
                    lds_barrier();     // chunk c and this plane are in LDS for every wave; every wave has left chunk c - 1 (and plane u - 1)
                    TT(tb)
                    TT_ADD(w2wait, ta, tb)
                    // issue: this chunk's share of the next plane first, then the W2 tile of chunk c + 2 (the counted wait relies on the order)
                    for (int q = i * 8 + wave; q < npn; q += 8 * NTP[o]) dma_plane(sn, gn, (u + 1) & 1, q);
                    if (c + 2 < nchunks) dma_w2x(c + 2, (c + 2) / 9, tap_of(c + 2));
                    else { for (int e = 0; e < n_w2; ++e) dma16(g_tail_zero, lds_w2 + ((c + 2) % T_W2_SLOTS) * W2_SLOT + (2 * wave) * W2_ROW); }
                    if (active) LDN_TAIL_CHUNK_MFMA(s_w2 + (c % T_W2_SLOTS) * W2_SLOT, hs, trow[TAPS[o][i]])
                    ++c;
                }
            }
        }
    } else {
    if (nchunks > 0) {
        for (int q = wave; q < nq; q += 8) dma_h1(0, q);        // slice 0
        dma_w2(0);
        dma_w2(1);                                              // nchunks >= 9
    }
    TT(tr1)
#ifdef LDN_TAIL_HALF_MAJOR   // (measured neutral on the headline, round 5: 12.15-12.31 vs 12.25 ms; kept as a switch)
    // Round 5: the chunk body in HALF-MAJOR order (K16 half 0 of every n-subtile, then half 1 -- per accumulator the same order of products)
    // with the B fragment of the NEXT tap read in place: the next tap's rows are rows of the SAME resident slice, so its half-0 fragment
    // replaces this tap's as soon as the half-0 steps have issued, its half-1 fragment at the end of the chunk -- no fragment read sits
    // between the barrier and the first MFMA (once per slice, at tap 0, it still does), and the chunk's DMA instructions go out behind the
    // first MFMA steps instead of in front of them.  (k_dense2 / k_rows3, DESIGN.md 4u: what the barrier -> first-MFMA path costs.)
    bf16x8 bh[2], bl[2];
    auto load_b_half = [&](const unsigned char* hs_, int tr, int half) {
        const unsigned rbase = (unsigned)tr * 128u, rx = ((unsigned)tr >> 1) & 7u;
        const unsigned sl = 2u * (2u * half + h);
        bh[half] = *reinterpret_cast<const bf16x8*>(hs_ + rbase + ((sl ^ rx) << 4));
        bl[half] = *reinterpret_cast<const bf16x8*>(hs_ + rbase + (((sl + 1) ^ rx) << 4));
    };
    for (int s = 0; s < nsub; ++s) {
        const unsigned char* hs = s_h1 + (s % SLICE_BUFS) * p.slice_bytes;
#pragma unroll
        for (int t = 0; t < 9; ++t) {                           // chunk c = 9 s + t lives in W2 slot c % 3 == t % 3
            const int c = 9 * s + t;
            TT(ta)
            wait_chunk();
            lds_barrier();     // chunk c (and, when t == 0, slice s) is in LDS for every wave; every wave has left chunk c - 1
            if (SLICE_BUFS == 1 && t == 0 && s > 0) {   // single slice slot: every wave has left slice s - 1 -> fetch slice s now (the
                for (int q = wave; q < nq; q += 8) dma_h1(s, q);   // co-resident workgroup covers the stall)
                wait_vm<0>();
                lds_barrier();
            }
            TT(tb)
            TT_ADD(w2wait, ta, tb)
            // this chunk's DMA instructions in the order the counted wait relies on: group 0 = the next slice's share, groups 1, 2 = the two
            // halves of this wave's share of the W2 tile of chunk c + 2
            auto issue = [&](int g) {
                if (g == 0) {
                    if (SLICE_BUFS == 2 && s + 1 < nsub) { const int q = t * 8 + wave; if (q < nq) dma_h1(s + 1, q); }
                } else if (c + 2 < nchunks) {
                    dma_w2e(c + 2, (c + 2) / 9, (c + 2) % 9, g - 1);
                } else {
                    dma_w2_dummy(t + 2, g - 1);
                }
            };
            if (!active) { issue(0); issue(1); issue(2); continue; }
            if (t == 0) { load_b_half(hs, trow[0], 0); load_b_half(hs, trow[0], 1); }
            const unsigned char* ws = s_w2 + (t % T_W2_SLOTS) * W2_SLOT;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    if (j < nsub) {
                        u32x2 e[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const u32x2*>(ws + a_lane + (8 * half + q) * W2_ROW + j * 256);
                        const u32x4 ahu = F32 ? u32x4{e[0][0], e[0][1], e[1][0], e[1][1]} : u32x4{e[0][0], e[1][0], e[2][0], e[3][0]};
                        const u32x4 alu = F32 ? u32x4{e[2][0], e[2][1], e[3][0], e[3][1]} : u32x4{e[0][1], e[1][1], e[2][1], e[3][1]};
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                        LDN_K16(F32, acc[j], ah, al, bh[half], bl[half])
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, F32 ? 8 : 3, 0);
                    }
                    if (j == 0) {      // (n-subtile 0 always exists) the DMA goes out behind the first step of each half
                        if (half == 0) { issue(0); issue(1); } else issue(2);
                    }
                }
                if (t < 8) load_b_half(hs, trow[t + 1], half);      // the next tap's fragment of this half, in place
            }
        }
    }
#else
    for (int s = 0; s < nsub; ++s) {
        const unsigned char* hs = s_h1 + (s % SLICE_BUFS) * p.slice_bytes;
#pragma unroll
        for (int t = 0; t < 9; ++t) {                           // chunk c = 9 s + t lives in W2 slot c % 3 == t % 3
            const int c = 9 * s + t;
            TT(ta)
            wait_chunk();
            lds_barrier();     // chunk c (and, when t == 0, slice s) is in LDS for every wave; every wave has left chunk c - 1
            if (SLICE_BUFS == 1 && t == 0 && s > 0) {   // single slice slot: every wave has left slice s - 1 -> fetch slice s now (the
                for (int q = wave; q < nq; q += 8) dma_h1(s, q);   // co-resident workgroup covers the stall)
                wait_vm<0>();
                lds_barrier();
            }
            // issue: next slice's share first, then the W2 tile of chunk c + 2 (issue order matters for the counted wait)
            TT(tb)
            TT_ADD(w2wait, ta, tb)
#if LDN_TAIL_PRIO == 2
            __builtin_amdgcn_s_setprio(2);
#endif
            if (SLICE_BUFS == 2 && s + 1 < nsub) { const int q = t * 8 + wave; if (q < nq) dma_h1(s + 1, q); }
            if (c + 2 < nchunks) dma_w2(c + 2);
            else { for (int e = 0; e < n_w2; ++e) dma16(g_tail_zero, lds_w2 + ((t + 2) % T_W2_SLOTS) * W2_SLOT + (2 * wave) * W2_ROW); }   // keeps the count
#if LDN_TAIL_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
            if (!active) continue;
#if LDN_TAIL_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#endif
            LDN_TAIL_CHUNK_MFMA(s_w2 + (t % T_W2_SLOTS) * W2_SLOT, hs, trow[t])
#if LDN_TAIL_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#endif
        }
    }
#endif
    }   // ST == 1
    wait_vm<0>();
    lds_barrier();         // every wave is out of the conv2 loop: the slice / W2 regions are free
    TT(tr2)
    // The lane index behind an optimisation barrier: left alone, hipcc computes the conv3 phase's per-lane addresses (store rows, scratch
    // slots, fragment offsets) at the top of the kernel and keeps them alive -- or spills them -- through the whole conv2 phase
    // (k_tail<2>: 22 spilled registers at its 128-register budget = ~150 MB of scratch writes per stage-1 launch, r04_pmc_stage12.txt)
    asm volatile("" : "+v"(lane));
    l31 = lane & 31;
    h = lane >> 5;

    // ======================================================================================================== conv3 (1x1)
    // Output channels are walked in chunks of CW (64; 32 for the widest layer, whose h2 operand alone takes 128 registers and
    // whose staged weight chunk would otherwise not leave room for the transpose scratch), NCS sub-passes of 32 channels each.
    const int nchunk3 = p.cout / CW;
    // W3 chunk cc: (Kp / 2) k-pair rows x CW output channels x 8 B; one DMA instruction = RPI rows of CW * 8 bytes
    auto dma_w3 = [&](int cc) {
        const unsigned slot = lds_w3 + (cc & 1) * W3_SLOT;
        constexpr int LPR = 64 / RPI;                            // lanes (16-byte pieces = channel pairs) per row
        for (int i = wave; i < Kp / (2 * RPI); i += 8) {
            const int u = RPI * i + lane / LPR;
            const int kch = s_kidx[2 * u];
            const unsigned char* src = kch >= 0 ? p.w3p + ((long)(kch >> 1) * p.cout + cc * CW + 2 * (lane % LPR)) * 8
                                                : reinterpret_cast<const unsigned char*>(g_tail_zero);
            dma16(src, slot + i * 1024);
        }
        if constexpr (PROJ) {
            // Wd's k-pair rows behind the image's Kp / 2 rows of W3.  Staged row 8 s + r of K16 step s holds source pair 8 s + PI[r]:
            // an A fragment of lane half h reads staged rows 2 h + {0, 1, 4, 5} of the step, the B fragment (one octet of the pre-split
            // input per lane) supplies channels 16 s + 8 h + {0 .. 7} = pairs 8 s + 4 h + {0 .. 3} -- PI = {0, 1, 4, 5, 2, 3, 6, 7}
            for (int i = wave; i < PROWS / RPI; i += 8) {
                const int u = RPI * i + lane / LPR;
                const int r = u & 7;
                const int sp = (u & ~7) | ((r & 1) | ((r & 2) << 1) | ((r & 4) >> 1));
                dma16(p.pw + ((long)sp * p.cout + cc * CW + 2 * (lane % LPR)) * 8, slot + (Kp / (2 * RPI) + i) * 1024);
            }
        }
    };
    if (nchunk3 > 0) dma_w3(0);

    // ---- conversion tables (gathered through the channel list), then acc -> h2 operand fragments in registers
    // (round 6: up to four gathers per thread in flight together -- as plain loops hipcc emits LDS read -> global load -> s_waitcnt vmcnt(0) ->
    // ds_write per entry, 16 NP / 512 memory latencies in a row)
    for (int i = tid; i < NP; i += 512) {
        const int ch = i < Kb ? s_kidx[i] : -1;
        const float a = p.sc2[max(ch, 0)], c = p.ps2[max(ch, 0)];
        s_tab[i] = ch >= 0 ? a : 0.f;
        s_tab[NP + i] = ch >= 0 ? c : 0.f;
    }
    {
        constexpr int TB = (16 * NP / 512) < 4 ? (16 * NP / 512) : 4;
        for (int i0 = tid; i0 < 16 * NP; i0 += 512 * TB) {
            float v[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int i = i0 + 512 * u;                       // (16 NP is a multiple of 512 TB: no ragged pass)
                const int k = i / NP, n = i - k * NP;
                const int ch = n < Kb ? s_kidx[n] : -1;
                const float x = p.sh2[k * W + max(ch, 0)];
                v[u] = ch >= 0 ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < TB; ++u) s_tab[2 * NP + i0 + 512 * u] = v[u];
        }
    }
    lds_barrier();
    // In place: the 16 fp32 accumulators of n-subtile j become 16 dwords of bf16 pairs -- for each K16 step t of conv3
    // [8t .. 8t+3] = the 8 hi halves, [8t+4 .. 8t+7] = the 8 lo halves of the lane's h2 values (k-slot e = 4 qq + i <->
    // accumulator register 8t + 4 qq + i).  No second register array: h2 of the widest layer alone takes 128 registers.
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float v[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int n0 = 32 * j + 8 * q4 + 4 * h;             // register quad q4: rows n = 32 j + 8 q4 + 4 h + {0..3}
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + NP + n0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + 2 * NP + cls * NP + n0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[4 * q4 + e] = fmaxf(acc[j][4 * q4 + e] * sc[e] + sh[e], 0.f) - ps[e];   // columns without a channel: 0 * 0 + 0
        }
        if constexpr (F32) {
            // true fp32: h2 stays fp32 in the accumulator registers -- k-slot e of conv3's K16 step t is register 8 t + e, as in the split form
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = v[r];
        } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < 4; ++d) {                       // dword d of the step = k-slots 2d, 2d + 1 = registers 8t + 2d, 8t + 2d + 1
                const float x0 = v[8 * t + 2 * d], x1 = v[8 * t + 2 * d + 1];
                const bf16x2 hi = {(__bf16)x0, (__bf16)x1};
                const bf16x2 lo = {(__bf16)(x0 - (float)hi[0]), (__bf16)(x1 - (float)hi[1])};
                acc[j][8 * t + d] = __builtin_bit_cast(float, hi);
                acc[j][8 * t + 4 + d] = __builtin_bit_cast(float, lo);
            }
        }
        // compiler memory barrier + scheduling barrier: left alone, hipcc hoists the table reads of ALL subtiles to the top
        // (96 ds_read_b128 = 384 registers) and spills them
        asm volatile("" : "+v"(acc[j]) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    auto frag_hi = [&](int j, int t) -> bf16x8 {
        const f32x4 x = {acc[j][8 * t], acc[j][8 * t + 1], acc[j][8 * t + 2], acc[j][8 * t + 3]};
        return __builtin_bit_cast(bf16x8, x);
    };
    auto frag_lo = [&](int j, int t) -> bf16x8 {
        const f32x4 x = {acc[j][8 * t + 4], acc[j][8 * t + 5], acc[j][8 * t + 6], acc[j][8 * t + 7]};
        return __builtin_bit_cast(bf16x8, x);
    };

    TT(tr3)
    const int trw = lane >> 3, tc = lane & 7;                  // epilogue layout: lane = (row trw + 8 it, 4 channels at 4 tc)
    const unsigned a3_lane = (unsigned)(2 * h * W3_ROW + l31 * 8);
    float* const scr = reinterpret_cast<float*>(s_scr + wave * 4096);   // this wave's 32 x 32 transpose scratch
    // PROJ: the lane's pixel of the block input, pre-split (K16 step s = octet 2 s + h: 8 hi | 8 lo), loaded ONCE for all chunks (round 6: re-loaded per
    // 32-channel chunk -- eight times -- these 8 KB per wave were assumed L2 hits and are not: profiles/r06_pmc_stage12.txt, FETCH x 2 = 1.67 GB per launch
    // against 0.52 GB of h1 + input; the launch's own output stream evicts them)
    bf16x8 pjh[PROJ ? T_PROJ_CIN / 16 : 1], pjl[PROJ ? T_PROJ_CIN / 16 : 1];
    if constexpr (PROJ) {
        const long q = out_row0 + min(wave * 32 + l31, npix - 1);
        const unsigned char* xr = p.pxs + (q >> 5) * ((long)T_PROJ_CIN * 128) + (q & 31) * 16 + h * 1024;
#pragma unroll
        for (int s2 = 0; s2 < T_PROJ_CIN / 16; ++s2) {
            pjh[s2] = *reinterpret_cast<const bf16x8*>(xr + s2 * 2048);
            pjl[s2] = *reinterpret_cast<const bf16x8*>(xr + s2 * 2048 + 512);
        }
    }
    for (int cc = 0; cc < nchunk3; ++cc) {
        if (cc == 0) { wait_vm<0>(); lds_barrier(); }           // W3(0) is in LDS for every wave
        if (cc + 1 < nchunk3) dma_w3(cc + 1);                   // slot (cc + 1) & 1: every wave left it before the last barrier
        const unsigned char* ws = s_w3 + (cc & 1) * W3_SLOT;
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs) {
            const int c0 = cc * CW + cs * 32;
            // residual tile in the layout the epilogue stores in, requested before the K loop that hides its latency
            f32x4 res[PROJ ? 1 : 4];
            if constexpr (!PROJ) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int prow = wave * 32 + trw + 8 * it;
                const float* src = (p.residual && prow < npix) ? p.residual + (size_t)(out_row0 + prow) * p.ldr + c0 + tc * 4 : g_tail_zero;
                res[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            }
            }
            f32x4 sh = *reinterpret_cast<const f32x4*>(p.sh3 + c0 + tc * 4);   // bn3 shift: requested with the residual
            f32x16 acc3;
            TT(ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (j < nsub && active) {
                    // k-pair rows P0 + {0, 1, 4, 5}, P0 = 16 j + 8 t + 2 h: the K order in which lane (pixel, h) holds h2
                    u32x2 e[2][4];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            e[t][q] = *reinterpret_cast<const u32x2*>(ws + a3_lane + (16 * j + 8 * t + (q & 1) + 4 * (q >> 1)) * W3_ROW + cs * 256);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const u32x4 ahu = F32 ? u32x4{e[t][0][0], e[t][0][1], e[t][1][0], e[t][1][1]} : u32x4{e[t][0][0], e[t][1][0], e[t][2][0], e[t][3][0]};
                        const u32x4 alu = F32 ? u32x4{e[t][2][0], e[t][2][1], e[t][3][0], e[t][3][1]} : u32x4{e[t][0][1], e[t][1][1], e[t][2][1], e[t][3][1]};
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                        const bf16x8 hb = frag_hi(j, t), lb = frag_lo(j, t);      // F32: registers 8 t .. + 3 / 8 t + 4 .. + 7 = k-slots 0-3 / 4-7 as floats
                        LDN_K16(F32, acc3, ah, al, hb, lb)
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
            }
            if constexpr (PROJ) {
                if (active) {
#pragma unroll
                    for (int jj = 0; jj < T_PROJ_CIN / 32; ++jj) {
                        u32x2 e[2][4];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                e[t][q] = *reinterpret_cast<const u32x2*>(ws + a3_lane + (Kp / 2 + 16 * jj + 8 * t + (q & 1) + 4 * (q >> 1)) * W3_ROW + cs * 256);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const u32x4 ahu = {e[t][0][0], e[t][1][0], e[t][2][0], e[t][3][0]};
                            const u32x4 alu = {e[t][0][1], e[t][1][1], e[t][2][1], e[t][3][1]};
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                            acc3 = t_mfma_bf16(al, pjh[2 * jj + t], acc3);
                            acc3 = t_mfma_bf16(ah, pjl[2 * jj + t], acc3);
                            acc3 = t_mfma_bf16(ah, pjh[2 * jj + t], acc3);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    }
                }
            }
            TT(tb)
            TT_ADD(w3k, ta, tb)
            if (cs == NCS - 1) wait_vm<0>();   // this wave's share of W3(cc + 1) has landed (its earlier stores and the residual too);
                                               // placed BEFORE this sub-pass's stores, which then fly through the next chunk
            // gfx9 counts loads and stores in ONE vmcnt and lets them complete out of order with each other: a residual register first
            // touched after a store has been issued makes hipcc wait vmcnt(0), i.e. for that store's acknowledgement -- four
            // serialised store latencies per tile.  Touch all of them here, before the first store.
            if constexpr (PROJ) asm volatile("" : "+v"(sh));
            else asm volatile("" : "+v"(res[0]), "+v"(res[PROJ ? 0 : 1]), "+v"(res[PROJ ? 0 : 2]), "+v"(res[PROJ ? 0 : 3]), "+v"(sh));
            TT(ta)
            TT_ADD(w3wait, tb, ta)
            // C layout (lane = pixel l31, register r = channel (r & 3) + 8 (r >> 2) + 4 h) -> rows of 32 channels per pixel,
            // 16-byte slots XOR-swizzled with the pixel so that both the writes and the row reads are conflict-free
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = {acc3[4 * q4], acc3[4 * q4 + 1], acc3[4 * q4 + 2], acc3[4 * q4 + 3]};
                *reinterpret_cast<f32x4*>(scr + l31 * 32 + (((2 * q4 + h) ^ (l31 & 7)) << 2)) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = trw + 8 * it, prow = wave * 32 + row;
                f32x4 x = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tc ^ (row & 7)) << 2));
                if constexpr (PROJ) x = x + sh;
                else x = x + sh + res[PROJ ? 0 : it];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
                if (prow < npix) {
                    __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p.out + (size_t)(out_row0 + prow) * p.ldo + c0 + tc * 4));
                    csum += x;
                }
            }
            if (p.colsum) {   // fused global-average-pool partials (the next block's channel masker): one slot per (block, wave)
#pragma unroll
                for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);      // (no LDS round trips: DPP + permlane swaps, same additions)
                if (trw == 0)
                    *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * (p.mblocks * 8) + mb * 8 + wave) * p.cout + c0 + tc * 4) = csum;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            TT(tb)
            TT_ADD(w3epi, ta, tb)
        }
        TT(ta)
        lds_barrier();     // every wave has left slot cc & 1 (refilled by the DMA of chunk cc + 2 in the next iteration)
        TT(tb)
        TT_ADD(w3bar, ta, tb)
    }
#ifdef LDN_TRACE
    TT(tr4)
    if (g_tail_trace && lane == 0) {
        unsigned long long* r = g_tail_trace + (((size_t)mb * p.B + b) * 8 + wave) * 12;
        r[0] = tr0; r[1] = tr1; r[2] = tr2; r[3] = tr3; r[4] = tr4; r[5] = w2wait; r[6] = w3k; r[7] = w3wait; r[8] = w3epi; r[9] = w3bar;
        r[10] = nsub; r[11] = 0;
    }
#endif
}

template <int NS, int ST, bool PROJ = false, bool F32 = false>
__global__ __launch_bounds__(512, ((NS == 2 && ST == 1) ? 4 : 2)) void k_tail(const TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tail_body<NS, ST, PROJ, F32>(p, blockIdx.x % p.B, blockIdx.x / p.B, smem, threadIdx.x);     // image-fastest: with B % 8 == 0 image b stays on XCD b % 8
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_tail)

// pixels resident per slice slot for a block of R output rows: stride 1 -- the halo'd input region (R + 2 rows); stride 2 -- the
// largest PARITY PLANE of the region ((R + 1) plane rows x Wo plane columns: the taps read one plane each)
static int tail_region_pixels(int R, int Hi, int Wi, int st) {
    if (st == 2) return min(R + 1, (Hi + 1) / 2) * ((Wi - 1) / 2 + 1);
    return min(R + 2, Hi) * Wi;
}

// bytes of LDS of the conv2 phase for blocks of R output rows (NS = width / 32)
static size_t tail_lds2(int R, int Hi, int Wi, int NS, int st) {
    const int nr = tail_region_pixels(R, Hi, Wi, st);
    const size_t slice = (size_t)round_up((round_up(nr, 8) + 1) * 128, 1024);
    return (size_t)T_KIDX_BYTES + (st == 2 ? 2 : (NS == 2 ? 1 : 2)) * slice + (size_t)T_W2_SLOTS * 16 * NS * 256;
}

// output rows per workgroup: as many as 256 pixels, the 160 KiB of LDS and the 72-piece slice pipeline allow; 0 = the map does not fit
static int tail_rows_per_block(int Hi, int Wi, int NS, int st, int* mblocks) {
    const int Ho = (Hi - 1) / st + 1, Wo = (Wi - 1) / st + 1;
    int R = 256 / Wo;
    if (R < 1) return 0;
    if (R > Ho) R = Ho;
    // (stride 1, two slots: a slice is prefetched one piece per wave and tap = at most 72 pieces; the other forms have no such limit)
    auto fits = [&](int r) {
        return tail_lds2(r, Hi, Wi, NS, st) <= 160 * 1024 && (st == 2 || NS == 2 || round_up(tail_region_pixels(r, Hi, Wi, st), 8) / 8 <= 72);
    };
    while (R > 1 && !fits(R)) --R;
    if (!fits(R)) return 0;
    const int mbk = ceil_div(Ho, R);
    *mblocks = mbk;
    // Rows per workgroup among the sizes that need mbk workgroups: the one with the fewest 32-pixel wave tiles over the image (the balanced
    // split first on ties).  A 28x28 map as 4 x 7 rows is 4 x 7 waves with a nearly empty last one (196 = 6.1 tiles); as 8 + 8 + 8 + 4
    // rows it is 7 + 7 + 7 + 4 full ones: 25 instead of 28 wave tiles of work for the same pixels (round 4).
    int best = ceil_div(Ho, mbk), best_waves = 1 << 30;
    for (int r = ceil_div(Ho, mbk); r <= R; ++r) {
        if (r * (mbk - 1) >= Ho) break;                 // the last workgroup would be empty
        int waves = 0;
        for (int i = 0; i < mbk; ++i) waves += ceil_div(min(r, Ho - r * i) * Wo, 32);
        if (waves < best_waves) { best_waves = waves; best = r; }
    }
    return best;
}

template <int NS, int ST, bool PROJ = false, bool F32 = false>
static int launch_tail(TailArgs& a, hipStream_t st) {
    constexpr int W = NS * 32;
    const int R = a.rows_per_blk;
    const int nr = tail_region_pixels(R, a.Hi, a.Wi, ST);
    a.slice_bytes = round_up((round_up(nr, 8) + 1) * 128, 1024);
    const size_t lds2 = tail_lds2(R, a.Hi, a.Wi, NS, ST);
    const size_t lds3 = (size_t)T_KIDX_BYTES + 2 * (size_t)(W / 2 + (PROJ ? T_PROJ_CIN / 2 : 0)) * ((NS == 8 || PROJ) ? 32 : 64) * 8 + (size_t)18 * W * 4 + 8 * 4096;
    const size_t lds = lds2 > lds3 ? lds2 : lds3;
    LDN_REQUIRE(lds <= 160 * 1024, "ldn_bottleneck_tail: %zu B of LDS exceed 160 KiB (map %dx%d, width %d)", lds, a.Ho, a.Wo, W);
    LDN_REQUIRE(ST == 2 || NS == 2 || round_up(nr, 8) / 8 <= 72, "ldn_bottleneck_tail: input region of %d pixels too large for the slice pipeline", nr);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_tail<NS, ST, PROJ, F32>), lds), "k_tail: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL((k_tail<NS, ST, PROJ, F32>), dim3((unsigned)a.B * a.mblocks), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_tail");
    return LDN_OK;
}

}  // namespace ldn

using namespace ldn;

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_chain_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
extern "C" int ldn_debug_set_ld_trace(void* buf);
extern "C" int ldn_debug_set_head_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_head_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
extern "C" int ldn_debug_set_tail_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tail_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int ldn_bottleneck_tail_splits(int H, int Wd, int width, int stride) {
    int mbk = 0;
    if (H < 1 || Wd < 1 || (stride != 1 && stride != 2) || (Wd - 1) / stride + 1 > 256 || (width != 64 && width != 128 && width != 256)) return 0;
    if (tail_rows_per_block(H, Wd, width / 32, stride, &mbk) == 0) return 0;
    return mbk * 8;
}

static int bottleneck_tail_impl(const void* h1_split, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs,
                                const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                                const float* scale2, const float* shift2_tab, const float* post_sub2,
                                const float* shift3, const float* residual, int ldr, float* out, int ldo,
                                float* colsum, const void* x_split, const void* wd_pairs, bool f32, void* stream);

extern "C" int ldn_bottleneck_tail(const void* h1_split, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs,
                                   const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                                   const float* scale2, const float* shift2_tab, const float* post_sub2,
                                   const float* shift3, const float* residual, int ldr, float* out, int ldo,
                                   float* colsum, void* stream) {
    return bottleneck_tail_impl(h1_split, ldh, B, H, Wd, stride, width, w2_pairs, w3_pairs, cout, ch_idx, ch_cnt, scale2, shift2_tab,
                                post_sub2, shift3, residual, ldr, out, ldo, colsum, nullptr, nullptr, false, stream);
}

/* true-fp32 arithmetic (v_mfma_f32_32x32x2_f32): the same kernel, operands in the fp32 twins of the pre-split layouts (include/ldn_hip.h) */
extern "C" int ldn_bottleneck_tail_f32(const void* h1, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs_f32,
                                       const void* w3_pairs_f32, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                                       const float* scale2, const float* shift2_tab, const float* post_sub2,
                                       const float* shift3, const float* residual, int ldr, float* out, int ldo,
                                       float* colsum, void* stream) {
    return bottleneck_tail_impl(h1, ldh, B, H, Wd, stride, width, w2_pairs_f32, w3_pairs_f32, cout, ch_idx, ch_cnt, scale2, shift2_tab,
                                post_sub2, shift3, residual, ldr, out, ldo, colsum, nullptr, nullptr, true, stream);
}

extern "C" int ldn_bottleneck_tail_proj_fits(int H, int Wd, int width, int cin) {
    int mbk = 0;
    if (H < 1 || Wd < 1 || Wd > 256 || width != 64 || cin != ldn::T_PROJ_CIN) return 0;
    return ldn::tail_rows_per_block(H, Wd, 2, 1, &mbk) > 0 ? 1 : 0;
}

extern "C" int ldn_bottleneck_tail_proj(const void* h1_split, int ldh, int B, int H, int Wd, int width, const void* w2_pairs,
                                        const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                                        const float* scale2, const float* shift2_tab, const float* post_sub2, const float* shift3d,
                                        const void* x_split, int cin, const void* wd_pairs, float* out, int ldo,
                                        float* colsum, void* stream) {
    LDN_REQUIRE(x_split && wd_pairs, "ldn_bottleneck_tail_proj: null pointer");
    LDN_REQUIRE(width == 64 && cin == ldn::T_PROJ_CIN, "ldn_bottleneck_tail_proj: the folded projection exists for width 64, cin 64 (got %d, %d; ldn_bottleneck_tail_proj_fits == 0)", width, cin);
    LDN_REQUIRE((uintptr_t)x_split % 16 == 0 && (uintptr_t)wd_pairs % 16 == 0, "ldn_bottleneck_tail_proj: x_split / wd_pairs must be 16-byte aligned");
    return bottleneck_tail_impl(h1_split, ldh, B, H, Wd, 1, width, w2_pairs, w3_pairs, cout, ch_idx, ch_cnt, scale2, shift2_tab,
                                post_sub2, shift3d, nullptr, 0, out, ldo, colsum, x_split, wd_pairs, false, stream);
}

static int bottleneck_tail_impl(const void* h1_split, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs,
                                const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                                const float* scale2, const float* shift2_tab, const float* post_sub2,
                                const float* shift3, const float* residual, int ldr, float* out, int ldo,
                                float* colsum, const void* x_split, const void* wd_pairs, bool f32, void* stream) {
    LDN_REQUIRE(h1_split && w2_pairs && w3_pairs && ch_idx && ch_cnt && scale2 && shift2_tab && post_sub2 && shift3 && out,
                "ldn_bottleneck_tail: null pointer");
    LDN_REQUIRE(width == 64 || width == 128 || width == 256, "ldn_bottleneck_tail: width must be 64, 128 or 256 (got %d)", width);
    LDN_REQUIRE(stride == 1 || stride == 2, "ldn_bottleneck_tail: stride must be 1 or 2 (got %d)", stride);
    LDN_REQUIRE(B > 0 && H > 0 && Wd > 0 && (Wd - 1) / stride + 1 <= 256, "ldn_bottleneck_tail: bad geometry (output map width must be <= 256)");
    LDN_REQUIRE(cout > 0 && cout % 64 == 0, "ldn_bottleneck_tail: cout must be a multiple of 64 (got %d)", cout);
    LDN_REQUIRE(ldh >= width && ldh % 8 == 0, "ldn_bottleneck_tail: ldh must be a multiple of 8 and >= width");
    LDN_REQUIRE(ldo >= cout && ldo % 4 == 0 && (!residual || (ldr >= cout && ldr % 4 == 0)), "ldn_bottleneck_tail: bad ldo / ldr");
    LDN_REQUIRE((uintptr_t)h1_split % 16 == 0 && (uintptr_t)w2_pairs % 16 == 0 && (uintptr_t)w3_pairs % 16 == 0 &&
                (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 && (uintptr_t)shift3 % 16 == 0 && (uintptr_t)colsum % 16 == 0,
                "ldn_bottleneck_tail: pointers must be 16-byte aligned");
    LDN_REQUIRE((long)9 * (width / 2) * (width / 2) * 16 < (1L << 31), "ldn_bottleneck_tail: weights too large");
    TailArgs a{};
    a.h1 = static_cast<const unsigned char*>(h1_split);
    a.h1_row_bytes = (long)ldh * 4;
    a.B = B; a.Hi = H; a.Wi = Wd; a.Ho = (H - 1) / stride + 1; a.Wo = (Wd - 1) / stride + 1; a.W = width; a.cout = cout;
    a.w2p = static_cast<const unsigned char*>(w2_pairs);
    a.w3p = static_cast<const unsigned char*>(w3_pairs);
    a.k_idx = ch_idx; a.k_cnt = ch_cnt;
    a.sc2 = scale2; a.sh2 = shift2_tab; a.ps2 = post_sub2; a.sh3 = shift3;
    a.residual = residual; a.ldr = ldr; a.out = out; a.ldo = ldo; a.colsum = colsum;
    a.pxs = static_cast<const unsigned char*>(x_split); a.pw = static_cast<const unsigned char*>(wd_pairs);
    a.rows_per_blk = tail_rows_per_block(H, Wd, width / 32, stride, &a.mblocks);
    LDN_REQUIRE(a.rows_per_blk > 0, "ldn_bottleneck_tail: a %dx%d map of width %d (stride %d) does not fit the workgroup (ldn_bottleneck_tail_splits == 0)", H, Wd, width, stride);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (f32) {
        LDN_REQUIRE(!x_split, "ldn_bottleneck_tail: the folded projection is a bf16x3 form");
        if (stride == 2) {
            if (width == 64) return launch_tail<2, 2, false, true>(a, st);
            if (width == 128) return launch_tail<4, 2, false, true>(a, st);
            return launch_tail<8, 2, false, true>(a, st);
        }
        if (width == 64) return launch_tail<2, 1, false, true>(a, st);
        if (width == 128) return launch_tail<4, 1, false, true>(a, st);
        return launch_tail<8, 1, false, true>(a, st);
    }
    if (stride == 2) {
        if (width == 64) return launch_tail<2, 2>(a, st);
        if (width == 128) return launch_tail<4, 2>(a, st);
        return launch_tail<8, 2>(a, st);
    }
    if (x_split) return launch_tail<2, 1, true>(a, st);
    if (width == 64) return launch_tail<2, 1>(a, st);
    if (width == 128) return launch_tail<4, 1>(a, st);
    return launch_tail<8, 1>(a, st);
}

namespace ldn {

// ================================================================================================================ k_head
// conv1 of a channel-mode bottleneck in the same style as k_tail (transposed MFMA formulation, lane = pixel, bf16x3):
//     h1 = relu(bn1(conv1x1(x)[active output channels])) - c1          (laud_resnet.py:115-118 on the image's channel list)
// written PRE-SPLIT for k_tail ([pixel][octet][8 hi | 8 lo] bf16, zero-filled to a multiple of 32 columns).
// Why a second kernel for it (the general k_conv_bf3 did this launch in round 1): measured on the stage-3 launch, a third of
// k_conv_bf3's 180-196 k cycles per image were prologue + epilogue, and its two-deep staging kept only one K chunk in flight
// per CU (producers 112 k cycles in issue).  Here:
//   * the weights are pre-split once per module (n-major rows [n][octet][8 hi | 8 lo]): a weight fragment is two
//     ds_read_b128 and NO VALU, the row gather through the channel list is a pointer;
//   * x lands as raw fp32 through LDS-DMA and is split by the one wave that owns the pixels (24 VALU per K16 step, shared by
//     all of the wave's n-subtiles);
//   * a ring of D = 2..4 slots (as many as the image's subset leaves room for in 160 KiB) with counted vmcnt: D - 1 chunks of
//     52 KB are in flight per CU -- the launch is bounded by the CU's memory pipe (x + the gathered weight rows), not by latency;
//   * the epilogue pairs the two half-waves of a pixel with v_permlane32_swap so that every lane stores 16 contiguous bytes.
struct HeadArgs {
    const float* x; int ldx;
    int B, HW, cin, W;
    const unsigned char* w1s;                         // [W][cin / 8][32 B]
    const int32_t* n_idx; const int32_t* n_cnt;       // [B][W], [B]
    const float* sc1; const float* sh1; const float* ps1;   // [W]
    unsigned char* h1; long h1_row_bytes;
    int pix_per_blk, mblocks;
    unsigned char* xs;                                // optional: x itself written pre-split in 32-pixel tiles (ldn_bottleneck_head_split) for a folded projection
};

template <int N> __device__ __forceinline__ void wait_vm_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_rt(int n) {   // counted wait with a run-time (wave-uniform) count, 0..16
    switch (n) {
        case 0: wait_vm_n<0>(); break;   case 1: wait_vm_n<1>(); break;   case 2: wait_vm_n<2>(); break;
        case 3: wait_vm_n<3>(); break;   case 4: wait_vm_n<4>(); break;   case 5: wait_vm_n<5>(); break;
        case 6: wait_vm_n<6>(); break;   case 7: wait_vm_n<7>(); break;   case 8: wait_vm_n<8>(); break;
        case 9: wait_vm_n<9>(); break;   case 10: wait_vm_n<10>(); break; case 11: wait_vm_n<11>(); break;
        case 12: wait_vm_n<12>(); break; case 13: wait_vm_n<13>(); break; case 14: wait_vm_n<14>(); break;
        case 15: wait_vm_n<15>(); break; default: wait_vm_n<16>(); break;
    }
}

// NF consecutive 1 KB pieces of a ring slot with ONE address set-up: piece f = sbase (wave-uniform) + vo[f] (per-lane byte offset, biased by
// the caller with (3 - f) * 1024 against sbase - 3072) -> LDS lds_base + f * 1024 + lane * 16.  The instruction offset of
// global_load_lds_dwordx4 moves the LDS destination as well as the global source (tools/experiments/dma_offset.hip).
template <int NF> __device__ __forceinline__ void dma16_pieces(const unsigned (&vo)[4], const void* sbase, unsigned lds_base) {
#if LDN_TAIL_ABLATE & 1
    return;
#endif
    unsigned keep;
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
__device__ __forceinline__ const void* uniform_cptr(const void* v) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}

constexpr int H_XROWS = 256;                          // rows of the x tile of a ring slot (pixels of the block, padded)

template <int NS, bool F32 = false, bool PRE_ROWS = false>
__device__ __forceinline__ void head_body(const HeadArgs& p, const int b, const int mb, unsigned char* const smem, const int lds_total, const int tid) {
    constexpr int W = NS * 32;
    int* const s_nidx = reinterpret_cast<int*>(smem);                    // [W + 32]
    float* const s_tab = reinterpret_cast<float*>(smem + T_KIDX_BYTES);  // sc1 | sh1 | ps1 of the packed columns, 3 x W
    unsigned char* const s_ring = smem + T_KIDX_BYTES + 3 * W * 4;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = mb * p.pix_per_blk;
    const int npix = min(p.pix_per_blk, p.HW - m0);
    const long row0 = (long)b * p.HW + m0;

    const int Nb = min(p.n_cnt[b], W);
    const int nsub = __builtin_amdgcn_readfirstlane(ceil_div(Nb, 32));
    const int wrows = round_up(max(Nb, 1), 64);                          // weight rows staged per chunk (DMA instructions cover 8 rows)
    if (tid < W + 32) s_nidx[tid] = tid < Nb ? p.n_idx[(size_t)b * W + tid] : -1;
    __syncthreads();
    for (int i = tid; i < 3 * W; i += 512) {
        const int k = i / W, n = i - k * W;
        const int ch = s_nidx[n];
        const float* src = k == 0 ? p.sc1 : (k == 1 ? p.sh1 : p.ps1);
        s_tab[i] = ch >= 0 ? src[ch] : 0.f;
    }
    // ring geometry: slot = [256 x rows | wrows weight rows] x 128 B; as many slots as fit (2..4)
    const int xrows = round_up(npix, 32);                                // x rows staged per chunk: whole 32-pixel subtiles of busy waves
    const int slot_bytes = (xrows + wrows) * 128;
#ifdef LDN_HEAD_L2PF     // (round 5, measured: 12.43-12.49 vs 11.99 ms -- SLOWER: the phase is not waiting on HBM latency; kept as a switch, off)
    constexpr int PF = 1;              // one L2-prefetch instruction per wave and chunk (see dma_chunk)
#else
    constexpr int PF = 0;
#endif
    const int avail = lds_total - (T_KIDX_BYTES + 3 * W * 4) - 256;     // (256 B at the end: the scratch the prefetch words land in)
    const int D = min(4, avail / slot_bytes);                            // >= 2 for W <= 256
    const int nchunks = p.cin / 32;
    const unsigned lds_ring = lds_off(s_ring);
    const int nw = wrows / 64;                                           // weight DMA instructions per wave and chunk (1..4)
    const bool active = wave * 32 < npix;
    const int per_chunk = (active ? 4 + PF : 0) + nw;                    // this wave's DMA instructions per chunk (4 = its 32 x rows, PF = the L2 prefetch)

    // DMA of chunk c into slot c % D.  Wave w moves x rows [32 w, 32 w + 32) and weight rows {64 i + 8 w .. + 7} (i < nw).
    // one instruction = 8 rows x 128 B: lane = (row r0 + (lane >> 3), physical slot lane & 7); XOR swizzle on the SOURCE address.
    // Round 3: a wave's pieces are CONSECUTIVE kilobytes of the slot (the weight rows are staged wave-major: piece i of wave w is the 1 KB
    // block w * nw + i) and take one address set-up per group (dma16_pieces): the per-lane source offsets are fixed for the whole K loop,
    // the chunk only moves the scalar base.  Rows beyond the block's pixels / the image's list fetch existing rows instead of a zero line:
    // those pixels are never stored, and the accumulator columns beyond the list meet zero epilogue tables.
    unsigned xo[4], wo[4];
    {
        const int r8 = lane >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + i * 8 + r8;
            const int ls = (lane & 7) ^ ((r >> 1) & 7);
            xo[i] = (unsigned)((min(r, npix - 1) * p.ldx + ls * 4) * 4 + (3 - i) * 1024);
            const int rw = i * 64 + wave * 8 + r8;
            const int lw = (lane & 7) ^ (r8 >> 1);
            const int ch = s_nidx[min(rw, max(Nb, 1) - 1)];
            wo[i] = (unsigned)(max(ch, 0) * p.cin * 4 + lw * 16 + (3 - i) * 1024);
        }
    }
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(p.x + row0 * p.ldx) - 3072;
    const unsigned char* const wbase = p.w1s - 3072;
    auto dma_chunk = [&](int c) {
        const unsigned slot = lds_ring + (c % D) * slot_bytes;
        const int cc = min(c, nchunks - 1);              // chunks beyond the K loop: the last one again (keeps the per-iteration DMA count constant)
        if (active) dma16_pieces<4>(xo, uniform_cptr(xbase + (long)cc * 128), __builtin_amdgcn_readfirstlane(slot + wave * 32 * 128));
        if (PF && active) {
            // L2 prefetch of this wave's x rows two and three chunks beyond the ring (round 5): the phase is bound by bytes in
            // flight / HBM latency at one workgroup per CU; a touched line waits in the XCD's L2 when its 16-byte DMA comes (lanes 0-31: chunk
            // c + 2, lanes 32-63: chunk c + 3; beyond the K range the last chunk again)
            const int pc = min(c + 2 + (lane >> 5), nchunks - 1);
            dma4_touch(reinterpret_cast<const unsigned char*>(p.x + (row0 + min(wave * 32 + l31, npix - 1)) * p.ldx) + (long)pc * 128,
                       __builtin_amdgcn_readfirstlane((unsigned)(lds_ring + (unsigned)avail)));
        }
        const void* wb = uniform_cptr(wbase + (long)cc * 128);
        const unsigned wl = __builtin_amdgcn_readfirstlane(slot + (xrows + wave * nw * 8) * 128);
        if (nw == 1) dma16_pieces<1>(wo, wb, wl);
        else if (nw == 2) dma16_pieces<2>(wo, wb, wl);
        else if (nw == 3) dma16_pieces<3>(wo, wb, wl);
        else dma16_pieces<4>(wo, wb, wl);
    };
    auto dma_dummy = [&](int c) { dma_chunk(c); };

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int c = 0; c < D - 1; ++c) { if (c < nchunks) dma_chunk(c); else dma_dummy(c); }
    const unsigned xrow = (unsigned)(wave * 32 + l31), xsw = (xrow >> 1) & 7u;
    const unsigned wsw = ((unsigned)l31 >> 1) & 3u;      // weight rows are swizzled by their row WITHIN the 8-row piece
    // list row n = 32 j + l31 is row n % 8 of piece n / 64 of wave (n % 64) / 8: LDS row ((4 (j & 1) + (l31 >> 3)) * nw + (j >> 1)) * 8 + (l31 & 7)
    const unsigned wrow_e = (unsigned)((((l31 >> 3)) * nw * 8 + (l31 & 7)) * 128);            // even j
    const unsigned wrow_o = (unsigned)((((4 + (l31 >> 3))) * nw * 8 + (l31 & 7)) * 128);      // odd j
#ifdef LDN_TRACE
    unsigned long long h0, h1, h2, h3, h4, h5, hw = 0, hb = 0, hi_ = 0, hs = 0, hm = 0, hstart;
    TT(hstart)
#endif
    // B operands of both K16 steps of a chunk (this wave's 32 pixels), split once for all of the image's n-subtiles.  The rows are staged by
    // THIS wave's own four DMA pieces (the first of its chunk), so they need its vmcnt only -- the workgroup barrier is for the weight
    // rows.  PRE_ROWS: chunk c + 1's rows are read and split right behind chunk c's MFMA steps, off the barrier -> first-MFMA path.
    bf16x8 bh[2], bl[2];
    auto load_b = [&](int c) {
        const unsigned char* xs = s_ring + (c % D) * slot_bytes;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned sl = 4u * half + 2u * h;          // logical 16-byte slot of this lane's 8 k values (x: fp32 k .. k+3, k+4 .. k+7)
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + xrow * 128 + ((sl ^ xsw) << 4));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + xrow * 128 + (((sl + 1) ^ xsw) << 4));
            if constexpr (F32) {     // raw floats: k-slots 0-3 / 4-7
                bh[half] = __builtin_bit_cast(bf16x8, x0);
                bl[half] = __builtin_bit_cast(bf16x8, x1);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? x0[e] : x1[e - 4];
                    const __bf16 hb = (__bf16)v;
                    bh[half][e] = hb;
                    bl[half][e] = (__bf16)(v - (float)hb);
                }
            }
        }
        if (!F32 && p.xs && (int)xrow < npix) {
            // the split x fragments ARE whole octets of the pre-split format: lane (pixel, h), K16 half `half` of chunk c = octet 4 c + 2 half + h.
            // (Stores share vmcnt with the LDS-DMA: the counted waits then ask for MORE completions than they need -- safe.)
            // TILED layout (tiles of 32 consecutive pixels of the flat batch): [tile][K16 step s][h][hi | lo][pixel % 32][16 B] -- the
            // consumer's B-fragment load (lane = pixel, fixed s / h / plane) then covers two contiguous 512-byte runs
            const long q = row0 + xrow;
            unsigned char* xo_ = p.xs + (q >> 5) * ((long)p.cin * 128) + (q & 31) * 16 + h * 1024;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                *reinterpret_cast<bf16x8*>(xo_ + (2 * c + half) * 2048) = bh[half];
                *reinterpret_cast<bf16x8*>(xo_ + (2 * c + half) * 2048 + 512) = bl[half];
            }
        }
    };
    // Inside the chained kernel it measured SLOWER on the headline (three interleaved pairs: 12.03 vs 11.92 ms -- the register allocation of
    // the 14 k-instruction kernel again), so k_chain instantiates PRE_ROWS = false; the stand-alone k_head takes it.
    constexpr bool PRE = PRE_ROWS;
    const int own_landed = per_chunk * (D - 1) - 4;      // own rows of the oldest chunk in flight = its first four pieces
    if (PRE && active && nchunks > 0) {
        wait_vm_rt(own_landed);
        load_b(0);
    }
    for (int c = 0; c < nchunks; ++c) {
        TT(h0)
        wait_vm_rt(per_chunk * (D - 2));     // chunk c has landed; the D - 2 chunks issued after it may still fly
        TT(h1)
        lds_barrier();                       // ... for every wave; every wave has left chunk c - 1
        TT(h2)
        // (Round 4, tried: the SIMDs' second waves issuing this AFTER their MFMA steps, so that one wave's DMA issue overlaps the other's
        // matrix time -- the K loop stayed at 4.4 k cycles per chunk and the step got 0.15 ms slower; profiles/r04_trace_head.txt)
        if (c + D - 1 < nchunks) dma_chunk(c + D - 1); else dma_dummy(c + D - 1);
        TT(h3)
        TT_ADD(hw, h0, h1) TT_ADD(hb, h1, h2) TT_ADD(hi_, h2, h3)
        if (!active) continue;
        const unsigned char* ws = s_ring + (c % D) * slot_bytes + xrows * 128;
        if (!PRE) load_b(c);
#ifdef LDN_TRACE
        asm volatile("" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
        TT(h4)
        TT_ADD(hs, h3, h4)
#endif
        // The image's n-subtiles in DESCENDING order as ONE linear, software-pipelined sequence with an entry point per subtile count
        // (a switch that falls through: no duplicated code, every accumulator keeps its registers).  Step j = the two K16 halves of
        // subtile j; a weight fragment is two ds_read_b128 (8 hi | 8 lo of row 32 j + l31, no VALU), double-buffered in a0 / a1: the
        // reads of (j, half 1) fly during the MFMAs of (j, half 0), those of (j - 1, half 0) during the MFMAs of (j, half 1).
        // (Left to itself hipcc emits read, s_waitcnt lgkmcnt(0), MFMA per fragment behind a branch per subtile: a full LDS latency
        // per 96 MFMA cycles, hidden only by the SIMD's other wave.)
        bf16x8 a0h, a0l, a1h, a1l;
        const unsigned sl0 = 2u * h, sl1 = 4u + 2u * h;
        auto frag0 = [&](int j) {
            const unsigned char* wr = ws + ((j & 1) ? wrow_o : wrow_e) + (j >> 1) * 1024;
            a0h = *reinterpret_cast<const bf16x8*>(wr + ((sl0 ^ wsw) << 4));
            a0l = *reinterpret_cast<const bf16x8*>(wr + (((sl0 + 1) ^ wsw) << 4));
        };
        auto frag1 = [&](int j) {
            const unsigned char* wr = ws + ((j & 1) ? wrow_o : wrow_e) + (j >> 1) * 1024;
            a1h = *reinterpret_cast<const bf16x8*>(wr + ((sl1 ^ wsw) << 4));
            a1l = *reinterpret_cast<const bf16x8*>(wr + (((sl1 + 1) ^ wsw) << 4));
        };
#define LDN_HEAD_STEP(J)                                                                                  \
        frag1(J);                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        LDN_K16(F32, acc[J], a0h, a0l, bh[0], bl[0])                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (J > 0) frag0(J > 0 ? J - 1 : 0);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        LDN_K16(F32, acc[J], a1h, a1l, bh[1], bl[1])                                                      \
        __builtin_amdgcn_sched_barrier(0);
        if (nsub > 0) {
            frag0(nsub - 1);
            switch (nsub) {
                default:
                    if constexpr (NS >= 8) { LDN_HEAD_STEP(7) }
                    [[fallthrough]];
                case 7:
                    if constexpr (NS >= 8) { LDN_HEAD_STEP(6) }
                    [[fallthrough]];
                case 6:
                    if constexpr (NS >= 8) { LDN_HEAD_STEP(5) }
                    [[fallthrough]];
                case 5:
                    if constexpr (NS >= 8) { LDN_HEAD_STEP(4) }
                    [[fallthrough]];
                case 4:
                    if constexpr (NS >= 4) { LDN_HEAD_STEP(3) }
                    [[fallthrough]];
                case 3:
                    if constexpr (NS >= 4) { LDN_HEAD_STEP(2) }
                    [[fallthrough]];
                case 2:
                    LDN_HEAD_STEP(1)
                    [[fallthrough]];
                case 1:
                    LDN_HEAD_STEP(0)
            }
        }
#undef LDN_HEAD_STEP
        if (PRE && c + 1 < nchunks) {
            wait_vm_rt(own_landed);
            load_b(c + 1);
        }
#ifdef LDN_TRACE
        asm volatile("" : "+v"(acc[0]));
        TT(h5)
        TT_ADD(hm, h4, h5)
#endif
    }
#ifdef LDN_TRACE
    if (g_head_trace && lane == 0) {
        unsigned long long hend;
        TT(hend)
        unsigned long long* r = g_head_trace + (((size_t)mb * p.B + b) * 8 + wave) * 8;
        r[0] = hw; r[1] = hb; r[2] = hi_; r[3] = hs; r[4] = hm; r[5] = hend - hstart; r[6] = nchunks; r[7] = nsub;
    }
#endif
    wait_vm_n<0>();      // no LDS-DMA may be in flight when the workgroup's LDS is released

    // ---- epilogue: bn1 + ReLU - c1, split, pair the half-waves, 16-byte stores of [8 hi] (lanes 0-31) / [8 lo] (lanes 32-63)
    const int pm = wave * 32 + l31;
    if (!active) return;
    unsigned char* orow = p.h1 + (row0 + min(pm, npix - 1)) * p.h1_row_bytes;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (j >= nsub) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int n0 = 32 * j + 8 * q4 + 4 * h;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + W + n0);
            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + 2 * W + n0);
            if constexpr (F32) {   // true fp32: the lane's four channels 32 j + 8 q4 + 4 h .. + 3 as plain floats (the same 16 bytes of the row)
                f32x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = fmaxf(acc[j][4 * q4 + e] * sc[e] + sh[e], 0.f) - ps[e];
                if (pm < npix) *reinterpret_cast<f32x4*>(orow + (4 * j + q4) * 32 + h * 16) = v4;
                continue;
            }
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            unsigned hi2[2], lo2[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float v0 = fmaxf(acc[j][4 * q4 + 2 * d] * sc[2 * d] + sh[2 * d], 0.f) - ps[2 * d];
                const float v1 = fmaxf(acc[j][4 * q4 + 2 * d + 1] * sc[2 * d + 1] + sh[2 * d + 1], 0.f) - ps[2 * d + 1];
                const bf16x2 hh = {(__bf16)v0, (__bf16)v1};
                const bf16x2 ll = {(__bf16)(v0 - (float)hh[0]), (__bf16)(v1 - (float)hh[1])};
                hi2[d] = __builtin_bit_cast(unsigned, hh);
                lo2[d] = __builtin_bit_cast(unsigned, ll);
            }
            // lanes (pixel, 0) hold channels 0-3 of the octet, lanes (pixel, 1) channels 4-7.  After the swaps the lower lane holds
            // the octet's 8 hi halves {own hi, partner's hi} and the upper lane its 8 lo halves {partner's lo, own lo}.
            u32x4 outv;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const auto r = __builtin_amdgcn_permlane32_swap(hi2[d], lo2[d], false, false);   // vdst = hi, src = lo
                const unsigned a = r[0], bq = r[1];   // lower lanes: a = own hi, bq = partner's hi; upper lanes: a = partner's lo, bq = own lo
                outv[d] = a;
                outv[2 + d] = bq;
            }
            if (pm < npix) *reinterpret_cast<u32x4*>(orow + (4 * j + q4) * 32 + h * 16) = outv;
        }
    }
}

// head_body2 (round 5) -- conv1 of a channel-mode block on the staging structure of k_rows3 / k_dense2, bf16x3:
//   * the image's gathered weight rows are the only thing the waves share: ring of TWO slots of one K32 step, one barrier per step;
//   * a wave's 32 x rows are staged by that wave alone into a private double buffer (its own vmcnt, no barrier);
//   * the B fragments are double-buffered in registers: step s + 1's rows are read and split BETWEEN the MFMAs of step s, the DMA of step
//     s + 3 reuses their slot right behind -- nothing but the barrier itself sits between two steps' MFMAs.
// Same products in the same order as head_body (chunk by chunk, K16 half 0 then half 1 per accumulator): bit-identical h1.
template <int NS>
__device__ __forceinline__ void head_body2(const HeadArgs& p, const int b, const int mb, unsigned char* const smem, const int tid) {
    constexpr int W = NS * 32;
    constexpr int WSLOT = W * 128;                    // a K32 step of (up to) W gathered weight rows, 16-byte units XOR-swizzled by ((row >> 1) & 7)
    constexpr int RSLOT = 32 * 128;
    constexpr int NWM = W / 64;                       // weight DMA instructions per wave and step when the image keeps every channel
    int* const s_nidx = reinterpret_cast<int*>(smem);                    // [W + 32]
    float* const s_tab = reinterpret_cast<float*>(smem + T_KIDX_BYTES);  // sc1 | sh1 | ps1 of the packed columns, 3 x W
    unsigned char* const s_w = smem + T_KIDX_BYTES + 3 * W * 4;
    unsigned char* const s_r = s_w + 2 * WSLOT;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = mb * p.pix_per_blk;
    const int npix = min(p.pix_per_blk, p.HW - m0);
    const long row0 = (long)b * p.HW + m0;

    const int Nb = min(p.n_cnt[b], W);
    const int nsub = __builtin_amdgcn_readfirstlane(ceil_div(Nb, 32));
    const int nw = __builtin_amdgcn_readfirstlane(ceil_div(max(Nb, 1), 64));      // weight DMA instructions per wave and step (64 rows each over the 8 waves)
    if (tid < W + 32) s_nidx[tid] = tid < Nb ? p.n_idx[(size_t)b * W + tid] : -1;
    __syncthreads();
    for (int i = tid; i < 3 * W; i += 512) {
        const int k = i / W, n = i - k * W;
        const int ch = s_nidx[n];
        const float* src = k == 0 ? p.sc1 : (k == 1 ? p.sh1 : p.ps1);
        s_tab[i] = ch >= 0 ? src[ch] : 0.f;
    }
    const int nstep = p.cin / 32;
    const unsigned lds_w = lds_off(s_w), lds_r = lds_off(s_r) + (unsigned)wave * 2u * RSLOT;
    unsigned char* const my_r = s_r + wave * 2 * RSLOT;
    const bool active = wave * 32 < npix;

    // per-lane sources: weight instruction k covers list rows 64 k + 8 wave .. + 7 (rows beyond the image's list re-read its last channel: their
    // accumulator columns meet zero epilogue tables); x instruction k covers the wave's pixels 8 k .. 8 k + 7 (beyond the block: its last pixel, never stored)
    const unsigned char* wsrc[NWM];
#pragma unroll
    for (int k = 0; k < NWM; ++k) {
        const int rr = 64 * k + 8 * wave + (lane >> 3);
        const int ch = s_nidx[min(rr, max(Nb, 1) - 1)];
        wsrc[k] = p.w1s + (long)max(ch, 0) * p.cin * 4 + (((lane & 7) ^ ((rr >> 1) & 7)) << 4);
    }
    const unsigned char* rsrc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 8 * k + (lane >> 3);
        rsrc[k] = reinterpret_cast<const unsigned char*>(p.x + (row0 + min(wave * 32 + r, npix - 1)) * p.ldx) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto dma_w = [&](int S, int k) {
        dma16(wsrc[k] + (long)min(S, nstep - 1) * 128, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_w + (unsigned)(S & 1) * WSLOT + (unsigned)(64 * k + 8 * wave) * 128u)));
    };
    auto dma_w_all = [&](int S) {
#pragma unroll
        for (int k = 0; k < NWM; ++k)
            if (k < nw) dma_w(S, k);
    };
    auto dma_r = [&](int s, int k) {
        dma16(rsrc[k] + (long)min(s, nstep - 1) * 128, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_r + (unsigned)(s & 1) * RSLOT + (unsigned)k * 1024u)));
    };

    if (!active) {      // a wave without pixels only stages its share of the weights
        dma_w_all(0);
        for (int S = 0; S < nstep; ++S) {
            wait_vm_n<0>();
            lds_barrier();
            dma_w_all(S + 1);
        }
        wait_vm_n<0>();
        return;
    }

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    bf16x8 bh[2][2], bl[2][2];
    f32x4 raw[4];
    const unsigned xsw = ((unsigned)l31 >> 1) & 7u;
    auto read_raw = [&](int s) {
        const unsigned char* xs = my_r + (s & 1) * RSLOT + l31 * 128;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned sl = 4u * half + 2u * h;
            raw[2 * half] = *reinterpret_cast<const f32x4*>(xs + ((sl ^ xsw) << 4));
            raw[2 * half + 1] = *reinterpret_cast<const f32x4*>(xs + (((sl + 1) ^ xsw) << 4));
        }
    };
    auto split_b = [&](int half, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = e < 4 ? raw[2 * half][e] : raw[2 * half + 1][e - 4];
            const __bf16 hb = (__bf16)v;
            dh[half][e] = hb;
            dl[half][e] = (__bf16)(v - (float)hb);
        }
    };

    // prologue: W(0), R(0), R(1); B(0) -> registers; R(2) into R(0)'s slot; R(1) landed before the loop's counted waits start
    dma_w_all(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(1, k);
    wait_vm_n<4>();
    read_raw(0);
    split_b(0, bh[0], bl[0]);
    split_b(1, bh[0], bl[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(2, k);
    wait_vm_n<4>();

    // one step with B register set CUR: [W(S) landed: vmcnt(4)] barrier | MFMA steps (K16 half 0 of every n-subtile, then half 1) with, behind
    // the first ones, W(S + 1), the raw read of x(S + 1) [x(S + 1) landed: vmcnt(4 + 2 nw)], its split into the other set, and R(S + 3)
    auto step = [&](int S, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        wait_vm_n<4>();
        lds_barrier();
        const unsigned char* wsl = s_w + (S & 1) * WSLOT + l31 * 128;
        int stage = 0;      // what has gone out of: 0 W(S + 1), 1 read, 2 split 0, 3 split 1, 4 R(S + 3)
        auto extra = [&]() {
            if (stage == 0) dma_w_all(S + 1);
            else if (stage == 1) { wait_vm_rt(4 + 2 * nw); read_raw(S + 1); }
            else if (stage == 2) split_b(0, bh[CUR ^ 1], bl[CUR ^ 1]);
            else if (stage == 3) split_b(1, bh[CUR ^ 1], bl[CUR ^ 1]);
            else if (stage == 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) dma_r(S + 3, k);
            }
            ++stage;
        };
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned uu = 2u * (2u * half + h);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (j < nsub) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wsl + j * 4096 + ((uu ^ xsw) << 4));
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(wsl + j * 4096 + (((uu + 1) ^ xsw) << 4));
                    acc[j] = t_mfma_bf16(al, bh[CUR][half], acc[j]);
                    acc[j] = t_mfma_bf16(ah, bl[CUR][half], acc[j]);
                    acc[j] = t_mfma_bf16(ah, bh[CUR][half], acc[j]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
                // compile-time positions (n-subtile 0 always exists, 1 and 2 almost always): one extra behind each of the first steps of a half
                if (j <= 2 && j < nsub) { __builtin_amdgcn_sched_barrier(0); extra(); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        while (stage < 5) extra();       // (images with fewer than three n-subtiles: the rest goes out behind the last step)
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int S = 0; S < nstep; S += 2) {      // (cin % 64 == 0: an even number of steps; the register sets alternate)
        step(S, std::integral_constant<int, 0>{});
        step(S + 1, std::integral_constant<int, 1>{});
    }
    wait_vm_n<0>();      // no LDS-DMA may be in flight when the workgroup's LDS is released

    // ---- epilogue (as head_body): bn1 + ReLU - c1, split, pair the half-waves, 16-byte stores of [8 hi] (lanes 0-31) / [8 lo] (lanes 32-63)
    const int pm = wave * 32 + l31;
    unsigned char* orow = p.h1 + (row0 + min(pm, npix - 1)) * p.h1_row_bytes;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (j >= nsub) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int n0 = 32 * j + 8 * q4 + 4 * h;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + W + n0);
            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + 2 * W + n0);
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            unsigned hi2[2], lo2[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float v0 = fmaxf(acc[j][4 * q4 + 2 * d] * sc[2 * d] + sh[2 * d], 0.f) - ps[2 * d];
                const float v1 = fmaxf(acc[j][4 * q4 + 2 * d + 1] * sc[2 * d + 1] + sh[2 * d + 1], 0.f) - ps[2 * d + 1];
                const bf16x2 hh = {(__bf16)v0, (__bf16)v1};
                const bf16x2 ll = {(__bf16)(v0 - (float)hh[0]), (__bf16)(v1 - (float)hh[1])};
                hi2[d] = __builtin_bit_cast(unsigned, hh);
                lo2[d] = __builtin_bit_cast(unsigned, ll);
            }
            u32x4 outv;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const auto r = __builtin_amdgcn_permlane32_swap(hi2[d], lo2[d], false, false);
                outv[d] = r[0];
                outv[2 + d] = r[1];
            }
            if (pm < npix) *reinterpret_cast<u32x4*>(orow + (4 * j + q4) * 32 + h * 16) = outv;
        }
    }
}

template <int NS, bool F32 = false>
__global__ __launch_bounds__(512, 2) void k_head(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef LDN_HEAD_V2      // (round 5: measured neutral on the headline -- 12.11-12.26 vs 12.20 ms.  NOT because conv1 is bound by the CU's fetch of x, as this line used to say: DESIGN.md 4v
                        // retracts that -- the phase keeps its time without DMA and without MFMA; what binds it is the per-chunk serial chain)
    if constexpr (!F32) {
        if (p.xs == nullptr && p.cin % 64 == 0) {     // (wave-uniform) the round-5 staging structure; the x_split by-product keeps the old body
            head_body2<NS>(p, blockIdx.x % p.B, blockIdx.x / p.B, smem, threadIdx.x);
            return;
        }
    }
#endif
#ifdef LDN_HEAD_NO_PRE
    head_body<NS, F32, false>(p, blockIdx.x % p.B, blockIdx.x / p.B, smem, 160 * 1024, threadIdx.x);
#else
    head_body<NS, F32, true>(p, blockIdx.x % p.B, blockIdx.x / p.B, smem, 160 * 1024, threadIdx.x);
#endif
}

template <int NS, bool F32 = false>
static int launch_head(HeadArgs& a, hipStream_t st) {
    const size_t lds = 160 * 1024;
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_head<NS, F32>), lds), "k_head: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL((k_head<NS, F32>), dim3((unsigned)a.B * a.mblocks), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_head");
    return LDN_OK;
}

}  // namespace ldn

static int bottleneck_head_impl(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                                const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                                const float* post_sub1, void* h1_split, int ldh, void* x_split, bool f32, void* stream);

extern "C" int ldn_bottleneck_head(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                                   const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                                   const float* post_sub1, void* h1_split, int ldh, void* stream) {
    return bottleneck_head_impl(x, ldx, B, HW, cin, w1_split, width, ch_idx, ch_cnt, scale1, shift1, post_sub1, h1_split, ldh, nullptr, false, stream);
}

extern "C" int ldn_bottleneck_head_f32(const float* x, int ldx, int B, int HW, int cin, const float* w1, int width,
                                       const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                                       const float* post_sub1, float* h1, int ldh, void* stream) {
    return bottleneck_head_impl(x, ldx, B, HW, cin, w1, width, ch_idx, ch_cnt, scale1, shift1, post_sub1, h1, ldh, nullptr, true, stream);
}

extern "C" size_t ldn_x_split_bytes(size_t pixels, int cin) { return ((pixels + 31) / 32) * (size_t)cin * 128; }

extern "C" int ldn_bottleneck_head_split(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                                         const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                                         const float* post_sub1, void* h1_split, int ldh, void* x_split, void* stream) {
    LDN_REQUIRE(x_split && (uintptr_t)x_split % 16 == 0, "ldn_bottleneck_head_split: x_split must be a 16-byte aligned buffer of ldn_x_split_bytes(B * HW, cin) bytes");
    return bottleneck_head_impl(x, ldx, B, HW, cin, w1_split, width, ch_idx, ch_cnt, scale1, shift1, post_sub1, h1_split, ldh, x_split, false, stream);
}

static int bottleneck_head_impl(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                                const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                                const float* post_sub1, void* h1_split, int ldh, void* x_split, bool f32, void* stream) {
    using namespace ldn;
    LDN_REQUIRE(x && w1_split && ch_idx && ch_cnt && scale1 && shift1 && post_sub1 && h1_split, "ldn_bottleneck_head: null pointer");
    LDN_REQUIRE(width == 64 || width == 128 || width == 256, "ldn_bottleneck_head: width must be 64, 128 or 256 (got %d)", width);
    LDN_REQUIRE(B > 0 && HW > 0 && cin > 0 && cin % 32 == 0, "ldn_bottleneck_head: cin must be a multiple of 32 (got %d)", cin);
    LDN_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldh >= width && ldh % 8 == 0, "ldn_bottleneck_head: bad ldx / ldh");
    LDN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w1_split % 16 == 0 && (uintptr_t)h1_split % 16 == 0, "ldn_bottleneck_head: pointers must be 16-byte aligned");
    HeadArgs a{};
    a.x = x; a.ldx = ldx; a.B = B; a.HW = HW; a.cin = cin; a.W = width;
    a.w1s = static_cast<const unsigned char*>(w1_split);
    a.n_idx = ch_idx; a.n_cnt = ch_cnt; a.sc1 = scale1; a.sh1 = shift1; a.ps1 = post_sub1;
    a.h1 = static_cast<unsigned char*>(h1_split); a.h1_row_bytes = (long)ldh * 4;
    a.xs = static_cast<unsigned char*>(x_split);
    a.mblocks = ceil_div(HW, 256);
    a.pix_per_blk = ceil_div(HW, a.mblocks);
    // whole 32-pixel wave tiles per workgroup where that still takes the same number of workgroups (784 pixels: 224 + 224 + 224 + 112
    // instead of 4 x 196 with a nearly empty seventh wave each)
    if (const int up = round_up(a.pix_per_blk, 32); up <= 256 && (long)up * (a.mblocks - 1) < HW) a.pix_per_blk = up;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (f32) {
        if (width == 64) return launch_head<2, true>(a, st);
        if (width == 128) return launch_head<4, true>(a, st);
        return launch_head<8, true>(a, st);
    }
    if (width == 64) return launch_head<2>(a, st);
    if (width == 128) return launch_head<4>(a, st);
    return launch_head<8>(a, st);
}

namespace ldn {

// ================================================================================================================ k_chain
// A RUN of consecutive stride-1 channel-mode bottlenecks whose map fits one workgroup (H * W <= 256: stage 3 of the ResNets)
// as ONE launch: workgroup b walks image b through   masker (GAP partials -> MLP -> channel list)  ->  conv1 (head_body)  ->
// conv2 + conv3 (tail_body)   of every block of the run (laud_resnet.py:104-147 block after block).  Eval-mode images are
// independent, so nothing synchronises across images: with one launch per phase every launch lasts as long as its heaviest
// image (conv2 work grows with the square of the image's active channels: measured max / mean = 1.32 at stage 3); chained,
// an image's cost is its own sum over the run, the light and heavy blocks of an image average out, and the workgroups drift apart so
// that some are in their MFMA-bound conv2 phase while others are in their memory-bound conv3 phase.
// Same device code as the three stand-alone kernels (channel_mlp_body / head_body / tail_body): results are bit-identical.
// Between phases the workgroup's own global writes (channel list, h1, the residual stream, GAP partials) are made visible to
// its own later reads: workgroup-scope release fence + barrier + acquire fence, and a scalar-cache invalidate.
struct ChainBlock {   // per-block constants; every field is 8 bytes (= ldn_chain_block of include/ldn_hip.h)
    const unsigned char* w1s; const float* sc1; const float* sh1; const float* ps1;
    const unsigned char* w2p; const unsigned char* w3p; const float* sc2; const float* sh2; const float* ps2; const float* sh3;
    const float* mw1; const float* mb1; const float* mw2; const float* mb2;
};
static_assert(sizeof(ChainBlock) == 14 * 8, "ChainBlock must mirror ldn_chain_block");

struct ChainArgs {
    const ChainBlock* blocks; int nblocks;
    const float* x_in; float* x_work; int ldx;        // block 0 reads x_in (input and residual), every block writes x_work
    int B, H, Wd, C;
    int hidden, G, gran;
    const float* gap_in; int gap_splits;              // GAP partials of x_in [B][gap_splits][C]
    float* colsum;                                    // [B][8][C]: GAP partials of the latest block's output (the run's GAP output)
    float* masks; int32_t* ch_idx; int32_t* ch_cnt;   // [nblocks][B][G], [nblocks][B][W], [nblocks][B]
    unsigned char* h1; long h1_row_bytes;
    int slice_bytes, lds_total;
};

__device__ __forceinline__ void phase_fence() {
    // WORKGROUP scope: the waves of a workgroup share their CU's vector L1, which is coherent with its own writes.  (An
    // agent-scope fence writes back and invalidates the XCD's L2 on gfx950 -- measured: +200 us per block.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// the thread index behind an optimisation barrier: the per-lane geometry of a phase (tap rows, swizzles, source offsets) is the
// same for every block of the run, and hoisted out of the block loop it would stay live through all three phases
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

template <typename T> __device__ __forceinline__ T uniform_ptr(T v) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T>(((unsigned long long)hi << 32) | lo);
}

#ifdef LDN_TRACE   // tuning only: per-image cycles of the masker / conv1 / conv2+conv3 phases and of the fences, summed over the run
#define CT(x) x = __builtin_amdgcn_s_memtime();
#else
#define CT(x)
#endif

template <int NS, bool F32 = false>
__global__ __launch_bounds__(512, (NS == 2 ? 4 : 2)) void k_chain(const ChainArgs p) {
    constexpr int W = NS * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int HW = p.H * p.Wd;
#ifdef LDN_TRACE
    unsigned long long c0, c1, c2, c3, c4, c5, c6, am = 0, ah = 0, at = 0, af = 0;
#endif
    for (int i = 0; i < p.nblocks; ++i) {
        CT(c0)
        const ChainBlock* cb = p.blocks + i;
        float* const mask_i = p.masks + (size_t)i * p.B * p.G;
        int32_t* const idx_i = p.ch_idx + (size_t)i * p.B * W;
        int32_t* const cnt_i = p.ch_cnt + (size_t)i * p.B;
        const float* const xin = i == 0 ? p.x_in : p.x_work;
        {   // ---- channel masker of block i on the GAP of its input
            float* const s_f = reinterpret_cast<float*>(smem);
            int* const s_w = reinterpret_cast<int*>(s_f + p.C + (p.hidden > 0 ? p.hidden : 1) + 2 * p.G);
            channel_mlp_body<512>(b, i == 0 ? p.gap_in : p.colsum, HW, p.C, i == 0 ? p.gap_splits : 8, uniform_ptr(cb->mw1),
                                  uniform_ptr(cb->mb1), uniform_ptr(cb->mw2), uniform_ptr(cb->mb2), p.hidden, p.G, p.gran, nullptr,
                                  mask_i, nullptr, idx_i, cnt_i, s_f, s_w);
        }
        CT(c1)
        phase_fence();
        CT(c2)
        {   // ---- conv1 -> h1 (pre-split)
            HeadArgs ha;
            ha.x = xin; ha.ldx = p.ldx; ha.B = p.B; ha.HW = HW; ha.cin = p.C; ha.W = W;
            ha.w1s = uniform_ptr(cb->w1s); ha.n_idx = idx_i; ha.n_cnt = cnt_i;
            ha.sc1 = uniform_ptr(cb->sc1); ha.sh1 = uniform_ptr(cb->sh1); ha.ps1 = uniform_ptr(cb->ps1);
            ha.h1 = p.h1; ha.h1_row_bytes = p.h1_row_bytes; ha.pix_per_blk = HW; ha.mblocks = 1; ha.xs = nullptr;
#ifdef LDN_HEAD_V2
            if constexpr (!F32) head_body2<NS>(ha, b, 0, smem, opaque_tid());
            else
#endif
            head_body<NS, F32>(ha, b, 0, smem, p.lds_total, opaque_tid());
        }
        CT(c3)
        phase_fence();
        CT(c4)
        {   // ---- conv2 -> conv3 + residual, GAP partials of the output
            TailArgs ta;
            ta.h1 = p.h1; ta.h1_row_bytes = p.h1_row_bytes;
            ta.B = p.B; ta.Hi = p.H; ta.Wi = p.Wd; ta.Ho = p.H; ta.Wo = p.Wd; ta.W = W; ta.cout = p.C;
            ta.w2p = uniform_ptr(cb->w2p); ta.w3p = uniform_ptr(cb->w3p); ta.k_idx = idx_i; ta.k_cnt = cnt_i;
            ta.sc2 = uniform_ptr(cb->sc2); ta.sh2 = uniform_ptr(cb->sh2); ta.ps2 = uniform_ptr(cb->ps2); ta.sh3 = uniform_ptr(cb->sh3);
            ta.residual = xin; ta.ldr = p.ldx; ta.out = p.x_work; ta.ldo = p.ldx; ta.colsum = p.colsum;
            ta.rows_per_blk = p.H; ta.mblocks = 1; ta.slice_bytes = p.slice_bytes; ta.pxs = nullptr; ta.pw = nullptr;
            tail_body<NS, 1, false, F32>(ta, b, 0, smem, opaque_tid());
        }
        CT(c5)
        phase_fence();
        CT(c6)
#ifdef LDN_TRACE
        am += c1 - c0; ah += c3 - c2; at += c5 - c4; af += (c2 - c1) + (c4 - c3) + (c6 - c5);
#endif
    }
#ifdef LDN_TRACE
    if (g_chain_trace && threadIdx.x == 0) {
        unsigned long long* r = g_chain_trace + (size_t)b * 4;
        r[0] = am; r[1] = ah; r[2] = at; r[3] = af;
    }
#endif
}

}  // namespace ldn
#include "ldn_chain_ld.h"
namespace ldn {

// LDS of the three phases of a chained block (masker, conv2, conv3; conv1's ring takes whatever is left): do they fit 160 KiB?
static bool chain_fits(int H, int Wd, int width, int C, int hidden, int G) {
    const int NS = width / 32;
    const int nr = H * Wd;
    const size_t slice = (size_t)round_up((round_up(nr, 8) + 1) * 128, 1024);
    const size_t lds2 = (size_t)T_KIDX_BYTES + (NS == 2 ? 1 : 2) * slice + (size_t)T_W2_SLOTS * 16 * NS * 256;
    const size_t lds3 = (size_t)T_KIDX_BYTES + 2 * (size_t)(width / 2) * (NS == 8 ? 32 : 64) * 8 + (size_t)18 * width * 4 + 8 * 4096;
    const size_t ldsm = (size_t)(C + (hidden > 0 ? hidden : 1) + 2 * G) * 4 + 64;
    // conv1: at least two ring slots of (x rows + weight rows) x 128 B behind the channel list and the tables
    const size_t lds1 = (size_t)T_KIDX_BYTES + 3 * (size_t)width * 4 + 2 * (size_t)(round_up(nr, 32) + width) * 128;
    const size_t lds = 160 * 1024;
    return lds2 <= lds && lds3 <= lds && ldsm <= lds && lds1 <= lds && round_up(nr, 8) / 8 <= 72;
}

template <int NS, bool F32 = false>
static int launch_chain(ChainArgs& a, hipStream_t st) {
    constexpr int W = NS * 32;
    const int nr = a.H * a.Wd;
    a.slice_bytes = round_up((round_up(nr, 8) + 1) * 128, 1024);
    const size_t lds = 160 * 1024;
    LDN_REQUIRE(chain_fits(a.H, a.Wd, W, a.C, a.hidden, a.G), "ldn_bottleneck_chain: the phases need more than 160 KiB of LDS (map %dx%d, width %d; ldn_bottleneck_chain_fits == 0)", a.H, a.Wd, W);
    a.lds_total = (int)lds;
    if constexpr (!F32 && NS >= 4) {
        // A map of at most 224 pixels leaves the workgroup's eighth wave without pixels: the loader / consumer form (ldn_chain_ld.h; round 6).
        // LDN_CHAIN_LD=0 keeps round 5's kernel (A/B measurements).
        static const bool use_ld = [] { const char* e = getenv("LDN_CHAIN_LD"); return !(e && atoi(e) == 0); }();
        if (use_ld && nr <= 224) {
            {   // once per device (and never inside a graph capture): the kernel learns the address of the process's fault word
                static bool armed[64] = {};
                int dev = 0;
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !armed[dev] && hipStreamIsCapturing(st, &cs) == hipSuccess &&
                    cs == hipStreamCaptureStatusNone) {
                    int* d = fault_word_dev();
                    if (d && hipMemcpyToSymbol(HIP_SYMBOL(g_ld_fault_dev), &d, sizeof(d)) == hipSuccess) armed[dev] = true;
                    else (void)hipGetLastError();
                }
            }
            LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_chain_ld<NS>), lds), "k_chain_ld: cannot reserve %zu B of LDS", lds);
            hipLaunchKernelGGL((k_chain_ld<NS>), dim3((unsigned)a.B), dim3(512), lds, st, a);
            LDN_CHECK_LAUNCH("k_chain_ld");
            return LDN_OK;
        }
    }
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_chain<NS, F32>), lds), "k_chain: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL((k_chain<NS, F32>), dim3((unsigned)a.B), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_chain");
    return LDN_OK;
}

}  // namespace ldn

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_ld_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(ldn::g_ld_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif
namespace ldn {
// hand-off waits of k_chain_ld that ran into their bound (summed into ldn_plan_timeouts)
int tu_chain_stalls(unsigned* count, int reset) {
    unsigned v = 0u;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ld_stalls), sizeof(v)) != hipSuccess) return LDN_EHIP;
    *count += v;
    if (reset && v) {
        const unsigned z = 0u;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ld_stalls), &z, sizeof(z)) != hipSuccess) return LDN_EHIP;
    }
    return LDN_OK;
}
}  // namespace ldn

#ifdef LDN_DEBUG
// test hook of the debug build: the loader wave of workgroup `image` of every following k_chain_ld launch never publishes its "landed" word, so
// that image's consumer waves run into the bound of their hand-off wait (-1 = off)
extern "C" int ldn_debug_chain_stall(int image) {
    return hipMemcpyToSymbol(HIP_SYMBOL(ldn::g_chain_stall), &image, sizeof(image)) == hipSuccess ? LDN_OK : LDN_EHIP;
}
#endif

extern "C" int ldn_bottleneck_chain_fits(int H, int Wd, int C, int width, int hidden, int G) {
    if (H < 1 || Wd < 1 || H * Wd > 256 || C < 1 || G < 1 || hidden < 0 || (width != 64 && width != 128 && width != 256)) return 0;
    return ldn::chain_fits(H, Wd, width, C, hidden, G) ? 1 : 0;
}

static int bottleneck_chain_impl(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                                 const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran,
                                 const float* gap_in, int gap_splits, float* colsum, float* masks, int32_t* ch_idx,
                                 int32_t* ch_cnt, void* h1_split, int ldh, bool f32, void* stream);
extern "C" int ldn_bottleneck_chain(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                                    const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran,
                                    const float* gap_in, int gap_splits, float* colsum, float* masks, int32_t* ch_idx,
                                    int32_t* ch_cnt, void* h1_split, int ldh, void* stream) {
    return bottleneck_chain_impl(x_in, x_work, ldx, B, H, Wd, C, width, blocks, nblocks, hidden, G, gran, gap_in, gap_splits, colsum, masks,
                                 ch_idx, ch_cnt, h1_split, ldh, false, stream);
}
/* true-fp32 arithmetic: every block's w1s / w2p / w3p in the fp32 twins of the pre-split layouts, h1 a plain fp32 scratch */
extern "C" int ldn_bottleneck_chain_f32(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                                        const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran,
                                        const float* gap_in, int gap_splits, float* colsum, float* masks, int32_t* ch_idx,
                                        int32_t* ch_cnt, float* h1, int ldh, void* stream) {
    return bottleneck_chain_impl(x_in, x_work, ldx, B, H, Wd, C, width, blocks, nblocks, hidden, G, gran, gap_in, gap_splits, colsum, masks,
                                 ch_idx, ch_cnt, h1, ldh, true, stream);
}
static int bottleneck_chain_impl(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                                 const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran,
                                 const float* gap_in, int gap_splits, float* colsum, float* masks, int32_t* ch_idx,
                                 int32_t* ch_cnt, void* h1_split, int ldh, bool f32, void* stream) {
    using namespace ldn;
    LDN_REQUIRE(x_in && x_work && blocks && gap_in && colsum && masks && ch_idx && ch_cnt && h1_split, "ldn_bottleneck_chain: null pointer");
    LDN_REQUIRE(width == 64 || width == 128 || width == 256, "ldn_bottleneck_chain: width must be 64, 128 or 256 (got %d)", width);
    LDN_REQUIRE(B > 0 && H > 0 && Wd > 0 && H * Wd <= 256, "ldn_bottleneck_chain: the map must fit one workgroup (H * W <= 256, got %dx%d)", H, Wd);
    LDN_REQUIRE(C > 0 && C % 64 == 0, "ldn_bottleneck_chain: C must be a multiple of 64 (got %d)", C);
    LDN_REQUIRE(nblocks > 0 && gap_splits > 0 && hidden >= 0 && G > 0 && gran > 0 && gran % 2 == 0 && G * gran == width,
                "ldn_bottleneck_chain: bad run / masker shape (G * gran must equal width, gran even)");
    LDN_REQUIRE(ldx >= C && ldx % 4 == 0 && ldh >= width && ldh % 8 == 0, "ldn_bottleneck_chain: bad ldx / ldh");
    LDN_REQUIRE((uintptr_t)x_in % 16 == 0 && (uintptr_t)x_work % 16 == 0 && (uintptr_t)h1_split % 16 == 0 && (uintptr_t)colsum % 16 == 0 &&
                (uintptr_t)blocks % 8 == 0, "ldn_bottleneck_chain: pointers must be 16-byte aligned");
    LDN_REQUIRE((long)9 * (width / 2) * (width / 2) * 16 < (1L << 31), "ldn_bottleneck_chain: weights too large");
    ChainArgs a{};
    a.blocks = reinterpret_cast<const ChainBlock*>(blocks); a.nblocks = nblocks;
    a.x_in = x_in; a.x_work = x_work; a.ldx = ldx; a.B = B; a.H = H; a.Wd = Wd; a.C = C;
    a.hidden = hidden; a.G = G; a.gran = gran; a.gap_in = gap_in; a.gap_splits = gap_splits; a.colsum = colsum;
    a.masks = masks; a.ch_idx = ch_idx; a.ch_cnt = ch_cnt;
    a.h1 = static_cast<unsigned char*>(h1_split); a.h1_row_bytes = (long)ldh * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (f32) {
        if (width == 64) return launch_chain<2, true>(a, st);
        if (width == 128) return launch_chain<4, true>(a, st);
        return launch_chain<8, true>(a, st);
    }
    if (width == 64) return launch_chain<2>(a, st);
    if (width == 128) return launch_chain<4>(a, st);
    return launch_chain<8>(a, st);
}
