// k_conv_image / k_conv_bf3 -- channel mode: one ragged MFMA GEMM per image (gfx950 / CDNA4), in two arithmetics:
//   k_conv_image : fp32 operands on v_mfma_f32_32x32x2_f32 (math mode 0)
//   k_conv_bf3   : bf16x3 split precision on v_mfma_f32_32x32x16_bf16, fp32 accumulate (math mode 1); its own header
//                  comment further down lists what differs (weights split/transposed once per chunk by the producers,
//                  register-blocked consumers).  Geometry, A staging, tables and the epilogue are shared.
// This header describes the common structure as first built for the fp32 kernel.
//
// For image b the active channel list selects weight rows (output subset) and/or weight columns (input
// subset) while the weight tile is staged; activations are "left-packed" (column i of image b = channel
// k_idx[b][i]).  Without index lists it is a plain dense NHWC 1x1/3x3 conv (downsample branch, conv_linear
// masker).  See include/ldn_hip.h:ldn_conv_image for the exact contract.
//
// What bounds this kernel (measured): a CU can pull only ~10 B/clk from L2/HBM (miss-queue x
// latency), i.e. ~6 TB/s for the chip, while its four fp32 MFMA pipes retire 256 FLOP/clk.  The tile a CU
// works on must therefore carry >= ~50 FLOP per byte brought into the CU, and memory-instruction issue
// (which stalls when the miss queue is full) must never sit in front of an MFMA in program order.  Hence:
//
//   * ONE workgroup per CU owning up to 160 KiB of LDS: tile = MS*32 output pixels (a whole 14x14 image at
//     stage 3) x up to NS*32 packed output columns, K walked in chunks of 32 per 3x3 tap.
//   * WAVE SPECIALISATION: 512 threads = 8 wave64.  Waves 4-7 (one per SIMD) are PRODUCERS: they issue all
//     global loads of chunk c+1 and may stall on the memory pipeline for as long as it takes.  Waves 0-3
//     (one per SIMD) are CONSUMERS: they only read LDS fragments and issue v_mfma_f32_32x32x2_f32, so each
//     SIMD's matrix pipe is fed by a wave that never waits on memory.  Two LDS buffers, one s_barrier per
//     chunk shared by both roles.
//   * Raggedness at 32x32 MFMA-tile granularity in BOTH dimensions: the block's valid m-subtiles x
//     n-subtiles (run-time counts) are dealt round-robin to the 4 consumer waves.
//   * LDS tiles are unpadded 128-byte rows whose eight 16-byte slots are XOR-swizzled with (row >> 1) & 7:
//     conflict-free for the ds_read_b128 fragment reads and -- the swizzle being applied to the SOURCE
//     address -- compatible with global_load_lds (LDS destination = wave-uniform base + lane * 16).
//     A rows (NHWC pixels) and contiguous-K weight rows go L2 -> LDS directly (LDS-DMA, no VGPR round trip);
//     K-gathered weight rows (granularity 1/2) are staged through the producers' VGPRs.
//   * fragment reads: lane (i, h) holds k = 8*g + 4*h + q, so one ds_read_b128 per operand feeds four MFMAs;
//     the reads of group g+1 are pinned in front of the MFMAs of group g (sched_group_barrier).
//   * Epilogue (consumers): folded-BN affine per lane-column, then a per-wave 32x32 transpose through LDS so
//     residual loads and output stores are 16 B per lane along the channel axis.
#include "ldn_common.h"

#ifndef LDN_ABLATE
#define LDN_ABLATE 0   // tuning only: 1 = no MFMA, 2 = no A loads in the K loop, 4 = no weight loads, 8 = no weight split/store, 16 = no output stores, 32 = no residual loads (results are wrong)
#endif

namespace ldn {

struct ImgArgs {
    const float* a; int lda;
    int B, Hi, Wi, ksize, stride, Ho, Wo;
    const float* w; int cin, cout;
    const int32_t* k_idx; const int32_t* k_cnt;
    const int32_t* n_idx; const int32_t* n_cnt;
    const float* scale; const float* shift; int shift_classes;
    const float* post_sub; int relu;
    const float* residual; int ldr;
    float* out; int ldo;
    float* colsum;   // optional [B][ceil(HWo/32)][cout]: per 32-pixel subtile column sums of the output (fused GAP)
    // ---- packed pixel-list mode (spatial / layer / both): image b owns packed rows [row_prefix[b], row_prefix[b+1])
    //      (or, with B == 1 and row_prefix == NULL, rows [0, *m_count)); see ldn_conv_packed in include/ldn_hip.h
    int packed;                      // 0 = dense image mode (all of the above), 1 = packed rows
    const int32_t* row_prefix;       // [B+1] or NULL
    const int32_t* m_count;          // device row count for the single-image form, or NULL (= m_cap)
    int m_cap;                       // worst-case rows per image (grid sizing)
    const int32_t* a_map;            // [rows][taps] A row of each tap (-1 = zero), NULL = identity (taps == 1)
    const int32_t* out_map;          // [rows] destination row in out/residual, NULL = identity
    const int32_t* pix_map;          // [rows] flat output pixel (b*Ho*Wo + oy*Wo + ox) for the border classes, or NULL
    const int32_t* relu_if_neg;      // relu == 2: ReLU only where relu_if_neg[row] < 0
    int ntn;    // N blocks per image
    int mtn;    // M blocks per image
    int bn;     // columns per N block (multiple of 32, <= NS*32)
    int bm;     // pixels per M block (multiple of 32, <= MS*32; balanced over the image)
    int math;   // 0 = fp32 MFMA, 1 = bf16x3 (resolved by the entry point, never read from a global)
    int out_split;   // 1: out rows are written PRE-SPLIT for ldn_bottleneck_tail -- per octet of 8 channels [8 hi bf16 | 8 lo bf16]
                     //    (same 4 bytes per element), zero-filled up to the next multiple of 32 channels (bf16x3 kernel only)
};

constexpr int BK = 32;   // K chunk = one 128-byte LDS row

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) float g_one16[4] = {1.f, 1.f, 1.f, 1.f};

#ifdef LDN_TRACE   // tuning only: per-block timestamps {t0, t1, hw_id, xcc_id, ntiles | barrier wait, mma | issue}
__device__ unsigned long long* g_trace = nullptr;
#define LDN_TRACE_T(x) x = __builtin_amdgcn_s_memtime();
#define LDN_TRACE_ADD(acc, a, b) acc += (b) - (a);
#else
#define LDN_TRACE_T(x)
#define LDN_TRACE_ADD(acc, a, b)
#endif

__device__ __forceinline__ void glds16(const float* src, float* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// workgroup barrier that also publishes this wave's LDS-DMA / ds_write traffic
__device__ __forceinline__ void block_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// consumer-side form: consumers never use LDS-DMA, and their only outstanding global loads (residual tiles requested
// straight into the accumulators) are waited for by the compiler where the accumulators are first used
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// per-block geometry and the LDS tables behind the two staging buffers (built by tile_setup)
struct Tile {
    float *s_sc, *s_ps, *s_sh;
    int *s_pix, *s_cls, *s_nch, *s_kidx, *s_orow, *s_arow;
    int b, rbase, HWo, m0, n0, Kb, Nb, T, pad, msub, nsub;
};

// Decode blockIdx into (image, M block, N block), fill the LDS tables and synchronise the block.  Returns false
// (for the whole block) when the block has no work.
// TABLES = false leaves the per-column scale / post_sub / shift tables to tile_tables(), so that their (dependent)
// gathers need not finish before the first weight loads are issued.
template <int MS, int NS, bool TABLES = true>
__device__ __forceinline__ bool tile_setup(const ImgArgs& p, float* smem, Tile& o) {
    constexpr int BM = MS * 32, BNX = NS * 32;
    constexpr int BUF = (BM + BNX) * BK;
    float* s_sc = smem + 2 * BUF;                          // [BNX] BN scale of the column's channel
    float* s_ps = s_sc + BNX;                              // [BNX] post-ReLU constant of the column's channel
    float* s_sh = s_ps + BNX;                              // [shift_classes][BNX] folded BN shift of the column's channel
    int* s_pix = reinterpret_cast<int*>(s_sh + p.shift_classes * BNX);   // [BM] (oy<<16|ox) or -1
    int* s_cls = s_pix + BM;                               // [BM] border class * BNX
    int* s_nch = s_cls + BM;                               // [BNX] channel of column, -1 zero pad, -2 skip
    int* s_kidx = s_nch + BNX;                             // [cin] (only with k_idx)
    int* s_orow = s_kidx + (p.k_idx ? p.cin : 0);          // packed mode: [BM] destination row (-1 invalid)
    int* s_arow = s_orow + BM;                             // packed mode: [BM][taps] A row per tap (-1 zero)

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    // XCD-aware block order.  Workgroups go round-robin to the 8 XCDs (XCD = bid % 8), each with its own 4 MiB L2.  A
    // "unit" = (image, M block) owns one A tile that all of its ntn N blocks read: the N blocks of a unit get
    // CONSECUTIVE slots of ONE XCD, so they are co-resident and the A tile comes from HBM once and from that XCD's
    // L2 (or an in-flight miss) the other ntn - 1 times.  (With N blocks of a unit a whole batch apart in launch
    // order, the residual / output streams evicted the tile between its uses: conv3 re-fetched A ~8x through the fabric.)
    // Units are image-fastest, so with B % 8 == 0 everything image b touches stays on XCD b % 8.
    // With an output-channel list the trailing N blocks of an image may be empty (Nb is data-dependent): those launches keep
    // N slowest, so that the empty blocks sit at the end of the grid instead of taking every other dispatch slot.
    int nt, unit;
    if (p.n_idx) {
        unit = bid % (p.B * p.mtn);
        nt = bid / (p.B * p.mtn);
        if (nt >= p.ntn) return false;
    } else {
        const int xcd = bid & 7, slot = bid >> 3;
        nt = slot % p.ntn;
        unit = (slot / p.ntn) * 8 + xcd;
        if (unit >= p.B * p.mtn) return false;
    }
    const int b = unit % p.B, mt = unit / p.B;
    // rows of this image: dense mode = its Ho*Wo output pixels; packed mode = its slice of the packed row lists
    int rbase, HWo;
    if (!p.packed) {
        HWo = p.Ho * p.Wo;
        rbase = b * HWo;
    } else if (p.row_prefix) {
        rbase = p.row_prefix[b];
        HWo = p.row_prefix[b + 1] - rbase;
    } else {
        rbase = 0;
        HWo = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap;
    }
    const int m0 = mt * p.bm, n0 = nt * p.bn;
    // the channel-list entries this thread will publish are requested BEFORE the counts they are checked against
    // arrive (the list buffers are allocated for all cin / cout entries): one global round trip instead of two
    int raw_n = 0, raw_k = 0;
    if (p.n_idx && tid < BNX && n0 + tid < p.cout) raw_n = p.n_idx[(size_t)b * p.cout + n0 + tid];
    if (p.k_idx && tid < p.cin) raw_k = p.k_idx[(size_t)b * p.cin + tid];
    const int Kb = p.k_idx ? p.k_cnt[b] : p.cin;
    const int Nb = p.n_idx ? p.n_cnt[b] : p.cout;
    LDN_DCHECK(Kb >= 0 && Kb <= p.cin, 101);                                   // channel counts within the list capacity
    LDN_DCHECK(Nb >= 0 && Nb <= p.cout, 102);
    LDN_DCHECK(!p.n_idx || !(tid < BNX && n0 + tid < Nb) || (raw_n >= 0 && raw_n < p.cout), 103);   // output-channel list entries
    LDN_DCHECK(!p.k_idx || !(tid < Kb) || (raw_k >= 0 && raw_k < p.cin), 104);                     // input-channel list entries
    const int Nb4 = min(round_up(Nb, p.out_split ? 32 : 4), p.cout);
    if (n0 >= Nb4 || m0 >= HWo) return false;
    const int T = p.packed ? p.ksize : p.ksize * p.ksize;   // packed mode: ksize carries the tap count (1 or 9)
    const int pad = p.packed ? (T == 9 ? 1 : 0) : p.ksize >> 1;
    const int msub = ceil_div(min(HWo - m0, p.bm), 32);        // valid m-subtiles (1..MS)
    const int nsub = ceil_div(min(Nb4 - n0, p.bn), 32);        // valid n-subtiles (1..NS)

    for (int i = tid; i < BM; i += 512) {
        const int m = m0 + i;
        int pix = -1, cls = 0, oy = 0, ox = 0;
        const bool valid = m < HWo && i < p.bm;
        if (valid) {
            if (!p.packed) {
                oy = m / p.Wo; ox = m - oy * p.Wo;
            } else if (p.pix_map) {
                const int q = p.pix_map[rbase + m] % (p.Ho * p.Wo);
                oy = q / p.Wo; ox = q - oy * p.Wo;
            }
            pix = (oy << 16) | ox;
            if (p.shift_classes > 1) {
                const int top = oy * p.stride - pad < 0, bot = oy * p.stride + pad >= p.Hi;
                const int lef = ox * p.stride - pad < 0, rig = ox * p.stride + pad >= p.Wi;
                cls = ((top | (bot << 1)) * 4 + (lef | (rig << 1))) * BNX;
            }
        }
        s_pix[i] = pix;
        s_cls[i] = cls;
        if (p.packed) {
            int orow = -1;
            if (valid) {
                orow = p.out_map ? p.out_map[rbase + m] : rbase + m;
                LDN_DCHECK(orow >= 0 && orow < 0x40000000, 105);               // destination rows are non-negative
                if (p.relu == 2 && p.relu_if_neg[rbase + m] < 0) orow |= 0x40000000;   // bit 30: apply ReLU to this row
            }
            s_orow[i] = orow;
        }
    }
    if (p.packed)
        for (int i = tid; i < BM * T; i += 512) {
            const int r = i / T, m = m0 + r;
            s_arow[i] = (m < HWo && r < p.bm) ? (p.a_map ? p.a_map[(size_t)(rbase + m) * T + (i - r * T)] : rbase + m) : -1;
            LDN_DCHECK(s_arow[i] >= -1, 106);                                  // gather rows: -1 (zero row) or a row index
        }
    static_assert(BNX <= 512, "one thread per tile column");
    if (tid < BNX) {
        const int i = tid, j = n0 + i;
        const bool in_block = i < p.bn;
        const int chn = (in_block && j < Nb) ? (p.n_idx ? raw_n : j) : ((in_block && j < Nb4) ? -1 : -2);
        s_nch[i] = chn;
        if (TABLES) {
            s_sc[i] = chn >= 0 ? (p.scale ? p.scale[chn] : 1.f) : 0.f;
            s_ps[i] = (chn >= 0 && p.post_sub) ? p.post_sub[chn] : 0.f;
            for (int c = 0; c < p.shift_classes; ++c) s_sh[c * BNX + i] = chn >= 0 ? p.shift[c * p.cout + chn] : 0.f;
        }
    }
    if (p.k_idx) {
        // positions Kb .. round_up(Kb, 8) name channel 0 (cin is a multiple of 8 whenever a list is given): the last,
        // partial octet then reads valid weight rows against exact-zero activations -- no per-row validity select
        if (tid < Kb) s_kidx[tid] = raw_k;
        for (int i = tid + 512; i < Kb; i += 512) s_kidx[i] = p.k_idx[(size_t)b * p.cin + i];
        if (tid < 8 && Kb + tid < min(round_up(Kb, 8), p.cin)) s_kidx[Kb + tid] = 0;
    }
    __syncthreads();
    o.s_sc = s_sc; o.s_ps = s_ps; o.s_sh = s_sh; o.s_pix = s_pix; o.s_cls = s_cls; o.s_nch = s_nch; o.s_kidx = s_kidx;
    o.s_orow = s_orow; o.s_arow = s_arow;
    o.b = b; o.rbase = rbase; o.HWo = HWo; o.m0 = m0; o.n0 = n0; o.Kb = Kb; o.Nb = Nb; o.T = T; o.pad = pad;
    o.msub = msub; o.nsub = nsub;
    return true;
}

// Per-column epilogue tables of the block (see tile_setup<.., TABLES = false>): filled by `nthreads` threads numbered `idx`.
template <int NS>
__device__ __forceinline__ void tile_tables(const ImgArgs& p, const Tile& t, int idx, int nthreads) {
    constexpr int BNX = NS * 32;
    // (issuing all of these gathers unconditionally and selecting afterwards measured 3-7 % SLOWER: the burst competes with
    //  the producers' first weight loads, which are on the critical path; the conditional form trickles behind them)
    for (int i = idx; i < BNX; i += nthreads) {
        const int chn = t.s_nch[i];
        t.s_sc[i] = chn >= 0 ? (p.scale ? p.scale[chn] : 1.f) : 0.f;
        t.s_ps[i] = (chn >= 0 && p.post_sub) ? p.post_sub[chn] : 0.f;
        for (int c = 0; c < p.shift_classes; ++c) t.s_sh[c * BNX + i] = chn >= 0 ? p.shift[c * p.cout + chn] : 0.f;
    }
}

// Residual operand of tile (mi, nj) in the layout tile_store consumes (4 x 16 bytes per lane).  Issued one tile ahead of
// the store so that the global-load latency is hidden behind the previous tile's transpose and stores (the loads cannot
// be hoisted by the compiler: residual may alias out in the in-place form).
__device__ __forceinline__ void tile_resid(const ImgArgs& p, const Tile& t, int mi, int nj, int lane, f32x4* res) {
    const int trow = lane >> 3, ccol = nj * 32 + (lane & 7) * 4;
    const bool col_ok = t.s_nch[ccol] != -2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = mi * 32 + trow + 8 * it;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col_ok && t.s_pix[row] >= 0) {
            const size_t orow = p.packed ? (size_t)(t.s_orow[row] & 0x3fffffff) : (size_t)t.rbase + t.m0 + row;
            v = *reinterpret_cast<const f32x4*>(p.residual + orow * p.ldr + t.n0 + ccol);
        }
        res[it] = v;
    }
}

// Epilogue of one 32x32 accumulator tile (mi, nj) held by one wave: transpose through the wave's 4 KiB LDS scratch, then
// folded-BN affine / residual / ReLU / post_sub on 4 consecutive channels per lane and 16-byte stores along the channel
// axis; optional fused global-average-pool partials (colsum).
__device__ __forceinline__ void tile_store(const ImgArgs& p, const Tile& t, float* scratch, const f32x16& acc, int mi,
                                           int nj, int lane, const f32x4* res) {
    const int l31 = lane & 31, h = lane >> 5;
    const int trow = lane >> 3, tc4 = (lane & 7) * 4;
    // phase 1: raw accumulators -> scratch (MFMA layout: lane = column, register r = row (r&3) + 8 (r>>2) + 4 h)
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // phase 2: lane = (row trow + 8 it, 4 consecutive columns): affine with the row's border class, residual, ReLU, store.
    // Columns without a channel have scale = shift = 0 in the tables and exact-zero accumulators (their weight rows
    // are staged as zeros), so they come out as 0.
    const int ccol = nj * 32 + tc4;
    if (t.s_nch[ccol] != -2) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(t.s_sc + ccol);
        const f32x4 ps = *reinterpret_cast<const f32x4*>(t.s_ps + ccol);
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = mi * 32 + trow + 8 * it;
            if (t.s_pix[row] < 0) continue;
            const int cls = p.shift_classes > 1 ? t.s_cls[row] : 0;
            const f32x4 sh = *reinterpret_cast<const f32x4*>(t.s_sh + cls + ccol);
            f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (trow + 8 * it) * 32 + tc4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
            size_t orow = (size_t)t.rbase + t.m0 + row;
            bool do_relu = p.relu == 1;
            if (p.packed) {
                const int o = t.s_orow[row];
                do_relu = do_relu || (o & 0x40000000);
                orow = (size_t)(o & 0x3fffffff);
            }
            if (p.residual) v += res[it];
            if (do_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            v -= ps;
            *reinterpret_cast<f32x4*>(p.out + orow * p.ldo + t.n0 + ccol) = v;
            csum += v;
        }
        if (p.colsum) {   // fused global-average-pool partials for the NEXT block's channel masker
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);
            if (trow == 0) {
                const size_t slot = ((size_t)t.b * ceil_div(t.HWo, 32) + (t.m0 >> 5) + mi) * p.cout + t.n0 + ccol;   // dense mode only
                *reinterpret_cast<f32x4*>(p.colsum + slot) = csum;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- epilogue of k_conv_bf3 (the fp32 kernel keeps tile_resid / tile_store above)
// Per-lane description of the 4 output rows (trow + 8 it of m-subtile mi) a lane stores in the epilogue; read from the
// block tables once per m-subtile instead of once per tile (the table reads are dependent LDS round trips).
struct RowInfo {
    int orow[4];    // destination row in out / residual, -1 = no row
    int cls[4];     // border class * BNX (offset into the shift table); bit 30 = apply ReLU to this row
};
constexpr int ROW_RELU = 1 << 30;

__device__ __forceinline__ void tile_rows(const ImgArgs& p, const Tile& t, int mi, int lane, RowInfo& ri) {
    const int trow = lane >> 3;
    int pix[4], orw[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = mi * 32 + trow + 8 * it;
        pix[it] = t.s_pix[row];
        ri.cls[it] = p.shift_classes > 1 ? t.s_cls[row] : 0;
        orw[it] = p.packed ? t.s_orow[row] : 0;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = mi * 32 + trow + 8 * it;
        if (p.relu == 1 || (p.packed && (orw[it] & 0x40000000))) ri.cls[it] |= ROW_RELU;
        ri.orow[it] = pix[it] < 0 ? -1 : (p.packed ? (orw[it] & 0x3fffffff) : t.rbase + t.m0 + row);
    }
}

// Residual operand of tile (mi, nj), 4 x 16 bytes per lane in the layout tile_store_rows consumes.  On gfx9 loads and
// stores share vmcnt and may complete out of order with each other, so a load consumed while stores are pending waits
// for every one of them (one such wait per tile here; requesting several tiles' residuals ahead costs registers the
// 128-VGPR shapes do not have).
__device__ __forceinline__ void tile_resid_rows(const ImgArgs& p, const Tile& t, const RowInfo& ri, int nj, int lane, f32x4* res) {
    const int ccol = nj * 32 + (lane & 7) * 4;
    const bool col_ok = t.s_nch[ccol] != -2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col_ok && ri.orow[it] >= 0) v = *reinterpret_cast<const f32x4*>(p.residual + (size_t)ri.orow[it] * p.ldr + t.n0 + ccol);
        res[it] = v;
    }
}

// Same epilogue as tile_store with the row table hoisted (RowInfo), the LDS reads batched two rows at a time, and
// compiler barriers instead of fences around the transpose: LDS instructions of one wave execute in order, and a
// release fence would also wait for every outstanding global store.
__device__ __forceinline__ void tile_store_rows(const ImgArgs& p, const Tile& t, float* scratch, const f32x16& acc,
                                                const RowInfo& ri, int mi, int nj, int lane, const f32x4* res, bool add_res) {
    const int l31 = lane & 31, h = lane >> 5;
    const int trow = lane >> 3, tc4 = (lane & 7) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[r];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int ccol = nj * 32 + tc4;
    if (t.s_nch[ccol] != -2) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(t.s_sc + ccol);
        const f32x4 ps = *reinterpret_cast<const f32x4*>(t.s_ps + ccol);
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i2 = 0; i2 < 4; i2 += 2) {
            f32x4 v[2], sh[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // the LDS reads of two rows first: one round trip
                v[u] = *reinterpret_cast<const f32x4*>(scratch + (trow + 8 * (i2 + u)) * 32 + tc4);
                sh[u] = *reinterpret_cast<const f32x4*>(t.s_sh + (ri.cls[i2 + u] & ~ROW_RELU) + ccol);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = i2 + u;
                if (ri.orow[it] < 0) continue;
                f32x4 x = v[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = x[e] * sc[e] + sh[u][e];
                if (add_res && p.residual) x += res[it];
                if (ri.cls[it] & ROW_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
                }
                x -= ps;
#if LDN_ABLATE & 16
                if (x[0] == 12345.678f)
#endif
                if (p.out_split) {
                    // this lane's 4 channels are half an octet: hi halves -> bytes [0,16) of the octet's 32, lo halves -> [16,32)
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)x[e];
                        lo[e] = (__bf16)(x[e] - (float)hi[e]);
                    }
                    float* oct = p.out + (size_t)ri.orow[it] * p.ldo + t.n0 + (ccol & ~7) + ((ccol & 4) >> 1);
                    *reinterpret_cast<bf16x4*>(oct) = hi;
                    *reinterpret_cast<bf16x4*>(oct + 4) = lo;
                } else
                *reinterpret_cast<f32x4*>(p.out + (size_t)ri.orow[it] * p.ldo + t.n0 + ccol) = x;
                csum += x;
            }
        }
        if (p.colsum) {   // fused global-average-pool partials for the NEXT block's channel masker
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);
            if (trow == 0) {
                const size_t slot = ((size_t)t.b * ceil_div(t.HWo, 32) + (t.m0 >> 5) + mi) * p.cout + t.n0 + ccol;   // dense mode only
                *reinterpret_cast<f32x4*>(p.colsum + slot) = csum;
            }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// MS x NS = m-subtiles x n-subtiles of 32 per block (two LDS buffers of (MS+NS)*4 KiB).
// BMODE= weight layout / staging path:
//        B_NK  : w[cout][taps][cin] ("n-major"), no K gather.  Tile rows = output channels (gathered through
//                n_idx for free: a row is a pointer), 128 contiguous bytes of K per row            -> LDS-DMA
//        B_KN4 : w[taps][cin][cout] ("k-major"), used whenever the INPUT channels are gathered (k_idx): tile
//                rows = the chunk's 32 packed k positions (row gather, free), columns = output channels,
//                contiguous or gathered in aligned runs of >= 4                                    -> LDS-DMA
//        B_KN2 / B_KN1 : same layout, output channels gathered in aligned pairs / singly           -> VGPRs
//        (k-major keeps every weight fetch inside one or a few 128-byte lines of a single row; the n-major
//         layout with a K gather touched ~2.6x more lines than it used)
// KSKIP= skip the empty 8-wide k groups of a partial chunk (pays when the per-tap K is short).
enum { B_NK = 0, B_KN4 = 1, B_KN2 = 2, B_KN1 = 3 };

template <int MS, int NS, int BMODE, bool KSKIP, int MINW>
__global__ __launch_bounds__(512, MINW) void k_conv_image(const ImgArgs p) {
    static_assert(MS >= 4 && MS + NS <= 18, "LDS budget");
    constexpr int ACC = (MS * NS + 3) / 4;                 // 32x32 accumulators per consumer wave
    constexpr int BM = MS * 32, BNX = NS * 32;
    constexpr int BUF = (BM + BNX) * BK;                   // floats per LDS buffer
    constexpr bool BGLDS = BMODE == B_NK || BMODE == B_KN4;
    constexpr bool KN = BMODE != B_NK;
    constexpr int BSL = BNX / 4;                           // 16-byte slots per k-major B row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Tile t;
    if (!tile_setup<MS, NS>(p, smem, t)) return;
    float *const s_sc = t.s_sc, *const s_ps = t.s_ps, *const s_sh = t.s_sh;
    int *const s_pix = t.s_pix, *const s_cls = t.s_cls, *const s_nch = t.s_nch, *const s_kidx = t.s_kidx;
    int *const s_orow = t.s_orow, *const s_arow = t.s_arow;
    const int tid = threadIdx.x;
    const int b = t.b, rbase = t.rbase, HWo = t.HWo, m0 = t.m0, n0 = t.n0, Kb = t.Kb, T = t.T, pad = t.pad;
    const int msub = t.msub, nsub = t.nsub, ntiles = msub * nsub;
    (void)s_sc; (void)s_ps; (void)s_sh; (void)s_cls; (void)s_orow; (void)rbase; (void)HWo; (void)m0; (void)n0;
#ifdef LDN_TRACE
    unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime(), tr_t0 = __builtin_amdgcn_s_memtime(), tr_bar = 0, tr_mma = 0, tr_iss = 0, tr_a = 0, tr_b = 0;
#endif

    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cpt = ceil_div(Kb, BK);
    const int nch = T * cpt;

    if (wave8 >= 4) {
        // ================================================================ producers (waves 4..7)
        const int wave = wave8 - 4;
        // one wave instruction fills 8 rows x 8 slots (1 KiB); this thread moves physical slot (lane & 7) of row
        // (wave + 4u) * 8 + (lane >> 3), i.e. logical slot qt = slot ^ ((row >> 1) & 7)
        const int rg = lane >> 3, pslot = lane & 7;
        const int qt = pslot ^ (((rg >> 1) + 4 * (wave & 1)) & 7);
        const int kq = qt * 4;                                 // K offset of this thread's slot inside a chunk
        long aoff[MS];   // element offset of this thread's A rows for the current tap, -1 = zero row
        long boff[NS];   // B_NK: element offset of w[ch_n][tap][0], -1 = zero row
        int cur_tap = 0;
        int kr[NS], cn0[NS], cn2[NS];   // k-major: this thread's k row / first and third column channel per slot
        if (KN) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int sid = (wave + 4 * u) * 64 + lane;
                kr[u] = sid / BSL;
                const int j = (sid - kr[u] * BSL) * 4;
                const bool in = j < nsub * 32;
                cn0[u] = in ? s_nch[j] : -1;
                cn2[u] = in ? s_nch[j + 2] : -1;
            }
        }
        // the producers share each SIMD with an MFMA wave that is older (wins issue arbitration): raise their priority
        // so address generation and load issue are never starved (an MFMA needs one issue slot per 64 cycles)
        __builtin_amdgcn_s_setprio(2);
        f32x4 rb[NS];    // VGPR-staged weight slots (only when !BGLDS)
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const int Kb4 = p.k_idx ? round_up(Kb, 4) : p.cin;

        auto set_tap = [&](int tap) {
            const int ksz = p.packed ? 3 : p.ksize;
            const int ky = tap / ksz, kx = tap - ky * ksz;
#pragma unroll
            for (int u = 0; u < MS; ++u) {
                long off = -1;
                if (u < msub) {
                    const int row = (wave + 4 * u) * 8 + rg;
                    if (p.packed) {
                        const int ar = s_arow[row * T + tap];
                        if (ar >= 0) off = (long)ar * p.lda;
                    } else {
                        const int pix = s_pix[row];
                        if (pix >= 0) {
                            const int iy = (pix >> 16) * p.stride + ky - pad, ix = (pix & 0xffff) * p.stride + kx - pad;
                            if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                                off = ((long)(b * p.Hi + iy) * p.Wi + ix) * p.lda;
                        }
                    }
                }
                aoff[u] = off;
            }
            if (!KN) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    long off = -1;
                    if (u < nsub) {
                        const int chn = s_nch[(wave + 4 * u) * 8 + rg];
                        if (chn >= 0) off = ((long)chn * T + tap) * p.cin;
                    }
                    boff[u] = off;
                }
            }
            cur_tap = tap;
        };
        // issue the loads of chunk (current tap, c0) into LDS buffer `buf`
        auto issue = [&](int c0, int buf) {
            const int c = c0 + kq;
            float* base = smem + buf * BUF;
#pragma unroll
            for (int u = 0; u < MS; ++u)
                if (u < msub)
                    glds16((aoff[u] >= 0 && c < Kb4) ? p.a + aoff[u] + c : g_zero16, base + (wave + 4 * u) * 8 * BK);
            if (!KN) {
                // n-major: row = output channel (wave+4u)*8+rg, this thread's slot = K offset kq of the chunk
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (u < nsub)
                        glds16((boff[u] >= 0 && c < p.cin) ? p.w + boff[u] + c : g_zero16,
                               base + (BM + (wave + 4 * u) * 8) * BK);
            } else {
                // k-major: the B tile is [32 k rows][BNX columns]; wave instruction (wave + 4u) covers 64 consecutive
                // 16-byte slots.  Slot -> (k row kr[u], columns) is fixed for the whole block (cn*[u] hoisted); per chunk
                // only the weight ROW changes.  Invalid lanes read the zero line instead of branching.
                int kch[NS];
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const int kpos = min(c0 + kr[u], Kb - 1);
                    kch[u] = p.k_idx ? s_kidx[kpos] : kpos;
                }
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const bool kok = c0 + kr[u] < Kb;
                    const float* wr = p.w + ((long)cur_tap * p.cin + kch[u]) * p.cout;
                    if (BMODE == B_KN4) {
                        glds16((kok && cn0[u] >= 0) ? wr + cn0[u] : g_zero16, base + BM * BK + (wave + 4 * u) * 256);
                    } else if (BMODE == B_KN2) {
                        const float2 lo = *reinterpret_cast<const float2*>((kok && cn0[u] >= 0) ? wr + cn0[u] : g_zero16);
                        const float2 hi = *reinterpret_cast<const float2*>((kok && cn2[u] >= 0) ? wr + cn2[u] : g_zero16);
                        rb[u][0] = lo.x; rb[u][1] = lo.y; rb[u][2] = hi.x; rb[u][3] = hi.y;
                    } else {
                        const int j = (((wave + 4 * u) * 64 + lane) % BSL) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int cn = j < nsub * 32 ? s_nch[j + e] : -1;
                            rb[u][e] = *((kok && cn >= 0) ? wr + cn : g_zero16);
                        }
                    }
                }
            }
        };
        auto bstore = [&](int buf) {
#pragma unroll
            for (int u = 0; u < NS; ++u)
                *reinterpret_cast<f32x4*>(smem + buf * BUF + BM * BK + ((wave + 4 * u) * 64 + lane) * 4) = rb[u];
        };

        if (nch > 0) {
            int tap = 0, c0 = 0;
            auto advance = [&]() {
                c0 += BK;
                if (c0 >= Kb) { c0 = 0; ++tap; if (tap < T) set_tap(tap); }
            };
            set_tap(0);
            issue(0, 0);
            advance();
            for (int ch = 0; ch < nch; ++ch) {
                const int buf = ch & 1;
                LDN_TRACE_T(tr_a)
                if (!BGLDS) bstore(buf);
#ifdef LDN_TRACE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                LDN_TRACE_T(tr_b)
                LDN_TRACE_ADD(tr_mma, tr_a, tr_b)     // producer: waiting for its loads (+ ds_write)
                block_sync();                 // barrier(ch): chunk ch is in LDS; consumers are done with chunk ch-1
                LDN_TRACE_T(tr_a)
                LDN_TRACE_ADD(tr_bar, tr_b, tr_a)     // producer: waiting at the barrier for the consumers
#if !(LDN_ABLATE & 2)
                if (ch + 1 < nch) { issue(c0, buf ^ 1); advance(); }
#else
                if (ch + 1 < nch) { advance(); }
#endif
                LDN_TRACE_T(tr_b)
                LDN_TRACE_ADD(tr_iss, tr_a, tr_b)     // producer: address generation + load issue
            }
            block_sync();                     // matches the consumers' final barrier
        }
#ifdef LDN_TRACE
        if (tid == 256 && g_trace) {
            unsigned long long* r = g_trace + ((size_t)blockIdx.x + gridDim.x) * 8;
            r[0] = tr_t0; r[1] = __builtin_amdgcn_s_memtime(); r[2] = 0; r[3] = 1;
            r[4] = ntiles | (tr_bar << 32); r[5] = (tr_mma << 32) | (tr_iss & 0xffffffffull);
        }
#endif
        return;
    }

    // ==================================================================== consumers (waves 0..3)
    const int wave = wave8;
    const int l31 = lane & 31, h = lane >> 5;
    const int swl = (l31 >> 1) & 7;
    int so[BK / 8];                            // swizzled float offset of k group g for this lane
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) so[g] = ((2 * g + h) ^ swl) * 4;
    f32x16 acc[ACC];
    int a_off[ACC], b_off[ACC];
#pragma unroll
    for (int s = 0; s < ACC; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        const int tt = min(wave + 4 * s, ntiles - 1);   // tile tt -> (m-subtile tt % msub, n-subtile tt / msub)
        const int nj = tt / msub, mi = tt - nj * msub;
        a_off[s] = (mi * 32 + l31) * BK;
        b_off[s] = KN ? nj * 32 + l31 : (BM + nj * 32 + l31) * BK;
    }
    const int my_tiles = ntiles > wave ? (ntiles - wave + 3) / 4 : 0;   // tiles wave, wave+4, ... < ntiles

    if (nch > 0) {
        int cin_chunk = 0;                     // chunk index within the tap (for KSKIP)
        for (int ch = 0; ch < nch; ++ch) {
            const int buf = ch & 1;
            const int kgroups = ceil_div(min(Kb - cin_chunk * BK, BK), 8);
            if (++cin_chunk == cpt) cin_chunk = 0;
            LDN_TRACE_T(tr_a)
            block_sync();                      // barrier(ch)
            LDN_TRACE_T(tr_b)
            LDN_TRACE_ADD(tr_bar, tr_a, tr_b)
            const float* tb = smem + buf * BUF;
            auto read_b = [&](int s, int g) -> f32x4 {
                if (!KN) return *reinterpret_cast<const f32x4*>(tb + b_off[s] + so[g]);
                f32x4 v;   // k-major tile: 4 consecutive k rows, this lane's column
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tb[BM * BK + (g * 8 + h * 4 + q) * BNX + b_off[s]];
                return v;
            };
            f32x4 af = *reinterpret_cast<const f32x4*>(tb + a_off[0] + so[0]);
            f32x4 bf = read_b(0, 0);
#pragma unroll
            for (int s = 0; s < ACC; ++s) {
                if (s < my_tiles) {
#pragma unroll
                    for (int g = 0; g < BK / 8; ++g) {
                        f32x4 an = af, bn = bf;
                        if (g + 1 < BK / 8) {
                            an = *reinterpret_cast<const f32x4*>(tb + a_off[s] + so[g + 1]);
                            bn = read_b(s, g + 1);
                        } else if (s + 1 < ACC) {   // offsets of unused slots are clamped to a valid tile
                            an = *reinterpret_cast<const f32x4*>(tb + a_off[s + 1] + so[0]);
                            bn = read_b(s + 1, 0);
                        }
                        if (!KSKIP || g < kgroups) {
#if LDN_ABLATE & 1
                            asm volatile("" ::"v"(af), "v"(bf));
#else
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bf[q], acc[s], 0, 0, 0);
#endif
                        }
                        // pin the schedule: the fragment reads of the NEXT group first, then this group's 4 MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x100, KN ? 5 : 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        af = an;
                        bf = bn;
                    }
                }
            }
            LDN_TRACE_T(tr_a)
            LDN_TRACE_ADD(tr_mma, tr_b, tr_a)
        }
        block_sync();   // every consumer is done with both buffers before buffer 0 becomes the epilogue scratch
    }

    // ---- epilogue
    float* scratch = smem + wave * (32 * 32);
    f32x4 res[4];
#pragma unroll
    for (int s = 0; s < ACC; ++s) {
        const int tt = wave + 4 * s;
        if (tt >= ntiles) continue;
        const int nj = tt / msub, mi = tt - nj * msub;
        if (p.residual) tile_resid(p, t, mi, nj, lane, res);   // in flight during the tile's LDS transpose
        tile_store(p, t, scratch, acc[s], mi, nj, lane, res);
    }
#ifdef LDN_TRACE
    if (tid == 0 && g_trace) {
        unsigned long long* r = g_trace + (size_t)blockIdx.x * 8;
        r[0] = tr_t0; r[1] = __builtin_amdgcn_s_memtime();
        r[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        r[3] = __builtin_amdgcn_s_memrealtime() - tr_r0;                        // 100 MHz ticks
        r[4] = ntiles | (tr_bar << 32); r[5] = (tr_mma << 32) | (tr_iss & 0xffffffffull);
    }
#endif
}

// ================================================================================================ bf16x3 kernel
// Split-precision variant (math mode 1).  Same block structure, A staging, tables and epilogue as k_conv_image, but
//   * every fp32 operand x is used as bf16 hi + bf16 lo (both round-to-nearest-even, x = hi + lo + O(2^-17 |x|)) and a
//     product is formed as lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 (8 passes each) with fp32 accumulation:
//     5.3x less matrix-pipe time per K than v_mfma_f32_32x32x2_f32;
//   * the WEIGHT tile goes through the producers' VGPRs: they split it once per chunk and write it to LDS as rows of
//     output channels whose 128 bytes hold four k-octets as [8 hi bf16 | 8 lo bf16] (k-major weights are transposed in
//     registers on the way), so a B fragment is two ds_read_b128 and no VALU in the consumers, for all four BMODEs;
//   * the ACTIVATION tile still lands as raw fp32 through LDS-DMA; each consumer wave owns whole m-subtiles (wave grid
//     WM x 4/WM over the block's subtiles), splits its A fragments once per K16 step and reuses them against every
//     n-subtile it owns -- register blocking that keeps both the VALU work and the LDS reads per MFMA low.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? x0[e] : x1[e - 4];
        const __bf16 hb = (__bf16)v;
        hi[e] = hb;
        lo[e] = (__bf16)(v - (float)hb);
    }
}

template <int MS, int NS, int WM, int BMODE, bool KSKIP, int MINW>
__global__ __launch_bounds__(512, MINW) void k_conv_bf3(const ImgArgs p) {
    static_assert(MS >= 1 && MS + NS <= 18 && NS % 2 == 0 && (WM == 1 || WM == 2 || WM == 4), "tile shape");
    constexpr int WN = 4 / WM;                             // consumer wave grid WM (m) x WN (n)
    constexpr int AM = (MS + WM - 1) / WM, CN = (NS + WN - 1) / WN;   // subtiles per wave in m / n
    constexpr int BM = MS * 32, BNX = NS * 32;
    constexpr int BUF = (BM + BNX) * BK;
    constexpr bool KN = BMODE != B_NK;
    constexpr int BSL = BNX / 4;                           // column quads of the weight tile
    constexpr int NBT = KN ? (BSL * 4 + 255) / 256 : NS / 2;   // weight tasks per producer lane and chunk
    // narrow k-major tiles (<= 128 columns, aligned column pairs): tasks of (column pair, octet) instead of (column
    // quad, octet) -- 4 * BNX / 2 <= 256 tasks spread over all four producer waves instead of two
    constexpr bool PAIR = (BMODE == B_KN4 || BMODE == B_KN2) && BNX <= 128;
    constexpr int NPR = BNX / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Tile t;
#ifdef LDN_TRACE
    unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime(), tr_t0 = __builtin_amdgcn_s_memtime(), tr_bar = 0, tr_mma = 0, tr_iss = 0, tr_a = 0, tr_b = 0, tr_pro = 0, tr_epi = 0;
#endif
    if (!tile_setup<MS, NS, false>(p, smem, t)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Kb = t.Kb, T = t.T, msub = t.msub, nsub = t.nsub;
    // ONE K axis over (tap, channel): tap t owns positions [t Kp, (t+1) Kp).  With gathered input channels Kp is the image's
    // K_b rounded to 8 (an octet never straddles taps), so a 3x3 over K_b = 162 channels walks ceil(9 * 168 / 32) = 48 chunks
    // instead of 9 * 6 = 54 -- 12 instead of 18 at stage 1 (K_b ~ 40): staging instructions, not MACs, are what a chunk costs.
    // Shared weights keep every tap padded to the chunk width (their K is a multiple of 32 in every shipped model).
    const int Kp = KN ? max(round_up(Kb, 8), BK) : round_up(p.cin, BK);
    const int Ktot = T * Kp;
    const int nch = Kb > 0 ? ceil_div(Ktot, BK) : 0;
    if (wave8 < 4) {
        // the consumers gather the epilogue tables while the producers' first loads are in flight; one extra
        // workgroup barrier (matched in the producer prologue) publishes them
        tile_tables<NS>(p, t, tid, 256);
        block_sync_lds();
    }

    if (wave8 >= 4) {
        // ================================================================ producers (waves 4..7)
        const int wave = wave8 - 4;
        // ---- A: LDS-DMA of raw fp32 rows, exactly as in k_conv_image
        const int rg = lane >> 3, pslot = lane & 7;
        const int qt = pslot ^ (((rg >> 1) + 4 * (wave & 1)) & 7);
        const int kq = qt * 4;
        long aoffA[MS], aoffB[MS];   // row offsets of the tap the chunk starts in / of the next tap (a chunk spans <= 2 taps: Kp >= 32)
        int tcur = 0;
        const int Kb4 = p.k_idx ? round_up(Kb, 4) : p.cin;
        __builtin_amdgcn_s_setprio(2);
        auto set_tap = [&](int tap, long (&aoff)[MS]) {
            const int ksz = p.packed ? 3 : p.ksize;
            const int ky = tap / ksz, kx = tap - ky * ksz;
#pragma unroll
            for (int u = 0; u < MS; ++u) {
                long off = -1;
                if (u < msub && tap < T) {
                    const int row = (wave + 4 * u) * 8 + rg;
                    if (p.packed) {
                        const int ar = t.s_arow[row * T + tap];
                        if (ar >= 0) off = (long)ar * p.lda;
                    } else {
                        const int pix = t.s_pix[row];
                        if (pix >= 0) {
                            const int iy = (pix >> 16) * p.stride + ky - t.pad, ix = (pix & 0xffff) * p.stride + kx - t.pad;
                            if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                                off = ((long)(t.b * p.Hi + iy) * p.Wi + ix) * p.lda;
                        }
                    }
                }
                aoff[u] = off;
            }
        };
        auto issue_a = [&](int c0, int buf) {
            const int f = c0 + kq;                       // flat K position of this lane's 16-byte slot
            const int bound = (tcur + 1) * Kp;
            const bool nxt = f >= bound;                 // the slot belongs to the next tap
            const int c = f - (nxt ? bound : tcur * Kp); // channel position inside its tap
            float* base = smem + buf * BUF;
#pragma unroll
            for (int u = 0; u < MS; ++u)
                if (u < msub) {
                    const long off = nxt ? aoffB[u] : aoffA[u];
                    glds16((off >= 0 && c < Kb4) ? p.a + off + c : g_zero16, base + (wave + 4 * u) * 8 * BK);
                }
        };

        // ---- B: global -> VGPR -> split -> LDS rows [n][4 octets x (8 hi | 8 lo)], slots XOR-swizzled like the A rows
        f32x4 rb[NBT][KN ? 8 : 2];
        // n-major weights: wave-task (wave + 4u) covers 16 rows x 4 octets; this lane = row rl, octet oct.  Eight
        // consecutive lanes write eight rows with distinct swizzles (conflict-free ds_write_b128).
        const int rl = ((lane & 7) << 1) | (lane >> 5), oct_nk = (lane >> 3) & 3;
        long bbase[NBT];            // B_NK: element offset of w[chn][0][0] for this lane's row, -1 = no row
        int cq[NBT], oct_kn[NBT];   // k-major: column quad / octet of task u
        int cn[NBT][BMODE == B_KN1 ? 4 : (BMODE == B_KN2 ? 2 : 1)];
        float2 rbp[PAIR ? 8 : 1];   // PAIR: the task's 8 k rows x 2 columns
        bool pr_tail = false, kn_tail[NBT];   // staged octet lies entirely behind K: KSKIP consumers never read it
#pragma unroll
        for (int u = 0; u < NBT; ++u) kn_tail[u] = false;
        int pr_oct = -1, pr_col = 0, pr_chn = -1;
        if (PAIR) {
            const int q = wave * 64 + lane;
            pr_oct = q / NPR;
            pr_col = 2 * (q - pr_oct * NPR);
            if (pr_oct >= 4 || pr_col >= nsub * 32) pr_oct = -1;
            pr_chn = pr_oct >= 0 ? t.s_nch[pr_col] : -1;
        }
#pragma unroll
        for (int u = 0; u < NBT; ++u) {
            if (PAIR) break;
            if (!KN) {
                const int row = (wave + 4 * u) * 16 + rl;
                const int chn = row < nsub * 32 ? t.s_nch[row] : -1;
                bbase[u] = chn >= 0 ? (long)chn * T * p.cin : -1;
            } else {
                const int q = wave * 64 + lane + 256 * u;
                oct_kn[u] = q / BSL;
                cq[u] = q - oct_kn[u] * BSL;
                const bool in = oct_kn[u] < 4 && cq[u] * 4 < nsub * 32;
                if (!in) oct_kn[u] = -1;
                if (BMODE == B_KN4) cn[u][0] = in ? t.s_nch[cq[u] * 4] : -1;
                if (BMODE == B_KN2) { cn[u][0] = in ? t.s_nch[cq[u] * 4] : -1; cn[u][1] = in ? t.s_nch[cq[u] * 4 + 2] : -1; }
                if (BMODE == B_KN1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cn[u][e] = in ? t.s_nch[cq[u] * 4 + e] : -1;
                }
            }
        }
        // k-major weight loads: row = gathered input channel of packed position kpos (positions past Kb name channel 0 and
        // meet zero activations), columns = the task's channels (columns without a channel read column 0: their
        // accumulators are multiplied by a zero scale).  No validity selects, 32-bit element offsets.
        auto kn_row = [&](unsigned tbase, const f32x4& lo4, const f32x4& hi4, int kbase, int j) -> unsigned {
            // element offset of the weight row of packed position kbase + j (k_idx is always present in the k-major modes)
            const int kch = p.k_idx ? __float_as_int(j < 4 ? lo4[j] : hi4[j - 4]) : min(kbase + j, p.cin - 1);
            return (tbase + (unsigned)kch) * (unsigned)p.cout;
        };
        const int Kb8 = round_up(Kb, 8);
        auto load_b = [&](int c0) {
            const float* kf = reinterpret_cast<const float*>(t.s_kidx);
            if (PAIR) {
                if (pr_oct >= 0) {
                    const int fbase = c0 + pr_oct * 8;          // flat K position of the octet
                    const int tap = fbase / Kp, kbase = fbase - tap * Kp;
                    const unsigned tbase = (unsigned)tap * (unsigned)p.cin;
                    const bool oct_tail = fbase >= Ktot || kbase >= Kb8;   // octet holds no channel: nothing to fetch
                    pr_tail = (fbase & ~15) >= Ktot;            // its whole K16 step lies behind K: KSKIP consumers skip it
                    if (!oct_tail) {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(kf + kbase), hi4 = *reinterpret_cast<const f32x4*>(kf + kbase + 4);
                        const unsigned col = (unsigned)max(pr_chn, 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            rbp[j] = *reinterpret_cast<const float2*>(p.w + (kn_row(tbase, lo4, hi4, kbase, j) + col));
                    } else if (!(KSKIP && pr_tail)) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) rbp[j] = make_float2(0.f, 0.f);
                    }
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                if (!KN) {
                    const int fb = c0 + oct_nk * 8;
                    const int tap = fb / Kp, c = fb - tap * Kp;
                    const float* src = p.w + bbase[u] + (long)tap * p.cin + c;
                    rb[u][0] = *reinterpret_cast<const f32x4*>((bbase[u] >= 0 && tap < T && c < p.cin) ? src : g_zero16);
                    rb[u][1] = *reinterpret_cast<const f32x4*>((bbase[u] >= 0 && tap < T && c + 4 < p.cin) ? src + 4 : g_zero16);
                } else if (oct_kn[u] >= 0) {
                    const int fbase = c0 + oct_kn[u] * 8;
                    const int tap = fbase / Kp, kbase = fbase - tap * Kp;
                    const unsigned tbase = (unsigned)tap * (unsigned)p.cin;
                    kn_tail[u] = (fbase & ~15) >= Ktot;   // the octet's whole K16 step lies behind K: KSKIP consumers skip it
                    if (fbase >= Ktot || kbase >= Kb8) {  // octet holds no channel: nothing to fetch
                        if (!(KSKIP && kn_tail[u])) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) rb[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        continue;
                    }
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(kf + kbase), hi4 = *reinterpret_cast<const f32x4*>(kf + kbase + 4);
                    if (BMODE == B_KN4) {
                        const unsigned col = (unsigned)max(cn[u][0], 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            rb[u][j] = *reinterpret_cast<const f32x4*>(p.w + (kn_row(tbase, lo4, hi4, kbase, j) + col));
                    } else if (BMODE == B_KN2) {
                        const unsigned c0l = (unsigned)max(cn[u][0], 0), c1l = (unsigned)max(cn[u][1], 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned ro = kn_row(tbase, lo4, hi4, kbase, j);
                            const float2 lo = *reinterpret_cast<const float2*>(p.w + (ro + c0l));
                            const float2 hi = *reinterpret_cast<const float2*>(p.w + (ro + c1l));
                            rb[u][j][0] = lo.x; rb[u][j][1] = lo.y; rb[u][j][2] = hi.x; rb[u][j][3] = hi.y;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned ro = kn_row(tbase, lo4, hi4, kbase, j);
#pragma unroll
                            for (int e = 0; e < 4; ++e) rb[u][j][e] = p.w[ro + (unsigned)max(cn[u][e], 0)];
                        }
                    }
                }
            }
        };
        auto store_b = [&](int buf) {
            float* bt = smem + buf * BUF + BM * BK;
            if (PAIR) {
                if (pr_oct >= 0 && !(KSKIP && pr_tail)) {
                    const float* f = reinterpret_cast<const float*>(rbp);   // f[2 j + e] = row j, column e
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x4 x0 = {f[e], f[2 + e], f[4 + e], f[6 + e]};
                        const f32x4 x1 = {f[8 + e], f[10 + e], f[12 + e], f[14 + e]};
                        bf16x8 hi, lo;
                        split_bf16(x0, x1, hi, lo);
                        const int row = pr_col + e, sw = (pr_col >> 1) & 7;
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * pr_oct) ^ sw) * 4) = hi;
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * pr_oct + 1) ^ sw) * 4) = lo;
                    }
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                if (!KN) {
                    const int row = (wave + 4 * u) * 16 + rl;
                    if (row < nsub * 32) {
                        bf16x8 hi, lo;
                        split_bf16(rb[u][0], rb[u][1], hi, lo);
                        const int sw = lane & 7;                                    // (row >> 1) & 7
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * oct_nk) ^ sw) * 4) = hi;
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * oct_nk + 1) ^ sw) * 4) = lo;
                    }
                } else if (oct_kn[u] >= 0 && !(KSKIP && kn_tail[u])) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // column e of the quad: its 8 k values sit in rb[u][0..7][e]
                        const f32x4 x0 = {rb[u][0][e], rb[u][1][e], rb[u][2][e], rb[u][3][e]};
                        const f32x4 x1 = {rb[u][4][e], rb[u][5][e], rb[u][6][e], rb[u][7][e]};
                        bf16x8 hi, lo;
                        split_bf16(x0, x1, hi, lo);
                        const int row = cq[u] * 4 + e, sw = (row >> 1) & 7;
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * oct_kn[u]) ^ sw) * 4) = hi;
                        *reinterpret_cast<bf16x8*>(bt + row * BK + ((2 * oct_kn[u] + 1) ^ sw) * 4) = lo;
                    }
                }
            }
        };

        if (nch > 0) {
            // two cursors: A is DMA'd one chunk ahead of the consumers, B is loaded into VGPRs two chunks ahead
            int c0_a = 0, c0_b = 0;      // flat K positions
            auto adv_a = [&]() {
                c0_a += BK;
                if (c0_a >= (tcur + 1) * Kp) {   // the next chunk starts in the next tap
                    ++tcur;
#pragma unroll
                    for (int u = 0; u < MS; ++u) aoffA[u] = aoffB[u];
                    set_tap(tcur + 1, aoffB);
                }
            };
            auto adv_b = [&]() { c0_b += BK; };
            set_tap(0, aoffA);
            set_tap(1, aoffB);
            load_b(0);
            adv_b();
            issue_a(0, 0);
            adv_a();
            block_sync();    // matches the consumers' table barrier (vmcnt(0) here costs nothing: store_b(0) needs B(0) anyway)
            store_b(0);
            if (nch > 1) {   // both buffers are free at the start: chunk 1 is staged behind chunk 0, before barrier(0)
                load_b(c0_b);
                adv_b();
                issue_a(c0_a, 1);
                adv_a();
                store_b(1);
                if (nch > 2) { load_b(c0_b); adv_b(); }
            }
            for (int ch = 0; ch < nch; ++ch) {
                const int buf = ch & 1;
                LDN_TRACE_T(tr_a)
#ifdef LDN_TRACE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                LDN_TRACE_T(tr_b)
                LDN_TRACE_ADD(tr_mma, tr_a, tr_b)     // producer: waiting for its own loads
                block_sync();                 // barrier(ch): chunk ch is in LDS, B(ch+1) is in VGPRs, buffer buf^1 is free
                LDN_TRACE_T(tr_a)
                LDN_TRACE_ADD(tr_bar, tr_b, tr_a)     // producer: waiting at the barrier for the consumers
                if (ch >= 1 && ch + 1 < nch) {
                    // Order matters: the compiler must assume that an LDS access may alias an LDS-DMA load in flight and puts
                    // s_waitcnt vmcnt(0) in front of the first ds_write / ds_read that follows one.  With the A tile's DMA
                    // issued FIRST, the weight split/store (ds_write) and the channel-list reads of load_b (ds_read) each
                    // waited for it, and only then were the weight loads issued: two memory latencies in series per chunk.
                    // LDS work first, then every load of the iteration back to back, one wait at the barrier.
#if !(LDN_ABLATE & 8)
                    store_b(buf ^ 1);
#endif
#if !(LDN_ABLATE & 4)
                    if (ch + 2 < nch) { load_b(c0_b); adv_b(); }
#endif
#if !(LDN_ABLATE & 2)
                    issue_a(c0_a, buf ^ 1);
#endif
                    adv_a();
                }
                LDN_TRACE_T(tr_b)
                LDN_TRACE_ADD(tr_iss, tr_a, tr_b)     // producer: DMA issue + weight split/store + weight load issue
            }
            block_sync();
        } else {
            block_sync();    // no K work at all: still match the consumers' table barrier
        }
#ifdef LDN_TRACE
        if (tid == 256 && g_trace) {
            unsigned long long* r = g_trace + ((size_t)blockIdx.x + gridDim.x) * 8;
            r[0] = tr_t0; r[1] = __builtin_amdgcn_s_memtime(); r[2] = 0; r[3] = 1;
            r[4] = (msub * nsub) | (tr_bar << 32); r[5] = (tr_mma << 32) | (tr_iss & 0xffffffffull);
        }
#endif
        return;
    }

    // ==================================================================== consumers (waves 0..3)
    const int wave = wave8;
    const int l31 = lane & 31, h = lane >> 5;
    const int swl = (l31 >> 1) & 7;
    const int wm = wave % WM, wn = wave / WM;
    const int my_am = msub > wm ? (msub - wm + WM - 1) / WM : 0;    // m-subtiles wm, wm + WM, ...
    const int my_cn = nsub > wn ? (nsub - wn + WN - 1) / WN : 0;    // n-subtiles wn, wn + WN, ...
    f32x16 acc[AM][CN];
    // Weights that carry the BN scale (scale == NULL): the residual tile is requested straight into the accumulators
    // here, in MFMA C layout (register r of lane (l31, h) = row (r&3) + 8 (r>>2) + 4 h, column l31: every wave
    // instruction fetches two whole 128-byte lines).  The loads fly during the producers' prologue and are only waited
    // for at the first MFMA that touches the tile; the epilogue then has no load -> store dependency (on gfx9 loads
    // and stores share vmcnt, so a residual consumed there waits for every store issued before it).
    const bool res_acc = p.residual != nullptr && p.scale == nullptr;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        long roff[16];
        if (res_acc && a < my_am) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm + WM * a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int pix = t.s_pix[row];
                const int orw = p.packed ? (t.s_orow[row] & 0x3fffffff) : t.rbase + t.m0 + row;
                roff[r] = pix < 0 ? -1 : (long)orw * p.ldr + t.n0;
            }
        }
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const int col = (wn + WN * c) * 32 + l31;
            const bool col_ok = res_acc && a < my_am && c < my_cn && t.s_nch[col] != -2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
#if !(LDN_ABLATE & 32)
                if (col_ok && roff[r] >= 0) v = p.residual[roff[r] + col];
#endif
                acc[a][c][r] = v;
            }
        }
    }
    const int a_row = (wm * 32 + l31) * BK, b_row = (BM + wn * 32 + l31) * BK;
    int sl[BK / 16][2];                        // swizzled float offsets of this lane's two slots in K16 step ks
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        sl[ks][0] = ((4 * ks + 2 * h) ^ swl) * 4;
        sl[ks][1] = ((4 * ks + 2 * h + 1) ^ swl) * 4;
    }

    if (nch > 0) {
        for (int ch = 0; ch < nch; ++ch) {
            const int buf = ch & 1;
            const int kgroups = ceil_div(min(Ktot - ch * BK, BK), 8);   // octets of this chunk that lie inside K (only the last chunk is partial)
            LDN_TRACE_T(tr_a)
            block_sync_lds();                  // barrier(ch)
            LDN_TRACE_T(tr_b)
            LDN_TRACE_ADD(tr_bar, tr_a, tr_b)
#ifdef LDN_TRACE
            if (ch == 0) tr_pro = tr_b - tr_t0;
#endif
            const float* tb = smem + buf * BUF;
            // lane's octet of K16 step ks: raw A floats [8 oct, 8 oct + 8) = slots 2 oct, 2 oct + 1; the B row holds the
            // octet's hi half in slot 2 oct and its lo half in slot 2 oct + 1.  All AM m-subtiles are computed (rows of
            // subtiles beyond msub hold stale data; their accumulators are never stored); n-subtiles beyond my_cn are
            // skipped by a wave-uniform branch, and the B fragment of subtile c + 1 is requested before the MFMAs of c.
            f32x4 ar[AM][2];
            bf16x8 bh, bl;
#pragma unroll
            for (int a = 0; a < AM; ++a) {
                ar[a][0] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (WM * 32 * BK) + sl[0][0]);
                ar[a][1] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (WM * 32 * BK) + sl[0][1]);
            }
            bh = *reinterpret_cast<const bf16x8*>(tb + b_row + sl[0][0]);
            bl = *reinterpret_cast<const bf16x8*>(tb + b_row + sl[0][1]);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (KSKIP && 2 * ks >= kgroups) break;
                bf16x8 ah[AM], al[AM];
#pragma unroll
                for (int a = 0; a < AM; ++a) split_bf16(ar[a][0], ar[a][1], ah[a], al[a]);
                if (ks + 1 < BK / 16) {
#pragma unroll
                    for (int a = 0; a < AM; ++a) {
                        ar[a][0] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (WM * 32 * BK) + sl[ks + 1][0]);
                        ar[a][1] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (WM * 32 * BK) + sl[ks + 1][1]);
                    }
                }
#pragma unroll
                for (int c = 0; c < CN; ++c) {
                    if (c < my_cn) {
                        const bf16x8 ch_ = bh, cl_ = bl;
                        if (c + 1 < CN) {       // rows beyond nsub * 32 exist in LDS (stale): harmless to read
                            bh = *reinterpret_cast<const bf16x8*>(tb + b_row + (c + 1) * (WN * 32 * BK) + sl[ks][0]);
                            bl = *reinterpret_cast<const bf16x8*>(tb + b_row + (c + 1) * (WN * 32 * BK) + sl[ks][1]);
                        }
#pragma unroll
                        for (int a = 0; a < AM; ++a) {
#if LDN_ABLATE & 1
                            asm volatile("" ::"v"(al[a]), "v"(ah[a]), "v"(ch_), "v"(cl_));
                            continue;
#endif
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], ch_, acc[a][c], 0, 0, 0);
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], cl_, acc[a][c], 0, 0, 0);
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], ch_, acc[a][c], 0, 0, 0);
                        }
                    }
                }
                if (ks + 1 < BK / 16) {
                    bh = *reinterpret_cast<const bf16x8*>(tb + b_row + sl[ks + 1][0]);
                    bl = *reinterpret_cast<const bf16x8*>(tb + b_row + sl[ks + 1][1]);
                }
            }
            LDN_TRACE_T(tr_a)
            LDN_TRACE_ADD(tr_mma, tr_b, tr_a)
        }
        block_sync_lds();   // every consumer is done with both buffers before buffer 0 becomes the epilogue scratch
    }
    LDN_TRACE_T(tr_epi)

    float* scratch = smem + wave * (32 * 32);
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        if (a >= my_am) break;
        const int mi = wm + WM * a;
        RowInfo ri;                      // the row table of the m-subtile is read once for all its n-subtiles
        tile_rows(p, t, mi, lane, ri);
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            if (c >= my_cn) break;
            f32x4 res[4];
            if (p.residual && !res_acc) tile_resid_rows(p, t, ri, wn + WN * c, lane, res);   // in flight during the tile's transpose
            tile_store_rows(p, t, scratch, acc[a][c], ri, mi, wn + WN * c, lane, res, !res_acc);
        }
    }
#ifdef LDN_TRACE
    if (tid == 0 && g_trace) {
        unsigned long long* r = g_trace + (size_t)blockIdx.x * 8;
        r[0] = tr_t0; r[1] = __builtin_amdgcn_s_memtime();
        r[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        r[3] = __builtin_amdgcn_s_memrealtime() - tr_r0;                        // 100 MHz ticks
        r[4] = (msub * nsub) | (tr_bar << 32); r[5] = (tr_mma << 32) | (tr_iss & 0xffffffffull);
        r[6] = tr_pro; r[7] = r[1] - tr_epi;
    }
#endif
}

// ================================================================================================ streaming 1x1 kernel
// k_conv1x1_stream -- the wide 1x1 convolutions that close a bottleneck (conv3: gathered input channels, all output
// channels, residual + ReLU) and the dense 1x1 projections, bf16x3 arithmetic.  Measured on k_conv_bf3 (tools/ablate):
// with ALL of its K-loop loads, residual loads and output stores removed a stage-3 conv3 launch still took 131 of 184 us --
// its 4096 short blocks (5 K chunks each) spend their time in per-block latency chains (lists -> weights -> first chunk,
// two-deep staging, transposing epilogue), not on bandwidth.  This kernel removes the chains instead of the bytes:
//   * ONE persistent workgroup owns up to 1024 consecutive output rows of an image and walks ALL of their
//     (M block of 128 rows) x (N tile of 128 columns) x (K chunk of 32) steps as one flat sequence: set-up, channel list
//     and row tables are paid once, and the staging pipeline never drains between tiles.
//   * the producers only issue LDS-DMA: raw fp32 A tiles into a ring of FOUR 16 KiB slots, raw fp32 weight tiles into a
//     ring of THREE.  Every iteration issues exactly 8 DMA instructions per producer wave and then waits with
//     s_waitcnt vmcnt(8) -- not 0 -- so the chunk issued two iterations ago has landed while two more are in flight:
//     memory latency is covered inside the block, not by co-resident blocks.  No load returns into a register, so no
//     register is ever "in flight" across a loop edge.
//   * the landed weight tile is converted LDS -> LDS by the producers (k-major tiles are transposed on the way) into the
//     bf16 hi|lo rows the consumers read (double buffer); A is split by the consumers as in k_conv_bf3.
//   * the residual tile is requested straight into the accumulators at the start of every tile (weights carry the BN
//     scale, scale == NULL), and the epilogue stores straight from the MFMA C layout (lane = column: every wave store
//     covers two whole 128-byte lines): no LDS scratch, no transposes, no load -> store dependency.
// All producer LDS traffic is inline asm: the compiler must assume that an LDS access may alias an LDS-DMA load in
// flight and would otherwise put s_waitcnt vmcnt(0) in front of every one of them.
// BMODE: B_KN4 = k-major weights [cin][cout] with a per-image input-channel list (conv3 of channel / both mode),
//        B_NK  = n-major weights [cout][cin], no lists (conv3 of spatial / layer mode, projection shortcuts, dense execution).
#ifndef LDN_STREAM_NT
#define LDN_STREAM_NT 3   // 1: residual loads non-temporal, 2: output stores non-temporal.  Both stream through the XCD's L2 exactly
#endif                    // once; without the hint they evict the A tile / weight rows every N tile re-reads (measured: 20.0 -> 19.6 ms per step)
constexpr int ST_BM = 128, ST_BN = 128;
constexpr int ST_TILE = ST_BM * BK;                       // 4096 floats = 16 KiB: one A slot, one raw-B slot, one converted-B slot
constexpr int ST_A_SLOTS = 4, ST_B_SLOTS = 3;
constexpr int ST_OFF_BRAW = ST_A_SLOTS * ST_TILE, ST_OFF_CB = ST_OFF_BRAW + ST_B_SLOTS * ST_TILE;
constexpr int ST_LDS_FLOATS = ST_OFF_CB + 2 * ST_TILE;    // 9 slots = 144 KiB
constexpr int ST_MAXROWS = 1024;                          // rows per workgroup (8 M blocks)
constexpr int ST_ROW_RELU = 1 << 30;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 8, "add the immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_write16(unsigned addr, const bf16x8& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int lds_read4_wait(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// eight 8-byte / two 16-byte reads with ONE wait (the outputs are only valid after the asm block)
__device__ __forceinline__ void lds_read8x8_wait(unsigned addr, f32x2 (&f)[8]) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\t"
                 "ds_read_b64 %3, %8 offset:1536\n\tds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\t"
                 "ds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
                 : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read4x4_wait(unsigned addr, int (&k)[4]) {   // four dwords 8 bytes apart, one wait
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:8\n\tds_read_b32 %2, %4 offset:16\n\tds_read_b32 %3, %4 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read2x16_wait(unsigned a0, unsigned a1, f32x4& x0, f32x4& x1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1) : "memory");
}

template <int BMODE>
__global__ __launch_bounds__(512, 2) void k_conv1x1_stream(const ImgArgs p) {
    constexpr bool KN = BMODE != B_NK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_arow = reinterpret_cast<int*>(smem + ST_LDS_FLOATS);   // [ST_MAXROWS] A row (-1 = zero row)
    int* s_orow = s_arow + ST_MAXROWS;                            // [ST_MAXROWS] out row | ROW_RELU (-1 = none)
    int* s_kidx = s_orow + ST_MAXROWS;                            // [cin + 32] (KN only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // image-fastest block order: with B % 8 == 0 everything image b touches stays on XCD b % 8
    // (p.mtn row blocks per image) x (p.ntn column ranges of p.bn N tiles each: only used when there are too few rows to fill the chip)
    const int b = blockIdx.x % p.B, jrest = blockIdx.x / p.B;
    const int jblk = jrest % p.mtn, nt_base = (jrest / p.mtn) * p.bn;
    int rbase, HWo;
    if (!p.packed) { HWo = p.Ho * p.Wo; rbase = b * HWo; }
    else if (p.row_prefix) { rbase = p.row_prefix[b]; HWo = p.row_prefix[b + 1] - rbase; }
    else { rbase = 0; HWo = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap; }
    const int r0 = jblk * p.bm;                       // first row of this workgroup inside the image (multiple of 128)
    if (r0 >= HWo) return;
    const int nrows = min(p.bm, HWo - r0);
    const int nmb = ceil_div(nrows, ST_BM);
    int raw_k = 0;
    if (KN && tid < p.cin) raw_k = p.k_idx[(size_t)b * p.cin + tid];       // requested before the count it is checked against
    const int Kb = KN ? p.k_cnt[b] : p.cin;
    const int Kb4 = KN ? round_up(Kb, 4) : p.cin;
    const int cpt = max(1, ceil_div(Kb, BK));         // Kb == 0: one all-zero chunk, so that every tile still has an epilogue
    const int ntn = min(p.bn, p.cout / ST_BN - nt_base);   // N tiles of this workgroup: nt_base .. nt_base + ntn - 1
    const int G = nmb * ntn * cpt;                    // K chunks of this workgroup, flattened over (M block, N tile, chunk)

    for (int i = tid; i < nmb * ST_BM; i += 512) {
        const int m = r0 + i;
        int ar = -1, orw = -1;
        if (i < nrows) {
            if (!p.packed) {
                const int oy = m / p.Wo, ox = m - oy * p.Wo;
                ar = (b * p.Hi + oy * p.stride) * p.Wi + ox * p.stride;
                orw = rbase + m;
            } else {
                ar = p.a_map ? p.a_map[rbase + m] : rbase + m;
                orw = p.out_map ? p.out_map[rbase + m] : rbase + m;
            }
            if (p.relu == 1 || (p.relu == 2 && p.relu_if_neg[rbase + m] < 0)) orw |= ST_ROW_RELU;
        }
        s_arow[i] = ar;
        s_orow[i] = orw;
    }
    if (KN) {
        // list positions Kb .. cpt * 32 name channel 0: they meet exact-zero activations (A columns >= roundup4(Kb) are staged
        // from the zero line, columns Kb .. roundup4(Kb) were written as zeros by the producing conv)
        if (tid < Kb) s_kidx[tid] = raw_k;
        for (int i = tid + 512; i < Kb; i += 512) s_kidx[i] = p.k_idx[(size_t)b * p.cin + i];
        if (tid < 32 && Kb + tid < cpt * BK) s_kidx[Kb + tid] = 0;
    }
    __syncthreads();

    if (wave8 >= 4) {
        // ================================================================ producers (waves 4..7)
        const int wave = wave8 - 4;
#ifndef LDN_STREAM_PRIO
#define LDN_STREAM_PRIO 2
#endif
        __builtin_amdgcn_s_setprio(LDN_STREAM_PRIO);
        const unsigned lds_arow = lds_addr(s_arow), lds_kidx = lds_addr(s_kidx), lds0 = lds_addr(smem);
        // every load address is "base + (valid ? offset : offset of the zero line)": one instruction per slot, never a branch
        const long zoff_a = g_zero16 - p.a, zoff_w = g_zero16 - p.w;
        // ---- A (and the n-major weight tile): one wave instruction = 8 rows x 128 B; this lane moves physical slot
        //      (lane & 7) of row (wave + 4u) * 8 + (lane >> 3), whose logical slot is slot ^ ((row >> 1) & 7)
        const int rg = lane >> 3, pslot = lane & 7;
        const int kq = (pslot ^ (((rg >> 1) + 4 * (wave & 1)) & 7)) * 4;
        long aoff[4];
        auto set_mb = [&](int mb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ar = lds_read4_wait(lds_arow + 4u * (unsigned)(mb * ST_BM + (wave + 4 * u) * 8 + rg));
                aoff[u] = ar < 0 ? -1 : (long)ar * p.lda;
            }
        };
        int c_mb = 0, c_nt = 0, c_c0 = 0, issued = 0;   // issue cursor; past G the same instructions run on the zero line
        auto issue = [&](int aslot, int bslot) {
            float* abase = smem + aslot * ST_TILE;
            float* bbase = smem + ST_OFF_BRAW + bslot * ST_TILE;
            const bool live = issued < G;
            ++issued;
            const int c = c_c0 + kq;
            const int n0 = (nt_base + c_nt) * ST_BN;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                glds16(p.a + ((!(LDN_ABLATE & 2) && live && aoff[u] >= 0 && c < Kb4) ? aoff[u] + c : zoff_a), abase + (wave + 4 * u) * 8 * BK);
            if constexpr (KN) {
                // raw tile [32 k rows][128 columns]: instruction q = wave + 4u covers k rows 2q, 2q+1 (512 B each)
                // A wave DMAs exactly the rows it converts itself (octet `wave` = instructions 4 wave .. 4 wave + 3): its own
                // vmcnt is the only thing that orders its LDS reads behind a DMA -- there is no barrier before convert()
                const int kr = lane >> 5, col4 = (lane & 31) * 4;
                int kchs[4];   // list entries c_c0 + 8 wave + 2u + kr, u = 0..3: 2 entries = 8 bytes apart
                lds_read4x4_wait(lds_kidx + 4u * (unsigned)(c_c0 + 8 * wave + kr), kchs);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = 4 * wave + u;
                    const int kch = kchs[u];
                    glds16(p.w + ((!(LDN_ABLATE & 4) && live) ? (long)((unsigned)kch * (unsigned)p.cout + (unsigned)(n0 + col4)) : zoff_w), bbase + q * 256);
                }
            } else {
                // raw tile [128 n rows][32 k]: exactly the A pattern (swizzled 128-byte rows); k positions >= cin meet zero
                // activations, so out-of-range slots may read anything finite: the zero line
                // a wave DMAs the 8-row groups it converts itself (rows (wave + 4t) * 16 .. + 15, t = 0, 1): see the k-major case
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int grp = 2 * (wave + 4 * (u >> 1)) + (u & 1);            // rows grp * 8 .. grp * 8 + 7
                    const int cb_ = c_c0 + (pslot ^ (((rg >> 1) + 4 * (u & 1)) & 7)) * 4;   // logical slot of this lane in that group
                    const long roff = (long)(n0 + grp * 8 + rg) * p.cin + cb_;
                    glds16(p.w + ((!(LDN_ABLATE & 4) && live && cb_ < p.cin) ? roff : zoff_w), bbase + grp * 8 * BK);
                }
            }
            if (issued < G) {   // advance the cursor (it parks on the last chunk once the sequence is exhausted)
                c_c0 += BK;
                if (c_c0 >= cpt * BK) {
                    c_c0 = 0;
                    if (++c_nt == ntn) { c_nt = 0; ++c_mb; set_mb(c_mb); }
                }
            }
        };
        // landed raw weight tile -> bf16 hi|lo rows [n][4 octets x (8 hi | 8 lo)], slots XOR-swizzled like the A rows
        const int rl = ((lane & 7) << 1) | (lane >> 5), oct_nk = (lane >> 3) & 3;
        auto convert = [&](int bslot, int cslot) {
            const unsigned raw = lds0 + 4u * (unsigned)(ST_OFF_BRAW + bslot * ST_TILE);
            const unsigned cb = lds0 + 4u * (unsigned)(ST_OFF_CB + cslot * ST_TILE);
            const int sw = lane & 7;
            if constexpr (KN) {
                // task (octet = wave, column pair = lane): 8 k rows x 2 columns, transposed in registers
                f32x2 f[8];
                lds_read8x8_wait(raw + (unsigned)(wave * 8 * 512 + lane * 8), f);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4 x0 = {f[0][e], f[1][e], f[2][e], f[3][e]};
                    const f32x4 x1 = {f[4][e], f[5][e], f[6][e], f[7][e]};
                    bf16x8 hi, lo;
                    split_bf16(x0, x1, hi, lo);
                    const int row = 2 * lane + e;     // (row >> 1) & 7 == lane & 7
                    lds_write16(cb + 4u * (unsigned)(row * BK + ((2 * wave) ^ sw) * 4), hi);
                    lds_write16(cb + 4u * (unsigned)(row * BK + ((2 * wave + 1) ^ sw) * 4), lo);
                }
            } else {
                // two (row, octet) tasks per lane: row = (wave + 4t) * 16 + rl ((row >> 1) & 7 == lane & 7), 8 consecutive k
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = (wave + 4 * t) * 16 + rl;
                    f32x4 x0, x1;
                    lds_read2x16_wait(raw + 4u * (unsigned)(row * BK + ((2 * oct_nk) ^ sw) * 4),
                                      raw + 4u * (unsigned)(row * BK + ((2 * oct_nk + 1) ^ sw) * 4), x0, x1);
                    bf16x8 hi, lo;
                    split_bf16(x0, x1, hi, lo);
                    lds_write16(cb + 4u * (unsigned)(row * BK + ((2 * oct_nk) ^ sw) * 4), hi);
                    lds_write16(cb + 4u * (unsigned)(row * BK + ((2 * oct_nk + 1) ^ sw) * 4), lo);
                }
            }
        };
        // iteration i (i = -3 .. G-1): [barrier(i)] ; chunk i+1 has landed ; weights(i+1): raw -> converted buffer (i+1) & 1 ;
        // issue chunk i+3 ; publish.  A(i+3) overwrites chunk i-1, which the consumers left before barrier(i); the raw
        // weight slot of chunk i+3 was converted at iteration i-1.
        set_mb(0);
        int tile_cc = 0;
        for (int i = -3; i < G; ++i) {
            if (i >= 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#ifdef LDN_STREAM_SAFE
            wait_vmcnt<0>();
#else
            wait_vmcnt<8>();      // the 8 DMA instructions of iteration i-1 (chunk i+2) may still fly; chunk i+1 is in
#endif
#if !(LDN_ABLATE & 8)
            if (i >= -1 && i + 1 < G) convert((i + 1) % ST_B_SLOTS, (i + 1) & 1);
#endif
            issue((i + 3) & 3, (i + 3) % ST_B_SLOTS);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i >= 0 && ++tile_cc == cpt) {   // chunk i closes a tile: barrier E of the consumers' epilogue
                tile_cc = 0;
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        wait_vmcnt<0>();     // no LDS-DMA may be in flight when the workgroup's LDS is released
        return;
    }

    // ==================================================================== consumers (waves 0..3): 2 x 2 wave grid
    const int wave = wave8;
    const int l31 = lane & 31, h = lane >> 5;
    const int swl = (l31 >> 1) & 7;
    const int wm = wave & 1, wn = wave >> 1;
    const int a_row = (wm * 32 + l31) * BK, b_row = (wn * 32 + l31) * BK;
    int sl[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        sl[ks][0] = ((4 * ks + 2 * h) ^ swl) * 4;
        sl[ks][1] = ((4 * ks + 2 * h + 1) ^ swl) * 4;
    }
    const bool res_acc = p.residual != nullptr;
    f32x16 acc[2][2];
    int g = 0;
    for (int mb = 0; mb < nmb; ++mb) {
        for (int nt = 0; nt < ntn; ++nt) {
            const int n0 = (nt_base + nt) * ST_BN;
            // ---- tile start: the residual of the whole tile is requested now, 16 bytes per lane in the layout the epilogue
            //      stores in (lane = row trow + 8 it, 4 consecutive channels), and only consumed a K loop later: by then the
            //      loads AND the previous tile's stores have long completed (on gfx9 loads and stores share vmcnt, so a
            //      load consumed while stores are pending waits for every one of them)
            const int trow = lane >> 3, tc4 = (lane & 7) * 4;
            f32x4 res[2][2][4];
            int orw[2][4];
            f32x4 sh4[2], sc4[2], ps4[2];   // per-column epilogue constants: requested here for the same reason
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ccol = n0 + (wn + 2 * c) * 32 + tc4;
                sh4[c] = *reinterpret_cast<const f32x4*>(p.shift + ccol);
                sc4[c] = *reinterpret_cast<const f32x4*>(p.scale ? p.scale + ccol : g_one16);
                ps4[c] = *reinterpret_cast<const f32x4*>(p.post_sub ? p.post_sub + ccol : g_zero16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int it = 0; it < 4; ++it) orw[a][it] = s_orow[mb * ST_BM + (wm + 2 * a) * 32 + trow + 8 * it];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ccol = n0 + (wn + 2 * c) * 32 + tc4;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        // rows without a pixel (and launches without a residual) read the zero line: no branches
                        const float* src = (!(LDN_ABLATE & 32) && res_acc && orw[a][it] >= 0)
                                               ? p.residual + (size_t)(orw[a][it] & (ST_ROW_RELU - 1)) * p.ldr + ccol : g_zero16;
#if LDN_STREAM_NT & 1
                        res[a][c][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
#else
                        res[a][c][it] = *reinterpret_cast<const f32x4*>(src);
#endif
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
                }
            }
            // ---- K loop of the tile
            for (int cc = 0; cc < cpt; ++cc, ++g) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // barrier(g): A(g) is in slot g % 4, its weights in buffer g % 2
                asm volatile("" ::: "memory");
#if LDN_ABLATE & 1
                continue;
#endif
                const float* tb = smem + (g & 3) * ST_TILE;                 // A slot of chunk g
                const float* tw = smem + ST_OFF_CB + (g & 1) * ST_TILE;     // converted weights of chunk g
                const int krem = Kb - cc * BK;         // K positions of this chunk that hold data
                f32x4 ar[2][2];
                bf16x8 bh, bl;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    ar[a][0] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (64 * BK) + sl[0][0]);
                    ar[a][1] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (64 * BK) + sl[0][1]);
                }
                bh = *reinterpret_cast<const bf16x8*>(tw + b_row + sl[0][0]);
                bl = *reinterpret_cast<const bf16x8*>(tw + b_row + sl[0][1]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks == 1 && krem <= 16) break;
                    bf16x8 ah[2], al[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) split_bf16(ar[a][0], ar[a][1], ah[a], al[a]);
                    if (ks == 0) {
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            ar[a][0] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (64 * BK) + sl[1][0]);
                            ar[a][1] = *reinterpret_cast<const f32x4*>(tb + a_row + a * (64 * BK) + sl[1][1]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bf16x8 ch_ = bh, cl_ = bl;
                        if (c == 0) {
                            bh = *reinterpret_cast<const bf16x8*>(tw + b_row + 64 * BK + sl[ks][0]);
                            bl = *reinterpret_cast<const bf16x8*>(tw + b_row + 64 * BK + sl[ks][1]);
                        } else if (ks == 0) {
                            bh = *reinterpret_cast<const bf16x8*>(tw + b_row + sl[1][0]);
                            bl = *reinterpret_cast<const bf16x8*>(tw + b_row + sl[1][1]);
                        }
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], ch_, acc[a][c], 0, 0, 0);
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], cl_, acc[a][c], 0, 0, 0);
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], ch_, acc[a][c], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- epilogue.  Barrier E (matched by the producers at every tile end): every consumer has left the A slot of
            //      the tile's last chunk, which nobody refills before barrier(g) -- it is the scratch of the per-wave 32x32
            //      transposes that turn the MFMA C layout (lane = column) into 16 bytes per lane along the channel axis.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float* scratch = smem + ((g - 1) & 3) * ST_TILE + wave * 1024;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#if LDN_ABLATE & 64
                if (acc[a][0][0] != 12345.678f) continue;
#endif
                const int mi = wm + 2 * a;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ccol = n0 + (wn + 2 * c) * 32 + tc4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[a][c][r];
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
                    f32x4 v[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const f32x4*>(scratch + (trow + 8 * it) * 32 + tc4);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int o = orw[a][it];
                        f32x4 x = v[it] * sc4[c] + sh4[c] + res[a][c][it];
                        if (o & ST_ROW_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
                        }
                        x -= ps4[c];
                        if (o >= 0) {
#if LDN_ABLATE & 16
                            if (x[0] == 12345.678f)
#endif
#if LDN_STREAM_NT & 2
                            __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p.out + (size_t)(o & (ST_ROW_RELU - 1)) * p.ldo + ccol));
#else
                            *reinterpret_cast<f32x4*>(p.out + (size_t)(o & (ST_ROW_RELU - 1)) * p.ldo + ccol) = x;
#endif
                            csum += x;
                        }
                    }
                    if (p.colsum) {   // fused global-average-pool partials of the 32-pixel subtile (dense image mode only)
#pragma unroll
                        for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);
                        const int sub = ((r0 + mb * ST_BM) >> 5) + mi;
                        if (trow == 0 && sub * 32 < HWo)
                            *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * ceil_div(HWo, 32) + sub) * p.cout + ccol) = csum;
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host
// Arithmetic of a launch: 0 = fp32 MFMA, 1 = bf16x3 split precision.  It is an ARGUMENT of every conv entry point (ImgArgs::math);
// LDN_MATH_DEFAULT (-1) resolves to the read-only process default taken once from the environment (LDN_MATH_MODE).  The library
// keeps no mutable state.
static int default_math_mode() {
    static const int v = [] { const char* e = getenv("LDN_MATH_MODE"); return (e && e[0] == '1') ? 1 : 0; }();
    return v;
}

static size_t tile_lds_bytes(const ImgArgs& p, int MS, int NS) {
    return (size_t)2 * (MS + NS) * 32 * BK * sizeof(float) + (size_t)((2 + p.shift_classes) * NS * 32) * sizeof(float) +
           (size_t)(2 * MS * 32 + NS * 32 + (p.k_idx ? p.cin : 0) + (p.packed ? MS * 32 * (1 + p.ksize) : 0)) * sizeof(int);
}

template <int MS, int NS, int WM, int BMODE, bool KSKIP>
static int launch_bf3(const ImgArgs& p, hipStream_t st) {
    constexpr int MINW = (MS + NS) <= 8 ? 4 : 2;
    const size_t lds = tile_lds_bytes(p, MS, NS);
    LDN_REQUIRE(lds <= 160 * 1024, "k_conv_bf3: %zu B of LDS exceed 160 KiB (cin too large for k_idx)", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_conv_bf3<MS, NS, WM, BMODE, KSKIP, MINW>), lds),
                "k_conv_bf3: cannot reserve %zu B of LDS", lds);
    ImgArgs q = p;
    const int msubs = ceil_div(p.packed ? p.m_cap : p.Ho * p.Wo, 32);
    const int mtn = ceil_div(msubs, MS);
    q.bm = ceil_div(msubs, mtn) * 32;
    q.mtn = mtn;
    const unsigned grid = (unsigned)round_up(p.B * mtn, 8) * p.ntn;   // units padded to the 8 XCDs (tile_setup)
    hipLaunchKernelGGL((k_conv_bf3<MS, NS, WM, BMODE, KSKIP, MINW>), dim3(grid), dim3(512), lds, st, q);
    LDN_CHECK_LAUNCH("k_conv_bf3");
    return LDN_OK;
}

template <int MS, int NS, int BMODE, bool KSKIP>
static int launch_k(const ImgArgs& p, hipStream_t st) {
    if (p.math == 1) {
        // consumer wave grid WM x 4/WM: every wave owns whole m-subtiles (A fragments are split once and reused);
        // 4 x 1 for the tall 8 x 6 tile, 2 x 2 otherwise; 7x7 images (<= 2 m-subtiles) get a 2-subtile-high tile
        const int msubs = ceil_div(p.packed ? p.m_cap : p.Ho * p.Wo, 32);
        if constexpr (MS == 4 && NS == 10) {
            if (msubs <= 2) return launch_bf3<2, 10, 2, BMODE, KSKIP>(p, st);
        }
        if constexpr (MS == 4 && NS == 4) {
            // 4x4 tiles: 4 x 1 wave grid (one A split per four tiles) unless the epilogue adds a residual, where the
            // 2 x 2 grid measured 5-8 % faster (conv3-type launches)
            if (!p.residual) return launch_bf3<4, 4, 4, BMODE, KSKIP>(p, st);
        }
        return launch_bf3<MS, NS, (MS >= 8 ? 4 : 2), BMODE, KSKIP>(p, st);
    }
    // blocks of <= 64 KiB LDS run two per CU (memory-bound early stages need the extra waves in flight)
    constexpr int MINW = (MS + NS) <= 8 ? 4 : 2;
    const size_t lds = tile_lds_bytes(p, MS, NS);
    LDN_REQUIRE(lds <= 160 * 1024, "k_conv_image: %zu B of LDS exceed 160 KiB (cin too large for k_idx)", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_conv_image<MS, NS, BMODE, KSKIP, MINW>), lds),
                "k_conv_image: cannot reserve %zu B of LDS", lds);
    ImgArgs q = p;
    const int msubs = ceil_div(p.packed ? p.m_cap : p.Ho * p.Wo, 32);
    const int mtn = ceil_div(msubs, MS);              // M blocks per image, then balance their sizes
    q.bm = ceil_div(msubs, mtn) * 32;
    q.mtn = mtn;
    const unsigned grid = (unsigned)round_up(p.B * mtn, 8) * p.ntn;   // units padded to the 8 XCDs (tile_setup)
    hipLaunchKernelGGL((k_conv_image<MS, NS, BMODE, KSKIP, MINW>), dim3(grid), dim3(512), lds, st, q);
    LDN_CHECK_LAUNCH("k_conv_image");
    return LDN_OK;
}

template <int BMODE>
static int launch_shape(const ImgArgs& a, hipStream_t st) {
    ImgArgs p = a;
    const int nsubs = ceil_div(a.cout, 32);
    const int hw = a.packed ? a.m_cap : a.Ho * a.Wo;
    // tile shape: as many output pixels as the LDS budget allows for the layer's width, so that every byte pulled
    // into the CU is reused by as many MFMAs as possible (whole 14x14 / 7x7 images at stages 3 / 4)
    int per;                                   // n-subtiles per N block
    if (nsubs <= 4) per = nsubs;               // <= 128 columns: memory-bound layers, two 64 KiB blocks per CU
    else if (hw <= 128) per = min(nsubs, 10);  // 7x7 images: wide N blocks (weights dominate the traffic)
    else if (hw <= 256 && a.n_idx) per = min(nsubs, 6);   // 14x14 images: whole image x <= 192 columns per block
    // (measured: the wide 1x1 convs without an output list -- conv3, downsample -- are faster as two co-resident
    //  64 KiB blocks per CU whose phases overlap than as one whole-image block: 208 vs 294 us at stage 3)
    else per = 4;
    p.bn = per * 32;
    p.ntn = ceil_div(a.cout, p.bn);
    // KSKIP (skip empty k groups of a partial chunk) only pays for the narrow layers, which use the small shapes
    if (per <= 2) return launch_k<6, 2, BMODE, true>(p, st);
    if (per <= 4) return launch_k<4, 4, BMODE, true>(p, st);
    if (per <= 6) return launch_k<8, 6, BMODE, false>(p, st);
    return launch_k<4, 10, BMODE, false>(p, st);
}

// rows per workgroup of k_conv1x1_stream (tuning: env LDN_STREAM_ROWS, multiple of 128, <= 1024; 0 disables the kernel)
static int stream_rows() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LDN_STREAM_ROWS");
        v = e ? atoi(e) : 512;
        if (v > ST_MAXROWS) v = ST_MAXROWS;
    }
    return v;
}

template <int BMODE>
static int launch_stream(const ImgArgs& p, hipStream_t st) {
    const size_t lds = (size_t)ST_LDS_FLOATS * sizeof(float) + (size_t)(2 * ST_MAXROWS + p.cin + 32) * sizeof(int);
    LDN_REQUIRE(lds <= 160 * 1024, "k_conv1x1_stream: %zu B of LDS exceed 160 KiB", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_conv1x1_stream<BMODE>), lds),
                "k_conv1x1_stream: cannot reserve %zu B of LDS", lds);
    ImgArgs q = p;
    const int rows = p.packed ? p.m_cap : p.Ho * p.Wo;
    int rpb = stream_rows();                           // rows per workgroup: fewer when that is needed to fill the chip
    while (rpb > ST_BM && p.B * ceil_div(rows, rpb) < 512) rpb /= 2;
    int nblk = ceil_div(rows, rpb);
    q.bm = round_up(ceil_div(rows, nblk), ST_BM);      // balanced, whole M blocks
    nblk = ceil_div(rows, q.bm);
    q.mtn = nblk;
    // too few row blocks to fill the chip (shared-weight launches over a few thousand packed rows): split the N tiles too
    const int ntiles = p.cout / ST_BN;
    int nsplit = 1;
    while (p.B * nblk * nsplit < 512 && nsplit < ntiles) nsplit *= 2;
    q.bn = ceil_div(ntiles, nsplit);
    q.ntn = ceil_div(ntiles, q.bn);
    hipLaunchKernelGGL((k_conv1x1_stream<BMODE>), dim3((unsigned)p.B * nblk * q.ntn), dim3(512), lds, st, q);
    LDN_CHECK_LAUNCH("k_conv1x1_stream");
    return LDN_OK;
}

static int dispatch_mode(const ImgArgs& p, int kgran, hipStream_t st) {
    const long taps = p.packed ? p.ksize : (long)p.ksize * p.ksize;
    LDN_REQUIRE(!p.k_idx || p.cin % 8 == 0, "conv: an input-channel list needs cin to be a multiple of 8 (got %d)", p.cin);
    LDN_REQUIRE(taps * p.cin * p.cout < (1L << 31), "conv: weight tensor of %ld elements exceeds the 32-bit offsets of the weight staging",
                taps * p.cin * p.cout);
    // wide 1x1 convolutions without an output-channel list: the persistent streaming kernel (bf16x3 arithmetic only)
    if (p.math == 1 && stream_rows() >= ST_BM && taps == 1 && !p.n_idx && p.shift_classes == 1 &&
        p.cout % ST_BN == 0 && p.cout >= 2 * ST_BN && !(p.residual && p.scale) && p.cin <= 1024 &&
        (p.packed ? p.m_cap : p.Ho * p.Wo) >= 96)   // (7x7 images would leave most of every 128-row M block empty)
        return p.k_idx ? launch_stream<B_KN4>(p, st) : launch_stream<B_NK>(p, st);
    if (!p.k_idx) return launch_shape<B_NK>(p, st);                  // w is [cout][taps][cin]
    const int g = p.n_idx ? kgran : 4;                               // w is [taps][cin][cout]
    if (g % 4 == 0) return launch_shape<B_KN4>(p, st);
    if (g % 2 == 0) return launch_shape<B_KN2>(p, st);
    return launch_shape<B_KN1>(p, st);
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_conv)

}  // namespace ldn

using namespace ldn;

extern "C" int ldn_default_math_mode(void) { return default_math_mode(); }

static int resolve_math(int math_mode, int* out) {
    LDN_REQUIRE(math_mode >= -1 && math_mode <= 1, "math_mode must be LDN_MATH_DEFAULT (-1), LDN_MATH_FP32 (0) or LDN_MATH_BF16X3 (1), got %d", math_mode);
    *out = math_mode < 0 ? default_math_mode() : math_mode;
    return LDN_OK;
}

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_trace(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int ldn_conv_image(const float* a, int lda, int B, int Hi, int Wi, int ksize, int stride, int Ho, int Wo,
                              const float* w, int cin, int cout, const int32_t* k_idx, const int32_t* k_cnt,
                              int kgran, const int32_t* n_idx, const int32_t* n_cnt, const float* scale,
                              const float* shift, int shift_classes, const float* post_sub, int relu,
                              const float* residual, int ldr, float* out, int ldo, float* colsum, int out_format,
                              int math_mode, void* stream) {
    LDN_REQUIRE(a && w && shift && out, "ldn_conv_image: null pointer");
    int math;
    if (int rc = resolve_math(math_mode, &math)) return rc;
    LDN_REQUIRE(!colsum || (!n_idx && (uintptr_t)colsum % 16 == 0), "ldn_conv_image: colsum needs a dense output and 16-byte alignment");
    LDN_REQUIRE(ksize == 1 || ksize == 3, "ldn_conv_image: ksize must be 1 or 3 (got %d)", ksize);
    LDN_REQUIRE(stride >= 1 && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "ldn_conv_image: bad geometry");
    LDN_REQUIRE((Ho - 1) * stride < Hi && (Wo - 1) * stride < Wi, "ldn_conv_image: output grid exceeds input");
    LDN_REQUIRE(Ho < 65536 && Wo < 65536, "ldn_conv_image: spatial size too large");
    LDN_REQUIRE(cin > 0 && cout > 0 && cin % 4 == 0, "ldn_conv_image: cin must be a positive multiple of 4 (got %d)", cin);
    LDN_REQUIRE(lda % 4 == 0, "ldn_conv_image: lda must be a multiple of 4");
    LDN_REQUIRE((k_idx == nullptr) == (k_cnt == nullptr) && (n_idx == nullptr) == (n_cnt == nullptr),
                "ldn_conv_image: index list and count must be given together");
    LDN_REQUIRE(shift_classes == 1 || shift_classes == 16, "ldn_conv_image: shift_classes must be 1 or 16");
    LDN_REQUIRE(!k_idx || kgran >= 1, "ldn_conv_image: kgran must be >= 1");
    LDN_REQUIRE(ldo >= cout && ldo % 4 == 0 && cout % 4 == 0, "ldn_conv_image: cout and ldo must be multiples of 4, ldo >= cout");
    LDN_REQUIRE(!(n_idx && residual), "ldn_conv_image: residual with an output-channel subset is not supported");
    LDN_REQUIRE((uintptr_t)out % 16 == 0 && (!residual || ((uintptr_t)residual % 16 == 0 && ldr % 4 == 0)),
                "ldn_conv_image: out/residual must be 16-byte aligned with strides that are multiples of 4");
    LDN_REQUIRE(!residual || ldr >= cout, "ldn_conv_image: ldr < cout");
    LDN_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), "ldn_conv_image: a/w must be 16-byte aligned");
    ImgArgs p{a, lda, B, Hi, Wi, ksize, stride, Ho, Wo, w, cin, cout, k_idx, k_cnt, n_idx, n_cnt,
              scale, shift, shift_classes, post_sub, relu, residual, ldr, out, ldo, colsum,
              0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, math, out_format};
    LDN_REQUIRE(out_format == 0 || (out_format == 1 && math == 1 && !residual && !colsum && ldo % 32 == 0 && ldo >= round_up(cout, 32)),
                "ldn_conv_image: out_format 1 (pre-split bf16 pairs) needs math_mode bf16x3, no residual / colsum and ldo %% 32 == 0");
    return dispatch_mode(p, kgran, static_cast<hipStream_t>(stream));
}

extern "C" int ldn_conv_packed(const float* a, int lda, int B, const int32_t* row_prefix, const int32_t* m_count,
                               int m_cap, const int32_t* a_map, int taps, const int32_t* out_map,
                               const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride, const float* w,
                               int cin, int cout, const int32_t* k_idx, const int32_t* k_cnt, int kgran,
                               const int32_t* n_idx, const int32_t* n_cnt, const float* scale, const float* shift,
                               int shift_classes, const float* post_sub, int relu, const int32_t* relu_if_neg,
                               const float* residual, int ldr, float* out, int ldo, int math_mode, void* stream) {
    LDN_REQUIRE(a && w && shift && out, "ldn_conv_packed: null pointer");
    int math;
    if (int rc = resolve_math(math_mode, &math)) return rc;
    LDN_REQUIRE(taps == 1 || taps == 9, "ldn_conv_packed: taps must be 1 or 9 (got %d)", taps);
    LDN_REQUIRE(a_map || taps == 1, "ldn_conv_packed: a_map required when taps > 1");
    LDN_REQUIRE(B >= 1 && (row_prefix || B == 1), "ldn_conv_packed: B > 1 needs row_prefix");
    LDN_REQUIRE(cin > 0 && cout > 0 && cin % 4 == 0, "ldn_conv_packed: cin must be a positive multiple of 4 (got %d)", cin);
    LDN_REQUIRE(lda % 4 == 0 && lda >= (k_idx ? 4 : cin), "ldn_conv_packed: lda must be a multiple of 4 and >= cin");
    LDN_REQUIRE((k_idx == nullptr) == (k_cnt == nullptr) && (n_idx == nullptr) == (n_cnt == nullptr),
                "ldn_conv_packed: index list and count must be given together");
    LDN_REQUIRE((!k_idx && !n_idx) || row_prefix, "ldn_conv_packed: per-image channel lists need per-image row ranges");
    LDN_REQUIRE(shift_classes == 1 || (shift_classes == 16 && pix_map && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0 && stride >= 1),
                "ldn_conv_packed: shift_classes 16 needs pix_map and the layer geometry");
    LDN_REQUIRE(relu >= 0 && relu <= 2 && (relu != 2 || relu_if_neg), "ldn_conv_packed: bad relu mode");
    LDN_REQUIRE(ldo >= cout && ldo % 4 == 0 && cout % 4 == 0, "ldn_conv_packed: cout and ldo must be multiples of 4, ldo >= cout");
    LDN_REQUIRE(!(n_idx && residual), "ldn_conv_packed: residual with an output-channel subset is not supported");
    LDN_REQUIRE((uintptr_t)out % 16 == 0 && (!residual || ((uintptr_t)residual % 16 == 0 && ldr % 4 == 0 && ldr >= cout)),
                "ldn_conv_packed: out/residual must be 16-byte aligned with strides that are multiples of 4");
    LDN_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), "ldn_conv_packed: a/w must be 16-byte aligned");
    if (m_cap <= 0) return LDN_OK;
    ImgArgs p{a, lda, B, Hi > 0 ? Hi : 1, Wi > 0 ? Wi : 1, taps, stride >= 1 ? stride : 1, Ho > 0 ? Ho : 1, Wo > 0 ? Wo : 1,
              w, cin, cout, k_idx, k_cnt, n_idx, n_cnt, scale, shift, shift_classes, post_sub, relu, residual, ldr, out, ldo,
              nullptr, 1, row_prefix, m_count, m_cap, a_map, out_map, pix_map, relu_if_neg, 0, 0, 0, 0, math, 0};
    return dispatch_mode(p, kgran, static_cast<hipStream_t>(stream));
}

// the shared-weight packed-row convolution of the spatial / layer path: one "image" holding every active row
extern "C" int ldn_conv_rows(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count,
                             int m_cap, const float* w, int cin, int cout, const float* scale, const float* shift,
                             int relu, const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual,
                             int ldr, float* out, int ldo, int math_mode, void* stream) {
    return ldn_conv_packed(a, lda, 1, nullptr, m_count, m_cap, a_rows, taps, out_rows, nullptr, 0, 0, 0, 0, 1, w, cin, cout,
                           nullptr, nullptr, 1, nullptr, nullptr, scale, shift, 1, nullptr, relu, relu_if_neg, residual, ldr,
                           out, ldo, math_mode, stream);
}
