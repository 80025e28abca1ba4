// k_conv_image -- channel mode: one ragged fp32-MFMA GEMM per image (gfx950 / CDNA4).
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
#define LDN_ABLATE 0   // tuning only: 1 = no MFMA, 2 = no loads in the K loop (results are wrong)
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
    int bn;     // columns per N block (multiple of 32, <= NS*32)
    int bm;     // pixels per M block (multiple of 32, <= MS*32; balanced over the image)
};

constexpr int BK = 32;   // K chunk = one 128-byte LDS row

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

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
    // image-fastest block order: with B % 8 == 0 every block of image b runs on XCD b % 8, whose L2 then holds
    // that image's activations; the (shared) weights are resident in every XCD's L2.
    const int b = bid % p.B;
    const int t = bid / p.B;
    const int nt = t % p.ntn, mt = t / p.ntn;
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
    const int Kb = p.k_idx ? p.k_cnt[b] : p.cin;
    const int Nb = p.n_idx ? p.n_cnt[b] : p.cout;
    const int Nb4 = min(round_up(Nb, 4), p.cout);
    if (n0 >= Nb4 || m0 >= HWo) return;
    const int T = p.packed ? p.ksize : p.ksize * p.ksize;   // packed mode: ksize carries the tap count (1 or 9)
    const int pad = p.packed ? (T == 9 ? 1 : 0) : p.ksize >> 1;
    const int msub = ceil_div(min(HWo - m0, p.bm), 32);        // valid m-subtiles (1..MS)
    const int nsub = ceil_div(min(Nb4 - n0, p.bn), 32);        // valid n-subtiles (1..NS)
    const int ntiles = msub * nsub;
#ifdef LDN_TRACE
    unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime(), tr_t0 = __builtin_amdgcn_s_memtime(), tr_bar = 0, tr_mma = 0, tr_iss = 0, tr_a = 0, tr_b = 0;
#endif

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
                if (p.relu == 2 && p.relu_if_neg[rbase + m] < 0) orow |= 0x40000000;   // bit 30: apply ReLU to this row
            }
            s_orow[i] = orow;
        }
    }
    if (p.packed)
        for (int i = tid; i < BM * T; i += 512) {
            const int r = i / T, m = m0 + r;
            s_arow[i] = (m < HWo && r < p.bm) ? (p.a_map ? p.a_map[(size_t)(rbase + m) * T + (i - r * T)] : rbase + m) : -1;
        }
    for (int i = tid; i < BNX; i += 512) {
        const int j = n0 + i;
        const bool in_block = i < p.bn;
        const int chn = (in_block && j < Nb) ? (p.n_idx ? p.n_idx[(size_t)b * p.cout + j] : j)
                                             : ((in_block && j < Nb4) ? -1 : -2);
        s_nch[i] = chn;
        s_sc[i] = chn >= 0 ? p.scale[chn] : 0.f;
        s_ps[i] = (chn >= 0 && p.post_sub) ? p.post_sub[chn] : 0.f;
        for (int c = 0; c < p.shift_classes; ++c) s_sh[c * BNX + i] = chn >= 0 ? p.shift[c * p.cout + chn] : 0.f;
    }
    if (p.k_idx)
        for (int i = tid; i < Kb; i += 512) s_kidx[i] = p.k_idx[(size_t)b * p.cin + i];
    __syncthreads();

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
            unsigned long long* r = g_trace + ((size_t)blockIdx.x + gridDim.x) * 6;
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
    const int trow = lane >> 3, tc4 = (lane & 7) * 4;
#pragma unroll
    for (int s = 0; s < ACC; ++s) {
        const int tt = wave + 4 * s;
        if (tt >= ntiles) continue;
        const int nj = tt / msub, mi = tt - nj * msub;
        const int col = nj * 32 + l31;
        const int chn = s_nch[col];
        const float sc = s_sc[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = chn >= 0 ? acc[s][r] * sc + s_sh[s_cls[mi * 32 + row] + col] : 0.f;
            scratch[row * 32 + l31] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ccol = nj * 32 + tc4;
        if (s_nch[ccol] != -2) {
            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_ps + ccol);
            f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = mi * 32 + trow + 8 * it;
                if (s_pix[row] < 0) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (trow + 8 * it) * 32 + tc4);
                size_t orow = (size_t)rbase + m0 + row;
                bool do_relu = p.relu == 1;
                if (p.packed) {
                    const int o = s_orow[row];
                    do_relu = do_relu || (o & 0x40000000);
                    orow = (size_t)(o & 0x3fffffff);
                }
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + orow * p.ldr + n0 + ccol);
                if (do_relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                v -= ps;
                *reinterpret_cast<f32x4*>(p.out + orow * p.ldo + n0 + ccol) = v;
                csum += v;
            }
            if (p.colsum) {   // fused global-average-pool partials for the NEXT block's channel masker
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = csum[e];
                    t += __shfl_xor(t, 8, 64);
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    csum[e] = t;
                }
                if (trow == 0) {
                    const size_t slot = ((size_t)b * ceil_div(HWo, 32) + (m0 >> 5) + mi) * p.cout + n0 + ccol;   // dense mode only
                    *reinterpret_cast<f32x4*>(p.colsum + slot) = csum;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef LDN_TRACE
    if (tid == 0 && g_trace) {
        unsigned long long* r = g_trace + (size_t)blockIdx.x * 6;
        r[0] = tr_t0; r[1] = __builtin_amdgcn_s_memtime();
        r[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        r[3] = __builtin_amdgcn_s_memrealtime() - tr_r0;                        // 100 MHz ticks
        r[4] = ntiles | (tr_bar << 32); r[5] = (tr_mma << 32) | (tr_iss & 0xffffffffull);
    }
#endif
}

// ---------------------------------------------------------------------------------------------- host
template <int MS, int NS, int BMODE, bool KSKIP>
static int launch_k(const ImgArgs& p, hipStream_t st) {
    // blocks of <= 64 KiB LDS run two per CU (memory-bound early stages need the extra waves in flight)
    constexpr int MINW = (MS + NS) <= 8 ? 4 : 2;
    const size_t lds = (size_t)2 * (MS + NS) * 32 * BK * sizeof(float) + (size_t)((2 + p.shift_classes) * NS * 32) * sizeof(float) +
                       (size_t)(2 * MS * 32 + NS * 32 + (p.k_idx ? p.cin : 0) + (p.packed ? MS * 32 * (1 + p.ksize) : 0)) * sizeof(int);
    LDN_REQUIRE(lds <= 160 * 1024, "k_conv_image: %zu B of LDS exceed 160 KiB (cin too large for k_idx)", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_conv_image<MS, NS, BMODE, KSKIP, MINW>), lds),
                "k_conv_image: cannot reserve %zu B of LDS", lds);
    ImgArgs q = p;
    const int msubs = ceil_div(p.packed ? p.m_cap : p.Ho * p.Wo, 32);
    const int mtn = ceil_div(msubs, MS);              // M blocks per image, then balance their sizes
    q.bm = ceil_div(msubs, mtn) * 32;
    const unsigned grid = (unsigned)p.B * mtn * p.ntn;
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
    else per = 4;
    p.bn = per * 32;
    p.ntn = ceil_div(a.cout, p.bn);
    // KSKIP (skip empty k groups of a partial chunk) only pays for the narrow layers, which use the small shapes
    if (per <= 2) return launch_k<6, 2, BMODE, true>(p, st);
    if (per <= 4) return launch_k<4, 4, BMODE, true>(p, st);
    if (per <= 6) return launch_k<8, 6, BMODE, false>(p, st);
    return launch_k<4, 10, BMODE, false>(p, st);
}

static int dispatch_mode(const ImgArgs& p, int kgran, hipStream_t st) {
    if (!p.k_idx) return launch_shape<B_NK>(p, st);                  // w is [cout][taps][cin]
    const int g = p.n_idx ? kgran : 4;                               // w is [taps][cin][cout]
    if (g % 4 == 0) return launch_shape<B_KN4>(p, st);
    if (g % 2 == 0) return launch_shape<B_KN2>(p, st);
    return launch_shape<B_KN1>(p, st);
}

}  // namespace ldn

using namespace ldn;

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
                              const float* residual, int ldr, float* out, int ldo, float* colsum, void* stream) {
    LDN_REQUIRE(a && w && scale && shift && out, "ldn_conv_image: null pointer");
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
              0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    return dispatch_mode(p, kgran, static_cast<hipStream_t>(stream));
}

extern "C" int ldn_conv_packed(const float* a, int lda, int B, const int32_t* row_prefix, const int32_t* m_count,
                               int m_cap, const int32_t* a_map, int taps, const int32_t* out_map,
                               const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride, const float* w,
                               int cin, int cout, const int32_t* k_idx, const int32_t* k_cnt, int kgran,
                               const int32_t* n_idx, const int32_t* n_cnt, const float* scale, const float* shift,
                               int shift_classes, const float* post_sub, int relu, const int32_t* relu_if_neg,
                               const float* residual, int ldr, float* out, int ldo, void* stream) {
    LDN_REQUIRE(a && w && scale && shift && out, "ldn_conv_packed: null pointer");
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
              nullptr, 1, row_prefix, m_count, m_cap, a_map, out_map, pix_map, relu_if_neg, 0, 0, 0};
    return dispatch_mode(p, kgran, static_cast<hipStream_t>(stream));
}

// the shared-weight packed-row convolution of the spatial / layer path: one "image" holding every active row
extern "C" int ldn_conv_rows(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count,
                             int m_cap, const float* w, int cin, int cout, const float* scale, const float* shift,
                             int relu, const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual,
                             int ldr, float* out, int ldo, void* stream) {
    return ldn_conv_packed(a, lda, 1, nullptr, m_count, m_cap, a_rows, taps, out_rows, nullptr, 0, 0, 0, 0, 1, w, cin, cout,
                           nullptr, nullptr, 1, nullptr, nullptr, scale, shift, 1, nullptr, relu, relu_if_neg, residual, ldr,
                           out, ldo, stream);
}
