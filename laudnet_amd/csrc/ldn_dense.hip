// k_dense -- shared-weight 1x1 convolution over (packed) pixel rows in the round-2 style (gfx950, bf16x3):
//     out[dst(m), n] = act(scale[n] * sum_k A[src(m), k] * w[n, k] + shift[n] (+ residual[dst(m), n]))
// (laud_resnet.py:115-144 restricted to the active pixels: conv1 / conv3 / projection shortcut of spatial, layer and `both`
// blocks; the dense execution of channel-mode stage 4; LAD-RegNet a / c / proj).  Same contract as ldn_conv_rows with taps == 1.
//
// Structure (shared with k_head / k_tail, csrc/ldn_tail.hip): transposed MFMA formulation (A operand = weights, B operand =
// activations, lane = row); the weights are pre-split once per module into n-major rows [n][octet][8 hi | 8 lo] bf16 -- a
// weight fragment is two ds_read_b128 and no VALU; the activation rows land as raw fp32 through LDS-DMA (row gather = per-lane
// source address) and are split by the one wave that owns the 32 rows; no producer waves -- every wave issues its share of the
// DMA (inline asm) with counted vmcnt over a ring of 2-3 slots.  One workgroup = 256 rows x NT columns (NT = 64 / 128 / 256).
#include "ldn_common.h"
#include <type_traits>

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DenseArgs {
    const float* a; int lda;
    const int32_t* a_rows; const int32_t* m_count; int m_cap;
    const unsigned char* ws;                          // [cout][cin / 8][32 B]
    int cin, cout;
    const float* scale; const float* shift; int relu;
    const int32_t* relu_if_neg; const int32_t* out_rows;
    const float* residual; int ldr; float* out; int ldo;
    int taps;                                         // 1, or 9: a_rows is then [rows][9] (-1 = zero row) and w_split rows hold 9 * cin inputs (tap-major)
    int shift_classes; const int32_t* pix_map;        // 16: shift is [16][cout], selected by the border class of the row's output pixel
    int Hi, Wi, Ho, Wo, stride;                       //     (pix_map[row] = flat output pixel; geometry of the 3x3 layer)
    const float* post_sub;                            // optional [cout]: subtracted after the ReLU (channel algebra)
    const float* chmask; int rows_per_img;            // optional [B][cout] {0,1}: out *= chmask[dst / rows_per_img][n] (dense channel exec)
    int mtn, ntn;                                     // M tiles (of 256 rows), N tiles (of NT columns)
    const float* ln_stats; const float* ln_c1;        // optional LayerNorm of the A rows as an epilogue term: [rows_a][2] = {mean, rstd} of
                                                      // every SOURCE row, and c1[n] = sum_k w[n][k] (w carries the LayerNorm weight)
    float* pool; int pool_gy, pool_gx, pool_Sx;       // optional [B][S * Sx][cout]: the mean of the FINAL output over every patch of gy x gx pixels
                                                      // this launch writes (gy * gx = 4 or 16 consecutive packed rows = one patch: patch-major
                                                      // lists of k_plan) -- the next spatial masker's pooled means (models/utils.py:48-52)
    const float* a_gate; int gate_rows;               // optional (k_dense2<.., AG>) [slots][cin]: A row r is multiplied by a_gate[r / gate_rows][:] before the
                                                      // product -- the excitation of an SE block applied where conv c reads h_b (laud_regnet.py:196-197)
};

__device__ __attribute__((aligned(16))) float g_dense_zero[2048] = {0.f};     // a whole zero ROW for k_dense2 (cin <= 2048); k_dense reads its first 16 bytes
#ifdef LDN_TRACE   // tuning only: per-workgroup cycle split of the K loop (tools/trace_dense.py)
__device__ unsigned long long* g_dense_trace = nullptr;
#define DT(x) x = __builtin_amdgcn_s_memtime();
#else
#define DT(x)
#endif

#ifndef LDN_DENSE_ABLATE
#define LDN_DENSE_ABLATE 0   // tuning only (results are wrong): 1 = no LDS-DMA at all, 2 = every DMA reads the zero line, 4 = no MFMA, 8 = no weight-fragment reads
#endif
__device__ __forceinline__ void d_dma16(const void* gsrc, unsigned lds_base) {
#if LDN_DENSE_ABLATE & 1
    return;
#endif
#if LDN_DENSE_ABLATE & 2
    gsrc = g_dense_zero;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void d_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void d_wait_vm_rt(int n) {
    switch (n) {
        case 1: d_wait_vm<1>(); break;   case 2: d_wait_vm<2>(); break;   case 3: d_wait_vm<3>(); break;
        case 4: d_wait_vm<4>(); break;   case 5: d_wait_vm<5>(); break;   case 6: d_wait_vm<6>(); break;
        case 7: d_wait_vm<7>(); break;   case 8: d_wait_vm<8>(); break;   case 9: d_wait_vm<9>(); break;
        case 10: d_wait_vm<10>(); break; case 11: d_wait_vm<11>(); break; case 12: d_wait_vm<12>(); break;
        case 13: d_wait_vm<13>(); break; case 14: d_wait_vm<14>(); break; case 15: d_wait_vm<15>(); break;
        case 16: d_wait_vm<16>(); break;
        default: d_wait_vm<0>(); break;   // 0, and anything unexpected: wait for everything (always safe)
    }
}
__device__ __forceinline__ void d_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned d_lds_off(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}

constexpr int D_ROWS_MAX = 256;
constexpr int D_ROW_RELU = 1 << 30;

// The epilogue of the row kernels (k_dense, k_dense2): per n-subtile, C layout (lane = row, register = channel) -> rows of 32 channels through
// a per-wave 32 x 32 transpose scratch `scr` (4 KB of LDS owned by the wave), then the affine / residual / activation / mask terms and 16-byte stores.
// FEAT: the rarely used terms are compiled in -- the exact (erf) GELU of relu == 3, the LayerNorm fold (ln_stats), the per-image channel mask
// and post_sub of the dense channel execution.  Without them the unrolled epilogue is half as long (k_dense2's default instantiations).
template <int NSUB, bool T9, bool FULL, bool OF, bool FEAT = true>
__device__ __forceinline__ void dense_epilogue(const DenseArgs& p, f32x16 (&acc)[NSUB], const int* s_arow, const int* s_orow, const int* s_cls,
                                               float* scr, int wave, int lane, int n0, int nsub) {
    const int l31 = lane & 31, h = lane >> 5;
    // ---- epilogue: per n-subtile, C layout (lane = row, register = channel) -> rows of 32 channels, 16-byte accesses
    const int trw = lane >> 3, tc = lane & 7;
    int orw[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) orw[it] = s_orow[wave * 32 + trw + 8 * it];
    const bool gelu = FEAT && p.relu == 3;
    // pooled patch means (1x1, whole column tiles): which patch row of `pool` this lane stores, -1 = none.  16-pixel patches: the wave's
    // rows 0-15 / 16-31; the lanes with row-in-octet 0 of each half store.  4-pixel patches: rows 4q .. 4q+3, every lane stores one.
    long poff = -1;
    const int pb3 = (lane >> 3) & 1, pb4 = (lane >> 4) & 1;
    if constexpr (!T9 && FULL) {
        if (p.pool) {
            const int it_sel = p.pool_gy * p.pool_gx == 16 ? (lane >= 32 ? 2 : 0) : pb3 + 2 * pb4;
            const int o = it_sel == 0 ? orw[0] : it_sel == 1 ? orw[1] : it_sel == 2 ? orw[2] : orw[3];
            if (o >= 0 && (p.pool_gy * p.pool_gx == 4 || (lane & 0x18) == 0)) {
                const int f = o & (D_ROW_RELU - 1), hw = p.Ho * p.Wo;
                const int b = f / hw, pix = f - b * hw, y = pix / p.Wo, x = pix - y * p.Wo;
                poff = ((long)b * (p.Ho / p.pool_gy) * p.pool_Sx + (y / p.pool_gy) * p.pool_Sx + x / p.pool_gx) * p.cout;
            }
        }
    }
    auto load_res = [&](int j, f32x4 (&res)[4], f32x4& sc, f32x4& sh, f32x4& ps) {
        const int cb = n0 + 32 * j + tc * 4;
        const bool cok = FULL || cb < p.cout;                    // (a ragged last subtile: columns beyond cout are neither read nor stored)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float* src = (p.residual && orw[it] >= 0 && cok) ? p.residual + (size_t)(orw[it] & (D_ROW_RELU - 1)) * p.ldr + cb : g_dense_zero;
            res[it] = *reinterpret_cast<const f32x4*>(src);
        }
        sh = cok ? *reinterpret_cast<const f32x4*>(p.shift + cb) : f32x4{0.f, 0.f, 0.f, 0.f};   // (one class; the 16-class table is read per row below)
        sc = (p.scale && cok) ? *reinterpret_cast<const f32x4*>(p.scale + cb) : f32x4{1.f, 1.f, 1.f, 1.f};
        ps = (FEAT && p.post_sub && cok) ? *reinterpret_cast<const f32x4*>(p.post_sub + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        if (j >= nsub) continue;
        f32x4 res[4], sc, sh, ps;
        load_res(j, res, sc, sh, ps);
        f32x4 cm[4];
        const bool cok = FULL || n0 + 32 * j + tc * 4 < p.cout;
        // LayerNorm of the activation rows, applied AFTER the GEMM: LN(x) . w = rstd (x . w' - mean sum_k w'[k]) + const, w' = gamma * w
        float2 lst[4];
        f32x4 lc1 = {0.f, 0.f, 0.f, 0.f};
        if (FEAT && p.ln_stats) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int ar = s_arow[wave * 32 + trw + 8 * it];
                lst[it] = ar >= 0 ? *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)ar) : float2{0.f, 1.f};
            }
            if (cok) lc1 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n0 + 32 * j + tc * 4);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float* src = (FEAT && p.chmask && orw[it] >= 0 && cok) ? p.chmask + (size_t)((orw[it] & (D_ROW_RELU - 1)) / p.rows_per_img) * p.cout + n0 + 32 * j + tc * 4
                                                          : nullptr;
            cm[it] = src ? *reinterpret_cast<const f32x4*>(src) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = {acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
            *reinterpret_cast<f32x4*>(scr + l31 * 32 + (((2 * q4 + h) ^ (l31 & 7)) << 2)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // residual / scale / shift registers touched before the first store (gfx9: a load first used after a store waits vmcnt(0)
        // for that store's acknowledgement)
        asm volatile("" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(sc), "+v"(sh), "+v"(ps));
        asm volatile("" : "+v"(cm[0]), "+v"(cm[1]), "+v"(cm[2]), "+v"(cm[3]));
        f32x4 xs4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = trw + 8 * it;
            f32x4 x = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tc ^ (row & 7)) << 2));
            const f32x4 shr = (T9 && p.shift_classes > 1 && cok) ? *reinterpret_cast<const f32x4*>(p.shift + s_cls[wave * 32 + row] + n0 + 32 * j + tc * 4) : sh;
            if (FEAT && p.ln_stats) x = (x - lc1 * lst[it].x) * lst[it].y;
            x = x * sc + shr + res[it];
            if (orw[it] & D_ROW_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
            }
            if (gelu) {   // exact (erf) GELU of the token-skip MLP: the hidden activations never make a second trip through HBM
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = 0.5f * x[e] * (1.f + erff(x[e] * 0.70710678118654752f));
            }
            x = (x - ps) * cm[it];
            if constexpr (OF) {
                // pre-split rows: lanes tc = 2 q and 2 q + 1 hold channels 0-3 and 4-7 of octet 4 j + q (+ n0 / 8); they exchange their
                // quads (DPP quad_perm [1,0,3,2]); the even lane stores the octet's 8 hi, the odd lane its 8 lo (16 bytes each, adjacent)
                const bool odd = tc & 1;
                const u32x4_t o = presplit_store_quad(x, odd);
                if (orw[it] >= 0 && cok)
                    store16(reinterpret_cast<unsigned char*>(p.out) + ((size_t)(orw[it] & (D_ROW_RELU - 1)) * p.ldo + n0 + 32 * j + (tc & ~1) * 4) * 4 + (odd ? 16 : 0), o);
            } else {
                if (orw[it] >= 0 && cok)
                    store16(p.out + (size_t)(orw[it] & (D_ROW_RELU - 1)) * p.ldo + n0 + 32 * j + tc * 4, x);
            }
            xs4[it] = x;
        }
        if constexpr (!T9 && FULL) {
            if (p.pool) {   // (wave-uniform) fixed-order sums over the rows of each patch: a register tree across the lanes that hold the patch
                // cross-lane steps on the VALU (gfx950: v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows of
                // two registers, DPP row_ror:8 reaches lane ^ 8) -- no trip through the LDS crossbar; swap(a, b) leaves {a.lo, b.lo} and
                // {a.hi, b.hi}: their sum is a's pair sum in the lower lanes and b's in the upper ones
                // (inline asm: hipcc 7.2 folds the builtin's two results into one register when they are only added -- `v_add v0, v0, v0`
                // behind the swap; the s_nop covers the VALU-write -> permlane-read wait states the compiler would insert)
                auto swap32_sum = [](float a, float b) {
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                    return a + b;
                };
                auto swap16_sum = [](float a, float b) {
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                    return a + b;
                };
                auto ror8 = [](float a) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xf, 0xf, false)); };
                f32x4 m;
                if (p.pool_gy * p.pool_gx == 16) {
                    const f32x4 sa = xs4[0] + xs4[1], sb = xs4[2] + xs4[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = swap32_sum(sa[e], sb[e]);      // lanes 0-31: rows 0-15 (sa), lanes 32-63: rows 16-31 (sb)
                        v = swap16_sum(v, v);                    // (by value: two registers -- the instruction exchanges rows BETWEEN its operands)
                        m[e] = v + ror8(v);
                    }
                    m *= 0.0625f;
                } else {
                    const f32x4 k0 = pb3 ? xs4[1] : xs4[0], g0 = pb3 ? xs4[0] : xs4[1];
                    const f32x4 k2 = pb3 ? xs4[3] : xs4[2], g2 = pb3 ? xs4[2] : xs4[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float k01 = k0[e] + ror8(g0[e]), k23 = k2[e] + ror8(g2[e]);
                        m[e] = swap16_sum(k01, k23);             // even 16-lane rows: k01's pair sum, odd rows: k23's
                    }
                    m *= 0.25f;
                }
                if (poff >= 0) *reinterpret_cast<f32x4*>(p.pool + poff + n0 + 32 * j + tc * 4) = m;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// NSUB = n-subtiles of 32 columns per workgroup (2, 4, 5, 6, 8); D = ring depth (2 for NSUB >= 6, 3 otherwise)
// T9: 3x3 over a neighbour table (compiled apart: the 1x1 form carries none of its tables or branches); FULL: cout is a multiple of
// the tile width, so every workgroup owns NSUB whole n-subtiles (compiled apart: no per-subtile branch in the K loop)
// F32: true-fp32 arithmetic (v_mfma_f32_32x32x2_f32, fp32 multiply and accumulate -- the `fp32` math mode of the library).  The byte
// layouts coincide with the bf16x3 form: an element takes 4 bytes either way (bf16 hi + bf16 lo, or one float), so ws is then the plain
// row-major fp32 weight matrix [cout][taps * cin] ([n][octet][8 floats]), the staging pipeline is the same, and a K16 step is eight
// 32x32x2 instructions (instruction i pairs k-slot i of lane half 0 with k-slot i of lane half 1) instead of three 32x32x16 ones.
// R: rows per workgroup -- 256 (eight waves), or 128 (four waves, ring depth 2: two workgroups per CU fit) for the launches whose grid
// of 256-row tiles would leave CUs with one workgroup or none (the stage-3 3x3 of the spatial workload, stage 4)
// PS (round 5): the activation rows arrive PRE-SPLIT, [row][cin / 8][8 hi | 8 lo] bf16 (the layout the weights have; 4 bytes per element, so
// every address computation is the fp32 one): a B fragment is two ds_read_b128 and NO VALU -- the K loop splits nothing.
// OF (round 5): the epilogue STORES pre-split rows (the producer side of PS: conv1 of a spatial / layer block writes h1 for the packed
// 3x3, k_rows3, which writes h2 for conv3): two neighbouring lanes pair their four channels into one octet, one stores 8 hi, the other 8 lo.
template <int NSUB, bool T9, bool FULL, bool F32 = false, int R = 256, bool PS = false, bool OF = false>
__global__ __launch_bounds__(2 * R, 2) void k_dense(const DenseArgs p) {
    constexpr int NT = NSUB * 32;
    constexpr int D = (NSUB >= 6 || R == 128) ? 2 : 3;
    constexpr int D_ROWS = R, NTHR = 2 * R, WR = R / 4;     // WR: weight rows one DMA round of all waves covers (8 per wave)
    constexpr int SLOT = (D_ROWS + NT) * 128;
    constexpr int NWI = (NT + WR - 1) / WR;               // weight DMA instructions per wave and chunk (8 rows each) ...
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* const s_arow = reinterpret_cast<int*>(smem);            // [256] source row or -1
    int* const s_orow = s_arow + D_ROWS;                         // [256] destination row | D_ROW_RELU, or -1
    int* const s_cls = s_orow + D_ROWS;                          // [256] border class * cout (16-class shift table) (T9 only)
    int* const s_atap = s_cls + D_ROWS;                          // [256][9] source rows per tap (T9 only)
    unsigned char* const s_ring = smem + (T9 ? 12 : 2) * D_ROWS * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware order: the N tiles of one M tile get consecutive slots of ONE XCD (block b runs on XCD b % 8), so the A rows of
    // the tile come from HBM once and from that XCD's L2 for the other N tiles
    const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
    const int nt = slot_i % p.ntn, mt = (slot_i / p.ntn) * 8 + xcd;
    if (mt >= p.mtn) return;
    const int M = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap;
    const int m0 = mt * D_ROWS;
    if (m0 >= M) return;
    const int rows = min(D_ROWS, M - m0);
    const int n0 = nt * NT;
    const int nsub = FULL ? NSUB : min(NSUB, ceil_div(p.cout - n0, 32));   // (the last one may hold fewer than 32 columns)

    if (tid < D_ROWS) {
        int ar = -1, orw = -1;
        if (tid < rows) {
            ar = (p.a_rows && !T9) ? p.a_rows[m0 + tid] : m0 + tid;
            orw = p.out_rows ? p.out_rows[m0 + tid] : m0 + tid;
            if (p.relu == 1 || (p.relu == 2 && p.relu_if_neg[m0 + tid] < 0)) orw |= D_ROW_RELU;
        }
        int cls = 0;
        if (T9 && tid < rows && p.shift_classes > 1) {
            const int q = p.pix_map[m0 + tid] % (p.Ho * p.Wo);
            const int oy = q / p.Wo, ox = q - oy * p.Wo;
            const int top = oy * p.stride - 1 < 0, bot = oy * p.stride + 1 >= p.Hi;
            const int lef = ox * p.stride - 1 < 0, rig = ox * p.stride + 1 >= p.Wi;
            cls = ((top | (bot << 1)) * 4 + (lef | (rig << 1))) * p.cout;
        }
        LDN_DCHECK(tid >= rows || (ar >= -1 && (orw & (D_ROW_RELU - 1)) >= 0), 501);
        s_arow[tid] = ar;
        s_orow[tid] = orw;
        if (T9) s_cls[tid] = cls;
    }
    if (T9) {      // (round 6: all of a thread's table entries requested before the first is stored -- a plain loop is one memory latency per entry)
        constexpr int NE = (D_ROWS * 9 + NTHR - 1) / NTHR;
        int tv[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int i = tid + NTHR * u, r = i / 9;
            tv[u] = (i < D_ROWS * 9 && r < rows) ? p.a_rows[(size_t)(m0 + r) * 9 + (i - r * 9)] : -1;
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
            if (tid + NTHR * u < D_ROWS * 9) s_atap[tid + NTHR * u] = tv[u];
    }
    __syncthreads();

    const bool active = wave * 32 < rows;
    const int cpt = ceil_div(p.cin, 32);                         // chunks per tap (cin % 32 != 0: the tail of a tap's last chunk is zero-filled)
    const int wrow = (T9 ? 9 : 1) * (p.cin / 8);                 // octets per weight row
    const int nchunks = (T9 ? 9 : 1) * cpt;
    const unsigned lds_ring = d_lds_off(s_ring);
    // ... of which this wave issues nwi: with NT = 160 (NSUB 5) the last instruction only exists for the waves whose 8 rows are inside the tile
    const int nwi = (NT % WR == 0 || (NWI - 1) * WR + wave * 8 < NT) ? NWI : NWI - 1;
    const int per_chunk = (active ? 4 : 0) + nwi;
    long asrc[4];                                                // this lane's four source rows (element offsets), -1 = zero row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ar = s_arow[wave * 32 + i * 8 + (lane >> 3)];
        asrc[i] = ar >= 0 ? (long)ar * p.lda : -1;
    }
    auto dma_chunk = [&](int c) {
        const unsigned slot = lds_ring + (c % D) * SLOT;
        const int tap = T9 ? c / cpt : 0, ck = c - tap * cpt;    // K position = tap * cin + 32 ck
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave * 32 + i * 8 + (lane >> 3);
                const int ls = (lane & 7) ^ ((r >> 1) & 7);
                long off = asrc[i];
                if (T9) {
                    const int ar = s_atap[r * 9 + tap];
                    off = ar >= 0 ? (long)ar * p.lda : -1;
                }
                const float* src = (off >= 0 && ck * 32 + ls * 4 < p.cin) ? p.a + off + ck * 32 + ls * 4 : g_dense_zero;
                d_dma16(src, slot + (wave * 32 + i * 8) * 128);
            }
        }
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            if (i >= nwi) break;
            const int r = i * WR + wave * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ ((r >> 1) & 7);
            const int n = n0 + r;
            const unsigned char* src = (r < NT && n < p.cout && ck * 8 + ls < p.cin / 4)
                                           ? p.ws + ((long)n * wrow + tap * (p.cin / 8) + ck * 4) * 32 + ls * 16
                                                               : reinterpret_cast<const unsigned char*>(g_dense_zero);
            d_dma16(src, slot + (D_ROWS + i * WR + wave * 8) * 128);      // (i < nwi: the wave's 8 rows are inside the tile)
        }
    };
    auto dma_dummy = [&](int c) {
        const unsigned slot = lds_ring + (c % D) * SLOT;
        for (int i = 0; i < per_chunk; ++i) d_dma16(g_dense_zero, slot + (D_ROWS + wave * 8) * 128);
    };
    // instruction k (0 .. 3: this wave's x rows, 4 ..: its weight rows) of chunk c, for ACTIVE waves; a chunk beyond the K range
    // gets a dummy so that the per-iteration count the vmcnt waits rely on stays constant
    auto dma_one = [&](int c, int k) {
        const unsigned slot = lds_ring + (c % D) * SLOT;
        const bool real = c < nchunks;
        const int tap = T9 ? c / cpt : 0, ck = c - tap * cpt;
        const void* src = g_dense_zero;
        unsigned dst = slot + (D_ROWS + wave * 8) * 128;
        if (k < 4) {
            const int r = wave * 32 + k * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ ((r >> 1) & 7);
            long off = asrc[k];
            if (T9) {
                const int ar = s_atap[r * 9 + min(tap, 8)];
                off = ar >= 0 ? (long)ar * p.lda : -1;
            }
            if (real && off >= 0 && ck * 32 + ls * 4 < p.cin) src = p.a + off + ck * 32 + ls * 4;
            if (real) dst = slot + (wave * 32 + k * 8) * 128;
        } else {
            const int i = k - 4;
            const int r = i * WR + wave * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ ((r >> 1) & 7);
            const int n = n0 + r;
            if (real && r < NT && n < p.cout && ck * 8 + ls < p.cin / 4) src = p.ws + ((long)n * wrow + tap * (p.cin / 8) + ck * 4) * 32 + ls * 16;
            if (real && i * WR + wave * 8 < NT) dst = slot + (D_ROWS + i * WR + wave * 8) * 128;
        }
        d_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
    };

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int c = 0; c < D - 1; ++c) { if (c < nchunks) dma_chunk(c); else dma_dummy(c); }
    const unsigned xrow = (unsigned)(wave * 32 + l31), xsw = (xrow >> 1) & 7u;
    const unsigned wsw = ((unsigned)l31 >> 1) & 7u;
#ifdef LDN_TRACE
    unsigned long long d0, d1, d2, d3, d4, d5, a_wait = 0, a_bar = 0, a_issue = 0, a_prep = 0, a_mfma = 0, d_start, d_loop;
    DT(d_start)
#endif
    // B operands of both K16 steps of a chunk: the wave's 32 rows, split into bf16 hi / lo once for all n-subtiles.  The rows are staged
    // by THIS wave's own DMA, so they need its vmcnt only, not the workgroup barrier (that one is for the weights, staged by all waves):
    // full tiles read and split the NEXT chunk's rows behind the current chunk's MFMA steps (PRE), off the barrier-to-MFMA path.
    bf16x8 bh[2], bl[2];       // F32: bh = the lane's k-slots 0-3, bl = k-slots 4-7 as raw floats (bit patterns)
    auto load_b = [&](int c) {
        const unsigned char* xs = s_ring + (c % D) * SLOT;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned sl = 4u * half + 2u * h;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + xrow * 128 + ((sl ^ xsw) << 4));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + xrow * 128 + (((sl + 1) ^ xsw) << 4));
            if constexpr (F32 || PS) {     // (PS: 16-byte unit 2 o = the octet's 8 hi, unit 2 o + 1 = its 8 lo -- the fragment as it is)
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
    };
#ifndef LDN_DENSE_NO_PRE
    constexpr bool PRE = FULL;
#else
    constexpr bool PRE = false;
#endif
    // own rows of a chunk = the first four DMA instructions of that chunk: landed once at most (chunks in flight) x per_chunk - 4 are outstanding
    const int own_landed = per_chunk * (D - 1) - 4;
    if (PRE && active && nchunks > 0) {
        d_wait_vm_rt(own_landed <= 16 ? own_landed : 0);
        load_b(0);
    }
    for (int c = 0; c < nchunks; ++c) {
        DT(d0)
        d_wait_vm_rt(per_chunk * (D - 2));
        DT(d1)
        d_lds_barrier();
        DT(d2)
        // full tiles, computing waves: the DMA instructions of chunk c + D - 1 are issued one per MFMA step below (the texture
        // addresser takes 16 cycles per 1-KB instruction: issued in a burst after the barrier they cost every wave ~800 cycles during
        // which no MFMA runs); other waves / ragged tiles issue them here
        if (!(FULL && active)) { if (c + D - 1 < nchunks) dma_chunk(c + D - 1); else dma_dummy(c + D - 1); }
        DT(d3)
        if (!active) continue;
        const unsigned char* ws = s_ring + (c % D) * SLOT + D_ROWS * 128;
        if (!PRE) load_b(c);
        // one K16 step of n-subtile j_: bf16x3 = three 32x32x16 products of the hi / lo halves; F32 = eight 32x32x2 products of raw floats
#define LDN_DENSE_STEP(ACC_, AH_, AL_, BH_, BL_) \
        if constexpr (F32) { \
            const f32x4 a0_ = __builtin_bit_cast(f32x4, AH_), a1_ = __builtin_bit_cast(f32x4, AL_); \
            const f32x4 b0_ = __builtin_bit_cast(f32x4, BH_), b1_ = __builtin_bit_cast(f32x4, BL_); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[i_], b0_[i_], ACC_, 0, 0, 0); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[i_], b1_[i_], ACC_, 0, 0, 0); \
        } else { \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL_, BH_, ACC_, 0, 0, 0); \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BL_, ACC_, 0, 0, 0); \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BH_, ACC_, 0, 0, 0); \
        }
#ifdef LDN_TRACE
        asm volatile("" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
        DT(d4)
#endif
        auto frag = [&](int step, bf16x8& ah, bf16x8& al) {   // step = half * NSUB + j: weight row 32 j + l31, octet 2 half + h
            const int half = step / NSUB, j = step - half * NSUB;
            const unsigned sl = 4u * half + 2u * h;
#if LDN_DENSE_ABLATE & 8
            if (step > 1) return;
#endif
            ah = *reinterpret_cast<const bf16x8*>(ws + (32 * j + l31) * 128 + ((sl ^ wsw) << 4));
            al = *reinterpret_cast<const bf16x8*>(ws + (32 * j + l31) * 128 + (((sl + 1) ^ wsw) << 4));
        };
        if constexpr (FULL) {
            // one flat, branch-free sequence of 2 NSUB steps whose weight fragments are double-buffered: the two ds_read_b128 of step
            // s + 1 are in flight during the three MFMAs of step s.  (Left to itself hipcc issues read, s_waitcnt lgkmcnt(0), MFMA
            // per fragment behind a branch per subtile: a full LDS latency per 96 MFMA cycles, hidden only by the SIMD's other wave.)
            bf16x8 ah[2], al[2];
            frag(0, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2 * NSUB; ++st) {
                if (st + 1 < 2 * NSUB) frag(st + 1, ah[(st + 1) & 1], al[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);     // the schedule is pinned: the next step's reads are issued BEFORE this step's MFMAs
                const int half = st / NSUB, j = st - half * NSUB;
#if LDN_DENSE_ABLATE & 4
                asm volatile("" : "+v"(acc[j]) : "v"(al[st & 1]), "v"(ah[st & 1]), "v"(bh[half]), "v"(bl[half]));
#else
                LDN_DENSE_STEP(acc[j], ah[st & 1], al[st & 1], bh[half], bl[half])
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (st < 4 + nwi) {                   // one DMA instruction of the next chunk behind this step's MFMAs
                    dma_one(c + D - 1, st);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int k = 2 * NSUB; k < 4 + NWI; ++k)
                if (k < 4 + nwi) dma_one(c + D - 1, k);   // (64-column tiles: five instructions, four steps)
            if (PRE && c + 1 < nchunks) {              // the next chunk's rows: landed (own DMA), read and split behind this chunk's MFMAs
                d_wait_vm_rt(own_landed <= 16 ? own_landed : 0);
                load_b(c + 1);
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < NSUB; ++j) {
                    if (j < nsub) {
                        bf16x8 ah, al;
                        frag(half * NSUB + j, ah, al);
                        LDN_DENSE_STEP(acc[j], ah, al, bh[half], bl[half])
                    }
                }
            }
        }
#ifdef LDN_TRACE
        asm volatile("" : "+v"(acc[0]));
        DT(d5)
        a_wait += d1 - d0; a_bar += d2 - d1; a_issue += d3 - d2; a_prep += d4 - d3; a_mfma += d5 - d4;
#endif
    }
#undef LDN_DENSE_STEP
#ifdef LDN_TRACE
    DT(d_loop)
#endif
    d_wait_vm<0>();
    d_lds_barrier();       // every wave is out of the ring: it becomes the per-wave 32 x 32 transpose scratch
    if (!active) return;

    // ---- epilogue
    dense_epilogue<NSUB, T9, FULL, OF>(p, acc, s_arow, s_orow, s_cls, reinterpret_cast<float*>(s_ring + wave * 4096), wave, lane, n0, nsub);
#if defined(LDN_TRACE)
    if (g_dense_trace && lane == 0) {
        unsigned long long dend;
        DT(dend)
        unsigned long long* r = g_dense_trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        r[0] = a_wait; r[1] = a_bar; r[2] = a_issue; r[3] = a_prep; r[4] = a_mfma; r[5] = d_loop - d_start; r[6] = dend - d_loop; r[7] = nchunks;
    }
#endif
}

// k_dense2 (round 5) -- k_dense's arithmetic (the same products in the same order: bit-identical results) on the staging structure that took the
// packed 3x3 from 21 % to 35 % matrix-pipe busy (k_rows3, csrc/ldn_rows3.hip), for the whole-tile bf16x3 launches:
//   * only the WEIGHT tile is shared between waves: ring of two slots of one K step (64 wide for tiles of <= 128 columns, 32 wide above), ONE
//     workgroup barrier per step (k_dense: one per 32-wide chunk over a common ring that also held the activation rows);
//   * a wave's 32 activation rows are staged by that wave alone into a private double buffer of K32 sub-chunks, ordered by its own vmcnt;
//   * B fragments are double-buffered in registers: sub-chunk s + 1 is read -- and, unless the rows arrive pre-split (PS), split into bf16
//     hi / lo -- BETWEEN the MFMA steps of sub-chunk s, the LDS-DMA of sub-chunk s + 3 reuses its slot right behind; the matrix pipe never
//     waits for a split, a DMA issue or a fragment read at a chunk boundary.
// Whole column tiles only (cout % NT == 0), cin a multiple of the K step, cin <= 2048; T9 = the 3x3 through a neighbour table (tap-major K).
// RAG (1x1, fp32 rows): any cin % 8 == 0 (the K tail of the last sub-chunk / step is zero-sourced in both DMA streams) and a ragged last column tile (columns >= cout are
// neither staged nor stored) -- LAD-RegNet's 144- and 784-wide layers.
// AG (1x1, fp32 rows, no gather): every A row is scaled by the gate vector of its image (p.a_gate[row / p.gate_rows]) between its LDS read and its
// split -- the values the separate k_rows_scale pass would have written, so the results are bit-identical to scaling first; the gate vectors of
// the (few) images a 256-row tile touches sit in LDS.
template <int NSUB, bool T9, bool PS, bool OF, bool FEAT = false, bool RAG = false, bool AG = false>
__global__ __launch_bounds__(512, 2) void k_dense2(const DenseArgs p) {
    static_assert(!RAG || (!T9 && !PS && !OF), "the ragged form is the plain 1x1");
    static_assert(!AG || (!T9 && !PS), "the gated form reads fp32 rows of a 1x1");
    constexpr int NT = NSUB * 32;
    constexpr int KS = NSUB <= 4 ? 64 : 32;               // K elements of one weight step
    constexpr int SPS = KS / 32;                          // K32 sub-chunks per step
    constexpr int WROWB = KS * 4;                         // bytes of a staged weight row
    constexpr int UPR = WROWB / 16;                       // its 16-byte units (16 / 8), XOR-swizzled by the row
    constexpr int RPI = 1024 / WROWB;                     // weight rows per 1-KB DMA instruction (4 / 8)
    constexpr int NWI = (NT + 8 * RPI - 1) / (8 * RPI);   // weight DMA instructions per wave and step
    constexpr int WSLOT = NWI * 8 * 1024;                 // (160-column tiles: padded to whole instructions)
    constexpr int RSLOT = 32 * 128;                       // one wave's 32 rows of one K32 sub-chunk
    constexpr int D_ROWS = 256, NST = 2 * NSUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* const s_arow = reinterpret_cast<int*>(smem);
    int* const s_orow = s_arow + D_ROWS;
    int* const s_cls = s_orow + D_ROWS;                   // (T9 only)
    int* const s_atap = s_cls + D_ROWS;                   // (T9 only) [256][9]
    unsigned char* const s_w = smem + (T9 ? 12 : 2) * D_ROWS * 4;
    unsigned char* const s_r = s_w + 2 * WSLOT;
    float* const s_gate = reinterpret_cast<float*>(s_r + 8 * 2 * RSLOT);      // (AG) [slots of this tile][cin rounded up to 32, zero-padded]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
    const int nt = slot_i % p.ntn, mt = (slot_i / p.ntn) * 8 + xcd;
    if (mt >= p.mtn) return;
    const int M = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap;
    const int m0 = mt * D_ROWS;
    if (m0 >= M) return;
    const int rows = min(D_ROWS, M - m0);
    const int n0 = nt * NT;

    if (tid < D_ROWS) {
        int ar = -1, orw = -1;
        if (tid < rows) {
            ar = (p.a_rows && !T9) ? p.a_rows[m0 + tid] : m0 + tid;
            orw = p.out_rows ? p.out_rows[m0 + tid] : m0 + tid;
            if (p.relu == 1 || (p.relu == 2 && p.relu_if_neg[m0 + tid] < 0)) orw |= D_ROW_RELU;
        }
        int cls = 0;
        if (T9 && tid < rows && p.shift_classes > 1) {
            const int q = p.pix_map[m0 + tid] % (p.Ho * p.Wo);
            const int oy = q / p.Wo, ox = q - oy * p.Wo;
            const int top = oy * p.stride - 1 < 0, bot = oy * p.stride + 1 >= p.Hi;
            const int lef = ox * p.stride - 1 < 0, rig = ox * p.stride + 1 >= p.Wi;
            cls = ((top | (bot << 1)) * 4 + (lef | (rig << 1))) * p.cout;
        }
        LDN_DCHECK(tid >= rows || (ar >= -1 && (orw & (D_ROW_RELU - 1)) >= 0), 501);
        s_arow[tid] = ar;
        s_orow[tid] = orw;
        if (T9) s_cls[tid] = cls;
    }
    if (T9) {      // (round 6: all of a thread's table entries requested before the first is stored -- a plain loop is one memory latency per entry)
        constexpr int NE = (D_ROWS * 9 + 511) / 512;
        int tv[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int i = tid + 512 * u, r = i / 9;
            tv[u] = (i < D_ROWS * 9 && r < rows) ? p.a_rows[(size_t)(m0 + r) * 9 + (i - r * 9)] : -1;
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
            if (tid + 512 * u < D_ROWS * 9) s_atap[tid + 512 * u] = tv[u];
    }
    int goff = 0;                                         // (AG) this lane's row: offset of its image's gate vector in s_gate
    if constexpr (AG) {
        const int gld = round_up(p.cin, 32), q4 = gld >> 2;
        const int slot0 = m0 / p.gate_rows, nsl = (m0 + rows - 1) / p.gate_rows - slot0 + 1;
        for (int i = tid; i < nsl * q4; i += 512) {
            const int sl = i / q4, k = (i - sl * q4) * 4;
            reinterpret_cast<f32x4*>(s_gate)[i] = k < p.cin ? *reinterpret_cast<const f32x4*>(p.a_gate + (size_t)(slot0 + sl) * p.cin + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        goff = ((m0 + min(wave * 32 + (lane & 31), rows - 1)) / p.gate_rows - slot0) * gld + 8 * (lane >> 5);
    }
    __syncthreads();

    const bool active = wave * 32 < rows;
    const int cpt = RAG ? ceil_div(p.cin, 32) : p.cin >> 5;          // K32 sub-chunks per tap (RAG: the last one may be partial)
    const int nsubc = (T9 ? 9 : 1) * cpt, nstep = RAG ? ceil_div(nsubc, SPS) : nsubc / SPS;
    const int nsub_t = RAG ? min(NSUB, ceil_div(p.cout - n0, 32)) : NSUB;      // n-subtiles of this column tile
    const long wrow = (long)(T9 ? 9 : 1) * p.cin * 4;     // bytes per weight row
    const unsigned lds_w = d_lds_off(s_w), lds_r = d_lds_off(s_r) + (unsigned)wave * 2u * RSLOT;
    unsigned char* const my_r = s_r + wave * 2 * RSLOT;
    const unsigned char* const zrow = reinterpret_cast<const unsigned char*>(g_dense_zero);

    // per-lane sources (see k_rows3): weights advance by one step's bytes per step, steps past the end re-read the last one
    const unsigned char* wsrc[NWI];
    int wku[NWI];
#pragma unroll
    for (int k = 0; k < NWI; ++k) {
        wku[k] = 0;
        const int rr = 8 * RPI * k + RPI * wave + lane / UPR;
        const int sw = UPR == 16 ? (rr & 15) : ((rr >> 1) & 7);
        wsrc[k] = (rr < NT && (!RAG || n0 + rr < p.cout)) ? p.ws + (long)(n0 + rr) * wrow + (((lane % UPR) ^ sw) << 4) : zrow;
        if (RAG) wku[k] = ((lane % UPR) ^ sw) * 4;         // k offset (floats) of this lane's source unit inside a step
    }
    auto dma_w = [&](int S, int k) {
        const bool pad = NT % (8 * RPI) != 0 && 8 * RPI * k + RPI * wave >= NT;       // (wave-uniform: a padding instruction of a 160-column tile)
        const int Sc = min(S, nstep - 1);
        const bool ktail = RAG && Sc * KS + wku[k] >= p.cin;                          // (per lane) this unit lies beyond K: zero
        d_dma16(ktail ? zrow : wsrc[k] + (pad || wsrc[k] == zrow ? 0l : (long)Sc * WROWB),
                (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_w + (unsigned)(S & 1) * WSLOT + (unsigned)(8 * RPI * k + RPI * wave) * WROWB)));
    };
    const unsigned char* rsrc[4];
    int tap_r = 0, ck_r = 0;
    auto load_tap = [&](int tap) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * k + (lane >> 3);
            const int ar = T9 ? s_atap[(wave * 32 + r) * 9 + tap] : s_arow[wave * 32 + r];
            const unsigned char* base = ar >= 0 ? reinterpret_cast<const unsigned char*>(p.a) + (long)ar * p.lda * 4 : zrow;
            rsrc[k] = base + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
    };
    auto dma_r = [&](int s, int k) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_r + (unsigned)(s & 1) * RSLOT + (unsigned)k * 1024u));
        if constexpr (RAG) {
            const int r = 8 * k + (lane >> 3);
            const int ku = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
            d_dma16(ck_r * 32 + ku < p.cin ? rsrc[k] + ck_r * 128 : zrow, dst);
        } else {
            d_dma16(rsrc[k] + ck_r * 128, dst);
        }
    };
    auto next_r = [&]() {
        if constexpr (RAG) { ++ck_r; return; }          // (no wrap: sub-chunks past K are zero-sourced)
        if (++ck_r == cpt) {
            ck_r = 0;
            if (T9) {
                tap_r = min(tap_r + 1, 8);
                load_tap(tap_r);
            }
        }
    };

    if (!active) {
        for (int k = 0; k < NWI; ++k) dma_w(0, k);
        for (int S = 0; S < nstep; ++S) {
            d_wait_vm<0>();
            d_lds_barrier();
            for (int k = 0; k < NWI; ++k) dma_w(S + 1, k);
        }
        d_wait_vm<0>();
        return;
    }

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    bf16x8 bh[2][2], bl[2][2];          // B fragments: [register set = sub-chunk parity][K16 half]
    f32x4 raw[4];                       // (!PS) a sub-chunk's fp32 values between their LDS read and their split
    f32x4 gq[4];                        // (AG) the gate values of the same positions
    const unsigned xsw = ((unsigned)l31 >> 1) & 7u;
    const unsigned wsw = UPR == 16 ? ((unsigned)l31 & 15u) : (((unsigned)l31 >> 1) & 7u);
    auto read_b = [&](int s, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
        const unsigned char* xs = my_r + (s & 1) * RSLOT + l31 * 128;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned sl = 4u * half + 2u * h;
            if constexpr (PS) {
                dh[half] = *reinterpret_cast<const bf16x8*>(xs + ((sl ^ xsw) << 4));
                dl[half] = *reinterpret_cast<const bf16x8*>(xs + (((sl + 1) ^ xsw) << 4));
            } else {
                raw[2 * half] = *reinterpret_cast<const f32x4*>(xs + ((sl ^ xsw) << 4));
                raw[2 * half + 1] = *reinterpret_cast<const f32x4*>(xs + (((sl + 1) ^ xsw) << 4));
                if constexpr (AG) {      // (sub-chunks past K are read ahead and never multiplied: any finite vector does)
                    const float* g = s_gate + goff + min(s, cpt - 1) * 32 + 16 * half;
                    gq[2 * half] = *reinterpret_cast<const f32x4*>(g);
                    gq[2 * half + 1] = *reinterpret_cast<const f32x4*>(g + 4);
                }
            }
        }
    };
    auto split_b = [&](int half, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
        if constexpr (!PS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = e < 4 ? raw[2 * half][e] : raw[2 * half + 1][e - 4];
                if constexpr (AG) {      // the product ROUNDED to fp32, like the stored one (opaque to the compiler: no FMA contraction into the lo part below)
                    v *= e < 4 ? gq[2 * half][e] : gq[2 * half + 1][e - 4];
                    asm("" : "+v"(v));
                }
                const __bf16 hb = (__bf16)v;
                dh[half][e] = hb;
                dl[half][e] = (__bf16)(v - (float)hb);
            }
        }
    };

    // prologue: W(0), R(0), R(1); B(0) -> registers; R(2) into R(0)'s slot
    load_tap(0);
#pragma unroll
    for (int k = 0; k < NWI; ++k) dma_w(0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(0, k);
    next_r();
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(1, k);
    next_r();
    d_wait_vm<4>();
    read_b(0, bh[0], bl[0]);
    split_b(0, bh[0], bl[0]);
    split_b(1, bh[0], bl[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(2, k);
    next_r();
    if constexpr (SPS == 1) d_wait_vm<4>();         // (one sub-chunk per step: R(1) has landed before the loop's counted waits start)

    // in-order completion: what may still be outstanding at each wait (see k_rows3):
    //   two sub-chunks per step:  W(S): vmcnt(8);  R(s + 1): vmcnt(4 + NWI)        one per step:  W(S): vmcnt(4);  R(s + 1): vmcnt(4 + 2 NWI)
    // one weight step: barrier, then its SPS sub-chunks; CUR0 = register set of its first sub-chunk (compile-time)
    auto step = [&](int S, auto cur0_c) {
        constexpr int CUR0 = decltype(cur0_c)::value;
        d_wait_vm<SPS == 2 ? 8 : 4>();
        d_lds_barrier();
        const unsigned char* wsl = s_w + (S & 1) * WSLOT + l31 * WROWB;
        // the body of one sub-chunk with B set CUR: MFMA steps with, between them, W(S + 1) (first sub-chunk of the step), the read (+ split) of
        // sub-chunk s + 1 into the other set, and R(s + 3)
        auto body = [&](auto sub_c, auto cur_c) {
            constexpr int sub = decltype(sub_c)::value, CUR = decltype(cur_c)::value;
            const int s = SPS * S + sub;
            auto frag = [&](int st, bf16x8& ah, bf16x8& al) {
                const int half = st / NSUB, j = st - half * NSUB;
                const unsigned uu = 2u * ((SPS == 2 ? 4u * sub : 0u) + 2u * half + h);
                ah = *reinterpret_cast<const bf16x8*>(wsl + j * (32 * WROWB) + ((uu ^ wsw) << 4));
                al = *reinterpret_cast<const bf16x8*>(wsl + j * (32 * WROWB) + (((uu + 1) ^ wsw) << 4));
            };
            bf16x8 ah[2], al[2];
            frag(0, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int rd_at = sub == 0 ? (NWI < NST - 1 ? NWI : NST - 2) : 0;
            int r_done = 0, w_done = 0, sp_done = PS ? 2 : 0;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) frag(st + 1, ah[(st + 1) & 1], al[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int half = st / NSUB, j = st - half * NSUB;
                if (!RAG || j < nsub_t) {       // (wave-uniform; a ragged last tile skips its empty n-subtiles)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[st & 1], bh[CUR][half], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st & 1], bl[CUR][half], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st & 1], bh[CUR][half], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (sub == 0 && st < rd_at) {
                    if (w_done < NWI) dma_w(S + 1, w_done++);
                } else if (st == rd_at) {
                    if (sub == 0)
                        while (w_done < NWI) dma_w(S + 1, w_done++);
                    d_wait_vm<SPS == 2 ? 4 + NWI : 4 + 2 * NWI>();
                    read_b(s + 1, bh[CUR ^ 1], bl[CUR ^ 1]);
                } else if (sp_done < 2) {
                    split_b(sp_done++, bh[CUR ^ 1], bl[CUR ^ 1]);      // (waits for the raw reads: they were issued a whole MFMA step ago)
                } else {
                    if (r_done == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (r_done < 4) dma_r(s + 3, r_done++);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            while (sp_done < 2) split_b(sp_done++, bh[CUR ^ 1], bl[CUR ^ 1]);
            if (r_done == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            while (r_done < 4) dma_r(s + 3, r_done++);
            next_r();
            __builtin_amdgcn_sched_barrier(0);
        };
        body(std::integral_constant<int, 0>{}, std::integral_constant<int, CUR0>{});
        if constexpr (SPS == 2) body(std::integral_constant<int, 1>{}, std::integral_constant<int, CUR0 ^ 1>{});
    };
    if constexpr (SPS == 2) {
        for (int S = 0; S < nstep; ++S) step(S, std::integral_constant<int, 0>{});
    } else {      // one sub-chunk per step: the register sets alternate between steps -- two steps per iteration (cin % 64 == 0: an even count)
        for (int S = 0; S < nstep; S += 2) {
            step(S, std::integral_constant<int, 0>{});
            if (!RAG || S + 1 < nstep) step(S + 1, std::integral_constant<int, 1>{});
        }
    }
    d_wait_vm<0>();        // the trailing re-reads have landed: this wave's row slots become its 32 x 32 transpose scratch (private: no barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dense_epilogue<NSUB, T9, !RAG, OF, FEAT || T9>(p, acc, s_arow, s_orow, s_cls, reinterpret_cast<float*>(my_r), wave, lane, n0, nsub_t);
}

// LayerNorm statistics of the rows of a [rows, C] matrix: stats[r] = {mean, 1 / sqrt(var + eps)} (biased variance, as nn.LayerNorm),
// one wave per row, the row held in registers (two passes over registers, one over memory).
__global__ __launch_bounds__(256) void k_row_stats(const float* __restrict__ x, int ld, int rows, int C, float eps, float* __restrict__ stats,
                                                    const int32_t* __restrict__ list, const int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (list) {      // only the listed rows (first *count entries): the tokens a token-skip block works on
        if (r >= min(*count, rows)) return;
        r = list[r];
    }
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ld;
    f32x4 v[8];     // C <= 2048 (ViT-H: 1280)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < C ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)r) = float2{mean, rsqrtf(q / (float)C + eps)};
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_dense)

template <int NSUB, bool T9, bool FULL, bool F32 = false, int R = 256, bool PS = false, bool OF = false>
static int launch_dense_f(DenseArgs& a, hipStream_t st) {
    constexpr int NT = NSUB * 32;
    constexpr int D = (NSUB >= 6 || R == 128) ? 2 : 3;
    const size_t lds = (size_t)(T9 ? 12 : 2) * R * 4 + (size_t)D * (R + NT) * 128;
    a.ntn = ceil_div(a.cout, NT);
    a.mtn = ceil_div(a.m_cap, R);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_dense<NSUB, T9, FULL, F32, R, PS, OF>), lds), "k_dense: cannot reserve %zu B of LDS", lds);
    const unsigned grid = (unsigned)round_up(a.mtn, 8) * a.ntn;
    hipLaunchKernelGGL((k_dense<NSUB, T9, FULL, F32, R, PS, OF>), dim3(grid), dim3(2 * R), lds, st, a);
    LDN_CHECK_LAUNCH("k_dense");
    return LDN_OK;
}

template <int NSUB, bool T9, bool PS, bool OF, bool FEAT = false, bool RAG = false, bool AG = false>
static int launch_dense2(DenseArgs& a, hipStream_t st) {
    if constexpr (!FEAT && !T9 && !PS && !OF && !RAG) {       // the plain 1x1 form has a second instantiation with the rarely used epilogue terms (T9: always in)
        if (a.relu == 3 || a.ln_stats || a.chmask || a.post_sub) return launch_dense2<NSUB, T9, PS, OF, true>(a, st);
    }
    constexpr int NT = NSUB * 32, KS = NSUB <= 4 ? 64 : 32, WROWB = KS * 4, RPI = 1024 / WROWB, NWI = (NT + 8 * RPI - 1) / (8 * RPI);
    const size_t lds = (size_t)(T9 ? 12 : 2) * 256 * 4 + 2 * (size_t)NWI * 8 * 1024 + 8 * 2 * (size_t)32 * 128 +
                       (AG ? (size_t)(255 / a.gate_rows + 2) * round_up(a.cin, 32) * 4 : 0);      // (AG: the gate vectors of the images a tile touches)
    a.ntn = ceil_div(a.cout, NT);
    a.mtn = ceil_div(a.m_cap, 256);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_dense2<NSUB, T9, PS, OF, FEAT, RAG, AG>), lds), "k_dense2: cannot reserve %zu B of LDS", lds);
    const unsigned grid = (unsigned)round_up(a.mtn, 8) * a.ntn;
    hipLaunchKernelGGL((k_dense2<NSUB, T9, PS, OF, FEAT, RAG, AG>), dim3(grid), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_dense2");
    return LDN_OK;
}
// can k_dense2 run this launch with tiles of ns * 32 columns?  (whole column tiles, cin a multiple of the tile's K step, a zero row of cin floats)
static bool dense2_ok(int ns, int taps, int cin, int cout) {
    static const bool on = !(getenv("LDN_DENSE_V2") && atoi(getenv("LDN_DENSE_V2")) == 0);
    static const int max_ns = getenv("LDN_DENSE_V2_MAXNS") ? atoi(getenv("LDN_DENSE_V2_MAXNS")) : 8;      // (tuning: widest tile that takes k_dense2)
    return on && ns <= max_ns && cout % (ns * 32) == 0 && cin % 64 == 0 && cin <= 2048 && (taps == 1 || ns <= 4);
}

template <int NSUB, bool T9, bool F32 = false>
static int launch_dense(DenseArgs& a, hipStream_t st) {
    return a.cout % (NSUB * 32) == 0 ? launch_dense_f<NSUB, T9, true, F32>(a, st) : launch_dense_f<NSUB, T9, false, F32>(a, st);
}

}  // namespace ldn

using namespace ldn;

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_dense_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dense_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif

static int conv_rows_dense_impl(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                                const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                                const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr,
                                float* out, int ldo, const float* post_sub, const float* chan_mask, int rows_per_image,
                                int shift_classes, const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride,
                                const float* ln_stats, const float* ln_c1, bool f32, void* stream, float* pool = nullptr, int pool_S = 0,
                                int pool_Sx = 0, bool ps = false, bool of = false);

// Advisory: about how many rows the NEXT ldn_conv_rows_split / _pool call of this thread will find in its device-side count (which the
// host cannot read without a synchronisation) -- e.g. the count of the previous forward.  Used to choose the tile width only.
static thread_local long g_rows_hint = -1;
extern "C" int ldn_hint_rows(int rows) { g_rows_hint = rows; return LDN_OK; }

extern "C" int ldn_conv_rows_split(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                                   const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                                   const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr,
                                   float* out, int ldo, const float* post_sub, const float* chan_mask, int rows_per_image,
                                   int shift_classes, const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride,
                                   const float* ln_stats, const float* ln_c1, void* stream) {
    return conv_rows_dense_impl(a, lda, a_rows, taps, m_count, m_cap, w_split, cin, cout, scale, shift, relu, relu_if_neg, out_rows, residual, ldr,
                                out, ldo, post_sub, chan_mask, rows_per_image, shift_classes, pix_map, Hi, Wi, Ho, Wo, stride, ln_stats, ln_c1, false, stream);
}

extern "C" int ldn_conv_rows_f32(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                                 const float* w, int cin, int cout, const float* scale, const float* shift, int relu,
                                 const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr,
                                 float* out, int ldo, const float* post_sub, const float* chan_mask, int rows_per_image,
                                 int shift_classes, const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride,
                                 const float* ln_stats, const float* ln_c1, void* stream) {
    return conv_rows_dense_impl(a, lda, a_rows, taps, m_count, m_cap, w, cin, cout, scale, shift, relu, relu_if_neg, out_rows, residual, ldr,
                                out, ldo, post_sub, chan_mask, rows_per_image, shift_classes, pix_map, Hi, Wi, Ho, Wo, stride, ln_stats, ln_c1, true, stream);
}

// The 1x1 form with the pooled patch means of its output as a by-product (the fused spatial masker): `pool`
// [B][S * Sx][cout] receives, for every patch this launch writes, the mean of the final output (after residual and ReLU) over the
// patch's Ho / S x Wo / Sx pixels (4 or 16).  The packed rows must list whole patches, Ho / S * Wo / Sx consecutive rows each
// (ldn_mask_plan with patch_major = 1); out_rows are flat pixel indices b * Ho * Wo + y * Wo + x.  math_mode: 0 = fp32 weights /
// arithmetic (w = plain [cout][cin] floats), 1 = bf16x3 (w = the pre-split rows of ldn_split_rows_weight).
extern "C" int ldn_conv_rows_pool(const float* a, int lda, const int32_t* a_rows, const int32_t* m_count, int m_cap, const void* w,
                                  int cin, int cout, const float* scale, const float* shift, int relu, const int32_t* relu_if_neg,
                                  const int32_t* out_rows, const float* residual, int ldr, float* out, int ldo, float* pool, int S,
                                  int Sx, int Ho, int Wo, int math_mode, void* stream) {
    LDN_REQUIRE(pool, "ldn_conv_rows_pool: null pool");
    LDN_REQUIRE(math_mode == 0 || math_mode == 1, "ldn_conv_rows_pool: math_mode must be 0 (fp32) or 1 (bf16x3)");
    return conv_rows_dense_impl(a, lda, a_rows, 1, m_count, m_cap, w, cin, cout, scale, shift, relu, relu_if_neg, out_rows, residual, ldr,
                                out, ldo, nullptr, nullptr, 0, 1, nullptr, 0, 0, Ho, Wo, 1, nullptr, nullptr, math_mode == 0, stream, pool, S, Sx);
}

// The 1x1 form on PRE-SPLIT rows (round 5: the packed spatial / layer path keeps h1 and h2 pre-split between its three launches, so no
// operand is split in a K loop).  a_presplit: the rows of `a` are [cin / 8][8 hi | 8 lo] bf16 (lda still counts 4-byte elements);
// out_presplit: the rows of `out` are written in that layout (ldo likewise; no residual / pool with it).  Everything else as
// ldn_conv_rows_split with taps = 1; bf16x3 arithmetic; cin % 32 == 0, cout % 64 == 0.  With pool != NULL the pooled patch means of
// the output are left as by ldn_conv_rows_pool (S, Sx, Ho, Wo: its arguments).
extern "C" int ldn_conv_rows_ps(const float* a, int lda, int a_presplit, const int32_t* a_rows, const int32_t* m_count, int m_cap,
                                const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                                const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr, float* out, int ldo,
                                int out_presplit, float* pool, int S, int Sx, int Ho, int Wo, void* stream) {
    LDN_REQUIRE(a_presplit || out_presplit, "ldn_conv_rows_ps: neither side is pre-split (use ldn_conv_rows_split)");
    LDN_REQUIRE(cin % 32 == 0 && cout % 64 == 0, "ldn_conv_rows_ps: cin must be a multiple of 32 and cout of 64 (got %d, %d)", cin, cout);
    LDN_REQUIRE(!out_presplit || (!residual && !pool && !out_rows), "ldn_conv_rows_ps: a pre-split output takes no residual, scatter or pooled means");
    return conv_rows_dense_impl(a, lda, a_rows, 1, m_count, m_cap, w_split, cin, cout, scale, shift, relu, relu_if_neg, out_rows, residual, ldr,
                                out, ldo, nullptr, nullptr, 0, 1, nullptr, 0, 0, Ho, Wo, 1, nullptr, nullptr, false, stream, pool, S, Sx,
                                a_presplit != 0, out_presplit != 0);
}

// The 1x1 form with a per-image GATE on its input rows (the excitation of an SE block folded into the conv that follows it,
// laud_regnet.py:196-197 `x = self.se(x); x = self.c(x)`): row r of `a` (rows are whole images of gate_rows rows each, in order: packed
// rows of a layer-skip list, or a dense batch) is multiplied by gate[r / gate_rows][0..cin) before the product.  The products are formed
// from fp32(a * gate), the values a separate scaling pass would store: bit-identical to scaling first.  bf16x3 arithmetic; any cin % 8 == 0
// (<= 2048), any cout % 4 == 0; everything else as ldn_conv_rows_split with taps = 1 and no a_rows.
extern "C" int ldn_conv_rows_gated(const float* a, int lda, const int32_t* m_count, int m_cap, const void* w_split, int cin, int cout,
                                   const float* scale, const float* shift, int relu, const int32_t* relu_if_neg, const int32_t* out_rows,
                                   const float* residual, int ldr, float* out, int ldo, const float* gate, int gate_rows, void* stream) {
    LDN_REQUIRE(a && w_split && shift && out && gate, "ldn_conv_rows_gated: null pointer");
    LDN_REQUIRE(cin > 0 && cin % 8 == 0 && cin <= 2048 && cout > 0 && cout % 4 == 0, "ldn_conv_rows_gated: cin must be a multiple of 8, at most 2048, and cout a multiple of 4 (got %d, %d)", cin, cout);
    LDN_REQUIRE(gate_rows > 0, "ldn_conv_rows_gated: gate_rows must be positive");
    LDN_REQUIRE(lda % 4 == 0 && lda >= cin && ldo % 4 == 0 && ldo >= cout && (!residual || (ldr % 4 == 0 && ldr >= cout)),
                "ldn_conv_rows_gated: strides must be multiples of 4 and cover the row");
    LDN_REQUIRE(relu >= 0 && relu <= 2 && (relu != 2 || relu_if_neg), "ldn_conv_rows_gated: bad relu mode (0 none, 1 ReLU, 2 ReLU on the rows with relu_if_neg < 0)");
    LDN_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)w_split % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 &&
                (uintptr_t)shift % 16 == 0 && (uintptr_t)scale % 16 == 0 && (uintptr_t)gate % 16 == 0, "ldn_conv_rows_gated: pointers must be 16-byte aligned");
    if (m_cap <= 0) return LDN_OK;
    DenseArgs d{a, lda, nullptr, m_count, m_cap, static_cast<const unsigned char*>(w_split), cin, cout, scale, shift, relu,
                relu_if_neg, out_rows, residual, ldr, out, ldo, 1, 1, nullptr, 0, 0, 1, 1, 1, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0,
                gate, gate_rows};
    hipStream_t st = static_cast<hipStream_t>(stream);
    g_rows_hint = -1;
    // the gate vectors of every image a 256-row tile can touch sit in LDS behind the staging buffers (98 - 114 KB): small images x wide layers do not fit
    LDN_REQUIRE((size_t)(255 / gate_rows + 2) * round_up(cin, 32) * 4 <= 44 * 1024,
                "ldn_conv_rows_gated: the gate vectors of the %d images a 256-row tile touches (%d channels each) exceed the 44 KB of LDS left for them "
                "(scale the rows first and use ldn_conv_rows_split)", 255 / gate_rows + 2, cin);
    //  the 128-column tiles stage 130 KB: only 30 KB are left for the gate vectors there -- wider gates take the 160-column
    // tiles (114 KB of staging, ragged last tile), whose 46 KB cover the 44 KB bound above
    const size_t gate_bytes = (size_t)(255 / gate_rows + 2) * round_up(cin, 32) * 4;
    if ((cout % 128 == 0 || (cout > 64 && cout <= 128)) && gate_bytes <= 30 * 1024) return launch_dense2<4, false, false, false, true, true, true>(d, st);
    if (cout <= 64) return launch_dense2<2, false, false, false, true, true, true>(d, st);
    return launch_dense2<5, false, false, false, true, true, true>(d, st);
}

static int conv_rows_dense_impl(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                                const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                                const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr,
                                float* out, int ldo, const float* post_sub, const float* chan_mask, int rows_per_image,
                                int shift_classes, const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride,
                                const float* ln_stats, const float* ln_c1, bool f32, void* stream, float* pool, int pool_S, int pool_Sx,
                                bool ps, bool of) {
    const long hint = g_rows_hint;      // (consumed by this call whatever path it takes)
    g_rows_hint = -1;
    LDN_REQUIRE(a && w_split && shift && out, "ldn_conv_rows_split: null pointer");
    int pool_gy = 0, pool_gx = 0;
    if (pool) {
        LDN_REQUIRE(taps == 1 && cout % 128 == 0 && !post_sub && !chan_mask && !ln_stats && (uintptr_t)pool % 16 == 0,
                    "ldn_conv_rows_pool: 1x1 only, cout a multiple of 128, no post_sub / chan_mask / LayerNorm terms");
        LDN_REQUIRE(pool_S > 0 && pool_Sx > 0 && Ho > 0 && Wo > 0 && Ho % pool_S == 0 && Wo % pool_Sx == 0,
                    "ldn_conv_rows_pool: the %dx%d map must split evenly into %dx%d patches", Ho, Wo, pool_S, pool_Sx);
        pool_gy = Ho / pool_S; pool_gx = Wo / pool_Sx;
        LDN_REQUIRE(pool_gy * pool_gx == 4 || pool_gy * pool_gx == 16, "ldn_conv_rows_pool: patches of 4 or 16 pixels (got %dx%d)", pool_gy, pool_gx);
    }
    LDN_REQUIRE((ln_stats == nullptr) == (ln_c1 == nullptr) && (!ln_stats || taps == 1), "ldn_conv_rows_split: ln_stats and ln_c1 go together (1x1 only)");
    LDN_REQUIRE((uintptr_t)ln_stats % 8 == 0 && (uintptr_t)ln_c1 % 16 == 0, "ldn_conv_rows_split: ln_stats / ln_c1 must be 8 / 16-byte aligned");
    LDN_REQUIRE(cin > 0 && cin % 8 == 0 && cout > 0 && cout % 4 == 0, "ldn_conv_rows_split: cin must be a multiple of 8 and cout of 4 (got %d, %d)", cin, cout);
    LDN_REQUIRE(lda % 4 == 0 && lda >= cin && ldo % 4 == 0 && ldo >= cout && (!residual || (ldr % 4 == 0 && ldr >= cout)),
                "ldn_conv_rows_split: strides must be multiples of 4 and cover the row");
    LDN_REQUIRE(relu >= 0 && relu <= 3 && (relu != 2 || relu_if_neg), "ldn_conv_rows_split: bad relu mode (0 none, 1 ReLU, 2 ReLU on the rows with relu_if_neg < 0, 3 GELU)");
    LDN_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)w_split % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 &&
                (uintptr_t)shift % 16 == 0 && (uintptr_t)scale % 16 == 0, "ldn_conv_rows_split: pointers must be 16-byte aligned");
    LDN_REQUIRE(taps == 1 || (taps == 9 && a_rows), "ldn_conv_rows_split: taps must be 1, or 9 with a neighbour table");
    LDN_REQUIRE(shift_classes == 1 || (shift_classes == 16 && taps == 9 && pix_map && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0 && stride >= 1),
                "ldn_conv_rows_split: shift_classes 16 needs pix_map and the layer geometry");
    LDN_REQUIRE(!chan_mask || rows_per_image > 0, "ldn_conv_rows_split: chan_mask needs rows_per_image");
    LDN_REQUIRE((uintptr_t)post_sub % 16 == 0 && (uintptr_t)chan_mask % 16 == 0, "ldn_conv_rows_split: post_sub / chan_mask must be 16-byte aligned");
    if (m_cap <= 0) return LDN_OK;
    DenseArgs d{a, lda, a_rows, m_count, m_cap, static_cast<const unsigned char*>(w_split), cin, cout, scale, shift, relu,
                relu_if_neg, out_rows, residual, ldr, out, ldo, taps, shift_classes, pix_map, Hi, Wi, Ho > 0 ? Ho : 1, Wo > 0 ? Wo : 1,
                stride, post_sub, chan_mask, rows_per_image > 0 ? rows_per_image : 1, 0, 0, ln_stats, ln_c1, pool, pool_gy, pool_gx, pool_Sx, nullptr, 1};
    hipStream_t st = static_cast<hipStream_t>(stream);
    // columns per workgroup: as wide as the layer allows (fewer passes over the activation rows) while the grid still fills the chip
    const int mt = ceil_div(m_cap, D_ROWS_MAX);
    if (f32) {   // true-fp32 MFMA: the same tile rules (the matrix time per tile is 5.3x longer, the staging the same)
        if (taps == 9) return (cout % 128 == 0 || cout > 64) ? launch_dense<4, true, true>(d, st) : launch_dense<2, true, true>(d, st);
        if (cout % 256 == 0 && (long)mt * (cout / 256) >= 384) return launch_dense<8, false, true>(d, st);
        if (cout % 128 == 0) return launch_dense<4, false, true>(d, st);
        if (cout <= 64) return launch_dense<2, false, true>(d, st);
        if (cout % 160 == 0 || (cout % 32 != 0 && cout > 128)) return launch_dense<5, false, true>(d, st);
        return launch_dense<4, false, true>(d, st);
    }
    // 128-row tiles (four waves, two workgroups per CU) where the grid of 256-row tiles is at most one round of the chip's 512
    // workgroup slots (the workgroups that survive the device-side row count sit alone on their CUs).  MEASURED SLOWER
    // (round 4: channel 12.07 -> 12.30 ms, layer 12.92 -> 13.11, spatial 13.98 -> 14.28: a third more L2 -> LDS bytes per product
    // outweighs the second resident workgroup), so it is off; LDN_DENSE_R128=1 (or a workgroup-count threshold) switches it on for A/B
    static const int r128 = getenv("LDN_DENSE_R128") ? atoi(getenv("LDN_DENSE_R128")) : 0;
    const bool small_grid = r128 && cout % 128 == 0 && (long)mt * (cout / 128) <= (r128 > 1 ? r128 : 512);
    // Tile width by a cost model when the number of rows is KNOWN (no device-side count) or HINTED (ldn_hint_rows: the host cannot see
    // a device-side count, the caller passes what the previous forward's count was): the time of a launch is rounds x tile time
    // (DESIGN.md 4n / 4t), rounds = ceil(live workgroups / 256) (one workgroup per CU: 115-159 KB of LDS), tile time = chunks x (1700 + 500
    // NSUB) + 12000 NSUB cycles (the per-chunk law of 4r; the second constant, fitted: epilogue + pipeline fill per 32-column subtile).  Results do not depend on the choice.
    static const bool use_model = !(getenv("LDN_DENSE_MODEL") && atoi(getenv("LDN_DENSE_MODEL")) == 0);
    // the ragged form of k_dense2 (any cin % 8 == 0, ragged last column tile; 64- and 160-column tiles: LAD-RegNet's widths)
    static const bool rag2 = !(getenv("LDN_DENSE_V2") && atoi(getenv("LDN_DENSE_V2")) == 0) && !getenv("LDN_DENSE_NO_RAG");
    const long rows_known = !m_count ? (long)m_cap : (hint >= 0 ? (hint < m_cap ? hint : (long)m_cap) : -1);
    static long mc[3] = {1700, 500, 12000};   // LDN_DENSE_MODEL_C="a,b,c" overrides (tuning)
    static const bool mc_env = [] { const char* e = getenv("LDN_DENSE_MODEL_C"); if (e) sscanf(e, "%ld,%ld,%ld", &mc[0], &mc[1], &mc[2]); return e != nullptr; }();
    (void)mc_env;
    auto tile_cost = [&](int ns) {
        const long mtl = (rows_known + D_ROWS_MAX - 1) / D_ROWS_MAX;
        const long chunks = (long)taps * ((cin + 31) / 32);
        const long wgs = mtl * (cout / (ns * 32));
        return (double)((wgs + 255) / 256) * (double)(chunks * (mc[0] + mc[1] * ns) + mc[2] * ns);
    };
    LDN_REQUIRE(!((ps || of || taps == 9) && relu == 3), "ldn_conv_rows_split: the GELU epilogue exists on the plain 1x1 form only");
    LDN_REQUIRE(!((ps || of) && (post_sub || chan_mask || ln_stats)), "ldn_conv_rows_ps: no post_sub / chan_mask / LayerNorm terms on the pre-split forms");
    if (ps || of) {      // pre-split rows on one or both sides: 1x1, whole column tiles of 256 / 128 / 64
        int best = 0;
        double best_cost = 0.0;
        if (use_model && rows_known > 0) {
            for (int ns : {8, 4, 2}) {
                if (cout % (ns * 32) != 0) continue;
                const double cost = tile_cost(ns);
                if (!best || cost < best_cost) { best = ns; best_cost = cost; }
            }
        }
        if (!best) best = (cout % 256 == 0 && (long)mt * (cout / 256) >= 384) ? 8 : (cout % 128 == 0 ? 4 : 2);
        const bool v2 = dense2_ok(best, 1, cin, cout);
        if (ps && !of) {
            switch (best) {
                case 8: return v2 ? launch_dense2<8, false, true, false>(d, st) : launch_dense_f<8, false, true, false, 256, true, false>(d, st);
                case 4: return v2 ? launch_dense2<4, false, true, false>(d, st) : launch_dense_f<4, false, true, false, 256, true, false>(d, st);
                default: return v2 ? launch_dense2<2, false, true, false>(d, st) : launch_dense_f<2, false, true, false, 256, true, false>(d, st);
            }
        }
        LDN_REQUIRE(!ps, "ldn_conv_rows_ps: pre-split on both sides is not built");
        switch (best) {
            case 8: return v2 ? launch_dense2<8, false, false, true>(d, st) : launch_dense_f<8, false, true, false, 256, false, true>(d, st);
            case 4: return v2 ? launch_dense2<4, false, false, true>(d, st) : launch_dense_f<4, false, true, false, 256, false, true>(d, st);
            default: return v2 ? launch_dense2<2, false, false, true>(d, st) : launch_dense_f<2, false, true, false, 256, false, true>(d, st);
        }
    }
    if (taps == 9) {
        if (small_grid) return launch_dense_f<4, true, true, false, 128>(d, st);
        if (use_model && rows_known > 0 && cout % 128 == 0 && tile_cost(2) < tile_cost(4))
            return dense2_ok(2, 9, cin, cout) ? launch_dense2<2, true, false, false>(d, st) : launch_dense_f<2, true, true>(d, st);
        if (cout % 128 == 0 && dense2_ok(4, 9, cin, cout)) return launch_dense2<4, true, false, false>(d, st);
        if (cout % 128 != 0 && cout <= 64 && dense2_ok(2, 9, cin, cout)) return launch_dense2<2, true, false, false>(d, st);
        return (cout % 128 == 0 || cout > 64) ? launch_dense<4, true>(d, st) : launch_dense<2, true>(d, st);
    }
    // ONE K chunk (a 32-wide input: LAD-RegNet's stage 1 behind its 32-channel stem): nothing to pipeline inside a workgroup -- load, multiply, store
    // follow one another -- so the launch lives on workgroups overlapping EACH OTHER: 128-row tiles with a two-slot ring (49 KB: three per CU)
    // instead of 256-row tiles with the 122 KB ring of the long-K form (one per CU).  LDN_DENSE_SHORTK=<widest cin> (0: off) for A/B.
    static const int shortk = getenv("LDN_DENSE_SHORTK") ? atoi(getenv("LDN_DENSE_SHORTK")) : 32;      // widest input that takes this form (0: off; A/B)
    if (cin <= shortk && !small_grid && !post_sub && !chan_mask && !ln_stats && relu != 3) {
        if (cout <= 64) return cout % 64 == 0 ? launch_dense_f<2, false, true, false, 128>(d, st) : launch_dense_f<2, false, false, false, 128>(d, st);
        if (cout % 128 == 0) return launch_dense_f<4, false, true, false, 128>(d, st);
    }
    static const int shortk5 = getenv("LDN_DENSE_SHORTK5") ? atoi(getenv("LDN_DENSE_SHORTK5")) : 144;  // the same for 160-column tiles (ragged widths above 128; RegNet 3.25 -> 3.22 ms); 0: off
    if (cin <= shortk5 && !small_grid && !post_sub && !chan_mask && !ln_stats && relu != 3 && (cout % 160 == 0 || (cout % 32 != 0 && cout > 128)))
        return cout % 160 == 0 ? launch_dense_f<5, false, true, false, 128>(d, st) : launch_dense_f<5, false, false, false, 128>(d, st);
    if (use_model && rows_known > 0 && !small_grid) {
        int best = 0;
        double best_cost = 0.0;
        for (int ns : {8, 6, 5, 4, 2}) {
            if (cout % (ns * 32) != 0) continue;
            const double cost = tile_cost(ns);
            if (!best || cost < best_cost) { best = ns; best_cost = cost; }
        }
        const bool v2 = best && dense2_ok(best, 1, cin, cout);
        switch (best) {
            case 8: return v2 ? launch_dense2<8, false, false, false>(d, st) : launch_dense_f<8, false, true>(d, st);
            case 6: return v2 ? launch_dense2<6, false, false, false>(d, st) : launch_dense_f<6, false, true>(d, st);
            case 5: return v2 ? launch_dense2<5, false, false, false>(d, st) : (rag2 && cin <= 2048 ? launch_dense2<5, false, false, false, true, true>(d, st) : launch_dense_f<5, false, true>(d, st));
            case 4: return v2 ? launch_dense2<4, false, false, false>(d, st) : launch_dense_f<4, false, true>(d, st);
            case 2: return v2 ? launch_dense2<2, false, false, false>(d, st) : (rag2 && cin <= 2048 && cin > 64 ? launch_dense2<2, false, false, false, true, true>(d, st) : launch_dense_f<2, false, true>(d, st));
            default: break;
        }
    }
    if (cout % 256 == 0 && (long)mt * (cout / 256) >= 384) return dense2_ok(8, 1, cin, cout) ? launch_dense2<8, false, false, false>(d, st) : launch_dense<8, false>(d, st);
    if (small_grid) return launch_dense_f<4, false, true, false, 128>(d, st);
    if (cout % 128 == 0) return dense2_ok(4, 1, cin, cout) ? launch_dense2<4, false, false, false>(d, st) : launch_dense<4, false>(d, st);
    if (cout <= 64) return dense2_ok(2, 1, cin, cout) ? launch_dense2<2, false, false, false>(d, st) : (rag2 && cin <= 2048 && cin > 64 ? launch_dense2<2, false, false, false, true, true>(d, st) : launch_dense<2, false>(d, st));
    // 160-column tiles: layers whose width is a multiple of 160 (320: two whole tiles instead of 128 + 128 + 64), and ragged widths
    // above 128 (144 in one tile; 784 = 4 x 160 + 144) -- LAD-RegNet
    static const bool use5 = !getenv("LDN_DENSE_NO5");
    if (use5 && (cout % 160 == 0 || (cout % 32 != 0 && cout > 128))) {
        if (rag2 && cin <= 2048) return launch_dense2<5, false, false, false, true, true>(d, st);      // (the ragged form carries every epilogue term)
        return launch_dense<5, false>(d, st);
    }
    if (cout % 32 != 0 && cout > 128 && cout <= 256) return launch_dense<8, false>(d, st);   // a ragged layer in ONE column tile (144, 168, 216 ...)
    return launch_dense<4, false>(d, st);
}

extern "C" int ldn_row_stats(const float* x, int ld, int rows, int C, float eps, float* stats, void* stream) {
    LDN_REQUIRE(x && stats, "ldn_row_stats: null pointer");
    LDN_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= 2048 && ld >= C && ld % 4 == 0, "ldn_row_stats: C must be a multiple of 4, at most 2048 (got %d), ld >= C", C);
    LDN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)stats % 8 == 0, "ldn_row_stats: pointers must be 16 / 8-byte aligned");
    if (rows == 0) return LDN_OK;
    hipLaunchKernelGGL(ldn::k_row_stats, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, rows, C, eps, stats,
                       nullptr, nullptr);
    LDN_CHECK_LAUNCH("k_row_stats");
    return LDN_OK;
}

extern "C" int ldn_row_stats_list(const float* x, int ld, int rows, int C, float eps, const int32_t* list, const int32_t* count, float* stats,
                                  void* stream) {
    LDN_REQUIRE(x && stats && list && count, "ldn_row_stats_list: null pointer");
    LDN_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= 2048 && ld >= C && ld % 4 == 0, "ldn_row_stats_list: C must be a multiple of 4, at most 2048 (got %d), ld >= C", C);
    LDN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)stats % 8 == 0, "ldn_row_stats_list: pointers must be 16 / 8-byte aligned");
    if (rows == 0) return LDN_OK;
    hipLaunchKernelGGL(ldn::k_row_stats, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, rows, C, eps, stats,
                       list, count);
    LDN_CHECK_LAUNCH("k_row_stats");
    return LDN_OK;
}
