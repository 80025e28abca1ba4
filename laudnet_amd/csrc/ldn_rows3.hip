// k_rows3 -- the packed 3x3 convolution of the spatial / layer path on PRE-SPLIT rows (gfx950, bf16x3; round 5):
//     h2[m, n] = act(scale[n] * sum_{tap, k} h1[nbr[m][tap], k] * w2[n, tap, k] + shift[n])          (laud_resnet.py:119-126 on the
// active pixels: conv2 -> bn2 -> ReLU; the operator the reference's simulator calls gather_conv2, eval_example.py:39-48).
// Same arithmetic as k_dense<.., T9> (csrc/ldn_dense.hip) -- the same products in the same order, bit-identical results -- restructured
// around what bounded that kernel: a barrier-to-barrier chain of ~1 700 cycles per 32-wide K chunk of which a third was
// the in-loop bf16 split of the activation rows.  Here
//   * NO operand is split in the K loop: h1 arrives pre-split from conv1's epilogue (k_dense<.., OF>: [row][cin / 8][8 hi | 8 lo] bf16, the
//     weights' own layout, 4 bytes per element) -- a B fragment is two ds_read_b128, as an A fragment is;
//   * the workgroup barrier is paid once per 64-wide K step: only the WEIGHT tile is shared between waves (ring of two K64 slots, one
//     s_barrier per slot); a wave's 32 activation rows are staged by that wave ALONE into a private double buffer of 32-wide sub-chunks and
//     ordered by its own vmcnt -- no barrier, and 64 KB of LDS instead of the 96 KB a three-deep common ring of K32 slots took;
//   * B fragments are double-buffered in registers: the rows of sub-chunk s + 1 are read during the MFMAs of sub-chunk s, the LDS-DMA of
//     sub-chunk s + 3 reuses their slot right behind -- every DMA / LDS instruction of a wave sits between its MFMAs (one per K16 step).
// One workgroup = 256 packed rows (8 waves x 32 rows, lane = row: transposed formulation, A operand = weights) x NT = 32 NSUB columns.
// The output rows are fp32 or pre-split (OF: conv3 -- k_dense<.., PS> -- then splits nothing either).
#include "ldn_common.h"

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rows3Args {
    const unsigned char* a; long lda;                 // pre-split rows, lda in BYTES (>= 4 cin)
    const int32_t* nbr;                               // [m_cap][9] packed h1 row of every tap, -1 = zero row
    const int32_t* m_count; int m_cap;
    const unsigned char* ws;                          // [cout][9 cin / 8][32 B] pre-split, tap-major K
    int cin, cout;
    const float* scale; const float* shift; int relu;
    unsigned char* out; long ldo;                     // fp32 or pre-split rows, ldo in BYTES
    int mtn, ntn;
};

constexpr int ROWS3_MAX_CIN = 2048;      // widest input the zero row covers (ldn_conv3x3_rows_ps requires cin <= ROWS3_MAX_CIN; ops.rows_ps_ok mirrors it)
__device__ __attribute__((aligned(16))) float g_rows3_zero[ROWS3_MAX_CIN] = {0.f};      // a whole zero ROW: a missing neighbour is a row base like any other
#ifdef LDN_TRACE   // tuning only: per-wave cycle split of the K loop (tools/trace_rows3.py)
__device__ unsigned long long* g_rows3_trace = nullptr;
#define RT(x) x = __builtin_amdgcn_s_memtime();
#else
#define RT(x)
#endif

__device__ __forceinline__ void r3_dma16(const void* gsrc, unsigned lds_base) {
#ifdef R3_M0_CLOBBER     // tuning: M0 declared clobbered instead of saved / restored around every DMA (two scalar instructions fewer)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_base) : "memory", "m0");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void r3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void r3_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void r3_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned r3_lds_off(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}

constexpr int R3_ROWS = 256;

template <int NSUB, bool OF>
__global__ __launch_bounds__(512, 2) void k_rows3(const Rows3Args p) {
    constexpr int NT = NSUB * 32;
    constexpr int WSLOT = NT * 256;                   // weight rows of one K64 step: [NT][8 octets][32 B], 16-byte units XOR-swizzled by (row & 15)
    constexpr int RSLOT = 32 * 128;                   // one wave's 32 rows of one K32 sub-chunk: units XOR-swizzled by ((row >> 1) & 7)
    constexpr int NWI = NT / 32;                      // weight DMA instructions (1 KB = 4 rows each) per wave and K64 step
    static_assert(NWI >= 1 && NWI <= 4, "tile width");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* const s_atap = reinterpret_cast<int*>(smem);                       // [256][9] source row per tap, -1 = zero row
    unsigned char* const s_w = smem + R3_ROWS * 9 * 4;                      // 2 weight slots
    unsigned char* const s_r = s_w + 2 * WSLOT;                             // 8 waves x 2 row slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware order (block b runs on XCD b % 8): the N tiles of one M tile on one XCD -- its h1 rows come from that L2 for the other N tiles
    const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
    const int nt = slot_i % p.ntn, mt = (slot_i / p.ntn) * 8 + xcd;
    if (mt >= p.mtn) return;
    const int M = p.m_count ? min(p.m_count[0], p.m_cap) : p.m_cap;
    const int m0 = mt * R3_ROWS;
    if (m0 >= M) return;
    const int rows = min(R3_ROWS, M - m0);
    const int n0 = nt * NT;

    {   // the tile's neighbour table: all five entries of a thread requested before the first is stored (round 6: as a plain loop hipcc emits
        // load -> s_waitcnt vmcnt(0) -> ds_write per entry, five memory latencies in a row in front of the first DMA of every workgroup)
        constexpr int NE = (R3_ROWS * 9 + 511) / 512;
        int v[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int i = tid + 512 * u, r = i / 9;
            v[u] = (i < R3_ROWS * 9 && r < rows) ? p.nbr[(size_t)(m0 + r) * 9 + (i - r * 9)] : -1;
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int i = tid + 512 * u;
            LDN_DCHECK(v[u] >= -1, 601);
            if (i < R3_ROWS * 9) s_atap[i] = v[u];
        }
    }
    __syncthreads();

    const bool active = wave * 32 < rows;
    const int cpt = p.cin >> 5;                       // K32 sub-chunks per tap
    const int nsub = 9 * cpt, nsup = nsub >> 1;       // (cin % 64 == 0: a K64 step never straddles two taps' weight rows)
    const long wrow = (long)9 * p.cin * 4;            // bytes per weight row
    const unsigned lds_w = r3_lds_off(s_w), lds_r = r3_lds_off(s_r) + (unsigned)wave * 2u * RSLOT;
    unsigned char* const my_r = s_r + wave * 2 * RSLOT;

    // per-lane sources.  Weights: instruction k (0 .. NWI - 1) of a K64 step covers LDS rows 32 k + 4 wave .. + 3, lane = (row in quad, 16-byte
    // unit); its source advances by 256 B per step (steps past the end re-read the last one: harmless, keeps the vmcnt arithmetic constant).
    const unsigned char* wsrc[NWI];
#pragma unroll
    for (int k = 0; k < NWI; ++k) {
        const int rr = 32 * k + 4 * wave + (lane >> 4);
        wsrc[k] = p.ws + (long)(n0 + rr) * wrow + (((lane & 15) ^ (rr & 15)) << 4);
    }
    auto dma_w = [&](int S, int k) {
        r3_dma16(wsrc[k] + (long)min(S, nsup - 1) * 256,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_w + (unsigned)(S & 1) * WSLOT + (unsigned)(32 * k + 4 * wave) * 256u)));
    };
    // Rows: instruction k (0 .. 3) of a K32 sub-chunk covers the wave's rows 8 k .. 8 k + 7, lane = (row in octet, unit).  rsrc[k] = this lane's
    // row of the CURRENT tap (+ its swizzled unit); a missing neighbour (-1) is the zero row.  (tap_r, ck_r) = position of the next sub-chunk to
    // be issued, advanced by next_r(); the table is re-read when the tap changes (once per cin / 32 sub-chunks).
    const unsigned char* rsrc[4];
    int tap_r = 0, ck_r = 0;
    auto load_tap = [&](int tap) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * k + (lane >> 3);
            const int ar = s_atap[(wave * 32 + r) * 9 + tap];
            const unsigned char* base = ar >= 0 ? p.a + (long)ar * p.lda : reinterpret_cast<const unsigned char*>(g_rows3_zero);
            rsrc[k] = base + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
    };
    auto dma_r = [&](int s, int k) {      // (s only selects the slot: the source position is (tap_r, ck_r))
        r3_dma16(rsrc[k] + ck_r * 128, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_r + (unsigned)(s & 1) * RSLOT + (unsigned)k * 1024u)));
    };
    auto next_r = [&]() {
        if (++ck_r == cpt) {
            ck_r = 0;
            tap_r = min(tap_r + 1, 8);      // (sub-chunks past the end re-read tap 8)
            load_tap(tap_r);
        }
    };

    if (!active) {      // a wave without rows (ragged last tile) only stages its share of the weights
        for (int k = 0; k < NWI; ++k) dma_w(0, k);
        for (int S = 0; S < nsup; ++S) {
            r3_wait_vm<0>();
            r3_barrier();
            for (int k = 0; k < NWI; ++k) dma_w(S + 1, k);
        }
        r3_wait_vm<0>();
        return;
    }

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // B fragments (the wave's 32 rows, lane = row l31, k-octet 2 half + h of the K32 sub-chunk): [set = sub-chunk parity][half]
    bf16x8 bh[2][2], bl[2][2];
    const unsigned xsw = ((unsigned)l31 >> 1) & 7u, wsw = (unsigned)l31 & 15u;
    auto load_b = [&](int s, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
        const unsigned char* xs = my_r + (s & 1) * RSLOT + l31 * 128;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned sl = 4u * half + 2u * h;
            dh[half] = *reinterpret_cast<const bf16x8*>(xs + ((sl ^ xsw) << 4));
            dl[half] = *reinterpret_cast<const bf16x8*>(xs + (((sl + 1) ^ xsw) << 4));
        }
    };

#ifdef R3_PRIO            // tuning: static priority for the second-dispatched half (the arbitration loser on every segment, MI355X_MICROARCH.md)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
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
    r3_wait_vm<4>();
    load_b(0, bh[0], bl[0]);
    r3_wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_r(2, k);
    next_r();

#ifdef LDN_TRACE
    unsigned long long t0, t1, t2, t3, a_wait = 0, a_bar = 0, a_body = 0, t_start, t_loop;
    RT(t_start)
#endif
    // Issue order of a wave (in-order completion: the vmcnt waits below count what may still be outstanding):
    //   step S:  [wait W(S): vmcnt(8)] barrier | W(S+1) | sub 2S: [wait R(2S+1): vmcnt(4 + NWI)] read B(2S+1), R(2S+3) |
    //                                                     sub 2S+1: [wait R(2S+2): vmcnt(4 + NWI)] read B(2S+2), R(2S+4)
    for (int S = 0; S < nsup; ++S) {
        RT(t0)
        r3_wait_vm<8>();
        RT(t1)
        r3_barrier();
        RT(t2)
        const unsigned char* wsl = s_w + (S & 1) * WSLOT + l31 * 256;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int s = 2 * S + sub;
            // weight fragment of MFMA step st = half * NSUB + j: row 32 j + l31, octet 4 sub + 2 half + h of the K64 step
            auto frag = [&](int st, bf16x8& ah, bf16x8& al) {
                const int half = st / NSUB, j = st - half * NSUB;
                const unsigned uu = 2u * (4u * sub + 2u * half + h);
                ah = *reinterpret_cast<const bf16x8*>(wsl + j * (32 * 256) + ((uu ^ wsw) << 4));
                al = *reinterpret_cast<const bf16x8*>(wsl + j * (32 * 256) + (((uu + 1) ^ wsw) << 4));
            };
            bf16x8 ah[2], al[2];
            frag(0, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
            // where the other instructions of the sub-chunk sit between its 2 NSUB MFMA steps (one per step):
            //   sub 0: steps 0 .. NWI-1: W(S+1);  step NWI: wait + read B(s+1);  then R(s+3), one instruction per step (the rest behind the last step)
            //   sub 1: step 0: wait + read B(s+1);  then R(s+3)
            constexpr int NST = 2 * NSUB;
            const int rd_at = sub == 0 ? (NWI < NST - 1 ? NWI : NST - 2) : 0;
            int r_done = 0, w_done = 0;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) frag(st + 1, ah[(st + 1) & 1], al[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int half = st / NSUB, j = st - half * NSUB;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[st & 1], bh[sub][half], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st & 1], bl[sub][half], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st & 1], bh[sub][half], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (sub == 0 && st < rd_at) {
                    if (w_done < NWI) dma_w(S + 1, w_done++);
                } else if (st == rd_at) {
                    if (sub == 0)
                        while (w_done < NWI) dma_w(S + 1, w_done++);       // (narrow tiles: fewer steps than instructions)
                    r3_wait_vm<4 + NWI>();
                    load_b(s + 1, bh[sub ^ 1], bl[sub ^ 1]);
                } else {
                    if (r_done == 0) r3_wait_lgkm0();                      // B(s+1) is in registers: its slot may be refilled
                    if (r_done < 4) dma_r(s + 3, r_done++);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (r_done == 0) r3_wait_lgkm0();
            while (r_done < 4) dma_r(s + 3, r_done++);
            next_r();
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef LDN_TRACE
        asm volatile("" : "+v"(acc[0]));
        RT(t3)
        a_wait += t1 - t0; a_bar += t2 - t1; a_body += t3 - t2;
#endif
    }
#ifdef LDN_TRACE
    RT(t_loop)
#endif
    r3_wait_vm<0>();        // the trailing dummy DMAs have landed: this wave's row slots become its 32 x 32 transpose scratch (private: no barrier)
    r3_wait_lgkm0();

    // ---- epilogue: per n-subtile, C layout (lane = row, register = channel) -> rows of 32 channels, 16-byte accesses
    float* const scr = reinterpret_cast<float*>(my_r);
    const int trw = lane >> 3, tc = lane & 7;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const int cb = n0 + 32 * j + tc * 4;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + cb);
        const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cb) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = {acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
            *reinterpret_cast<f32x4*>(scr + l31 * 32 + (((2 * q4 + h) ^ (l31 & 7)) << 2)) = v;
        }
        r3_wait_lgkm0();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = trw + 8 * it;
            f32x4 x = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tc ^ (row & 7)) << 2));
            x = x * sc + sh;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
            }
            const bool ok = wave * 32 + row < rows;
            unsigned char* orow = p.out + (size_t)(m0 + wave * 32 + row) * p.ldo;
            if constexpr (OF) {     // pre-split rows: see k_dense's OF epilogue (csrc/ldn_dense.hip) -- same pairing, same conversions
                const bool odd = tc & 1;
                const u32x4_t o = presplit_store_quad(x, odd);
                if (ok) store16(orow + (size_t)(n0 + 32 * j + (tc & ~1) * 4) * 4 + (odd ? 16 : 0), o);
            } else {
                if (ok) store16(orow + (size_t)cb * 4, x);
            }
        }
        r3_wait_lgkm0();
        __builtin_amdgcn_wave_barrier();
    }
#ifdef LDN_TRACE
    if (g_rows3_trace && lane == 0) {
        unsigned long long t_end;
        RT(t_end)
        unsigned long long* r = g_rows3_trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        r[0] = a_wait; r[1] = a_bar; r[2] = a_body; r[3] = 0; r[4] = 0; r[5] = t_loop - t_start; r[6] = t_end - t_loop; r[7] = nsup;
    }
#endif
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_rows3)

template <int NSUB, bool OF>
static int launch_rows3(Rows3Args& a, hipStream_t st) {
    constexpr int NT = NSUB * 32;
    const size_t lds = (size_t)R3_ROWS * 9 * 4 + 2 * (size_t)NT * 256 + 8 * 2 * (size_t)32 * 128;
    a.ntn = a.cout / NT;
    a.mtn = ceil_div(a.m_cap, R3_ROWS);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_rows3<NSUB, OF>), lds), "k_rows3: cannot reserve %zu B of LDS", lds);
    const unsigned grid = (unsigned)round_up(a.mtn, 8) * a.ntn;
    hipLaunchKernelGGL((k_rows3<NSUB, OF>), dim3(grid), dim3(512), lds, st, a);
    LDN_CHECK_LAUNCH("k_rows3");
    return LDN_OK;
}

}  // namespace ldn

using namespace ldn;

#ifdef LDN_TRACE
extern "C" int ldn_debug_set_rows3_trace(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rows3_trace), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif

// The packed 3x3 of the spatial / layer path on pre-split rows (see the header of this file and include/ldn_hip.h).
extern "C" int ldn_conv3x3_rows_ps(const void* a_presplit, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap, const void* w_split,
                                   int cin, int cout, const float* scale, const float* shift, int relu, void* out, int ldo, int out_presplit,
                                   int rows_hint, void* stream) {
    LDN_REQUIRE(a_presplit && nbr && w_split && shift && out, "ldn_conv3x3_rows_ps: null pointer");
    LDN_REQUIRE(cin > 0 && cin % 64 == 0 && cout > 0 && cout % 64 == 0, "ldn_conv3x3_rows_ps: cin and cout must be multiples of 64 (got %d, %d)", cin, cout);
    LDN_REQUIRE(cin <= ldn::ROWS3_MAX_CIN, "ldn_conv3x3_rows_ps: cin must be <= %d (a missing neighbour is sourced from a zero row of that width; got %d -- use ldn_conv_rows_split(taps = 9))", ldn::ROWS3_MAX_CIN, cin);
    LDN_REQUIRE(lda >= cin && lda % 4 == 0 && ldo >= cout && ldo % 4 == 0, "ldn_conv3x3_rows_ps: strides (in 4-byte elements) must be multiples of 4 and cover the row");
    LDN_REQUIRE((uintptr_t)a_presplit % 16 == 0 && (uintptr_t)w_split % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)shift % 16 == 0 &&
                (uintptr_t)scale % 16 == 0, "ldn_conv3x3_rows_ps: pointers must be 16-byte aligned");
    LDN_REQUIRE(relu == 0 || relu == 1, "ldn_conv3x3_rows_ps: relu must be 0 or 1");
    if (m_cap <= 0) return LDN_OK;
    Rows3Args d{static_cast<const unsigned char*>(a_presplit), (long)lda * 4, nbr, m_count, m_cap, static_cast<const unsigned char*>(w_split), cin, cout,
                scale, shift, relu, static_cast<unsigned char*>(out), (long)ldo * 4, 0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    // tile width: 128 columns where the layer has them; 64-column tiles when the (hinted or known) row count leaves the grid of 128-column
    // tiles under half a round of the chip's 256 CUs and the narrower tiles fill more of it (the rule of k_dense's cost model)
    const long rows = !m_count ? (long)m_cap : (rows_hint >= 0 ? (rows_hint < m_cap ? rows_hint : m_cap) : -1);
    bool narrow = cout % 128 != 0;
    if (!narrow && rows > 0) {
        const long wg128 = ((rows + R3_ROWS - 1) / R3_ROWS) * (cout / 128);
        narrow = wg128 <= 128;
    }
    if (narrow) return out_presplit ? launch_rows3<2, true>(d, st) : launch_rows3<2, false>(d, st);
    return out_presplit ? launch_rows3<4, true>(d, st) : launch_rows3<4, false>(d, st);
}
