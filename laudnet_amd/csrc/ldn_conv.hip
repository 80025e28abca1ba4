// fp32-MFMA convolution kernels of the LAUDNet hot path (gfx950 / CDNA4).
//
//   k_conv_rows  : spatial / layer mode.  GEMM over PACKED ACTIVE ROWS (pixels) with weights shared by
//                  the whole batch; A rows are gathered on load through an index list (1x1) or a
//                  9-neighbour table (3x3); the epilogue applies the folded BN (+ReLU) and either writes
//                  packed rows or scatter-adds into the NHWC residual stream.
//   (channel mode lives in ldn_conv_image.hip)
//
// Main loop: 256 threads = 4 wave64; block tile BM x BN, K chunk 32; A and B tiles are
// K-contiguous in memory (NHWC pixels rows, [cout][tap][cin] weights) so both are staged with 16-byte
// loads into LDS rows padded to 36 floats (144 B: ds_write_b128 and ds_read_b128 are conflict-free), double
// buffered with one barrier per chunk; v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD) accumulates
// TM x TN 32x32 tiles per wave.  Each ds_read_b128 feeds four MFMAs: lane (i, h) holds k = 8*kk + 4*h + q.
#include "ldn_common.h"

namespace ldn {

constexpr int BK = 32;
constexpr int LDT = BK + 4;

template <int BM_, int BN_, int WGM_, int WGN_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    static constexpr int WM = BM / WGM, WN = BN / WGN;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be 32x32 multiples");
    static constexpr int AI = BM / 32, BI = BN / 32;
    static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDT;
};

template <class C>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[C::TM][C::TN], int a_row0, int b_row0, int l31, int h) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
        f32x4 af[C::TM], bf[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
            af[i] = *reinterpret_cast<const f32x4*>(As + (a_row0 + i * 32 + l31) * LDT + kk * 8 + h * 4);
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
            bf[j] = *reinterpret_cast<const f32x4*>(Bs + (b_row0 + j * 32 + l31) * LDT + kk * 8 + h * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
    }
}

// =====================================================================================================
// rows family
// =====================================================================================================
struct RowsArgs {
    const float* a; int lda;
    const int32_t* a_rows; int taps;
    const int32_t* m_count; int m_cap;
    const float* w; int cin; int cout;
    const float* scale; const float* shift; int relu;
    const int32_t* relu_if_neg; const int32_t* out_rows;
    const float* residual; int ldr;
    float* out; int ldo;
    int ntn;
};

template <class C>
__global__ __launch_bounds__(256) void k_conv_rows(const RowsArgs p) {
    constexpr int BM = C::BM, BN = C::BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDT;
    int* s_rows = reinterpret_cast<int*>(smem + C::LDS_FLOATS);  // [BM][taps]
    int* s_orow = s_rows + BM * p.taps;                           // [BM]

    const int tid = threadIdx.x;
    const int T = p.taps;
    int M = p.m_cap;
    if (p.m_count) M = min(__builtin_nontemporal_load(p.m_count), p.m_cap);
    // XCD-aware tile order: block b runs on XCD b%8; all N tiles of one M tile stay on one XCD (its L2
    // keeps the gathered A rows), consecutive M tiles spread over the 8 XCDs.
    const int bid = blockIdx.x;
    const int y = bid >> 3;
    const int nt = y % p.ntn;
    const int mt = (y / p.ntn) * kXcds + (bid & 7);
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 >= M) return;

    for (int i = tid; i < BM * T; i += 256) {
        const int r = i / T, m = m0 + r;
        s_rows[i] = m < M ? (p.a_rows ? p.a_rows[(size_t)m * T + (i - r * T)] : m) : -1;
    }
    for (int i = tid; i < BM; i += 256) {
        const int m = m0 + i;
        s_orow[i] = m < M ? (p.out_rows ? p.out_rows[m] : m) : -1;
    }
    __syncthreads();

    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    f32x4 ra[C::AI], rb[C::BI];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int tap, int c0) {
        const int c = c0 + lc4;
        const bool cok = c < p.cin;
#pragma unroll
        for (int u = 0; u < C::AI; ++u) {
            const int ridx = s_rows[(lrow + 32 * u) * T + tap];
            ra[u] = (cok && ridx >= 0) ? *reinterpret_cast<const f32x4*>(p.a + (size_t)ridx * p.lda + c) : zero4;
        }
#pragma unroll
        for (int u = 0; u < C::BI; ++u) {
            const int n = n0 + lrow + 32 * u;
            rb[u] = (cok && n < p.cout) ? *reinterpret_cast<const f32x4*>(p.w + ((size_t)n * T + tap) * p.cin + c)
                                        : zero4;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < C::AI; ++u)
            *reinterpret_cast<f32x4*>(As + buf * BM * LDT + (lrow + 32 * u) * LDT + lc4) = ra[u];
#pragma unroll
        for (int u = 0; u < C::BI; ++u)
            *reinterpret_cast<f32x4*>(Bs + buf * BN * LDT + (lrow + 32 * u) * LDT + lc4) = rb[u];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int cpt = ceil_div(p.cin, BK);
    const int nch = T * cpt;
    int tap = 0, c0 = 0;
    auto advance = [&]() {
        c0 += BK;
        if (c0 >= p.cin) { c0 = 0; ++tap; }
    };
    gload(0, 0);
    lstore(0);
    advance();
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) gload(tap, c0);
        mma_chunk<C>(As + (ch & 1) * BM * LDT, Bs + (ch & 1) * BN * LDT, acc, wm * C::WM, wn * C::WN, l31, h);
        if (more) {
            lstore((ch + 1) & 1);
            advance();
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
        const int n = n0 + wn * C::WN + j * 32 + l31;
        const bool nok = n < p.cout;
        const float sc = nok ? p.scale[n] : 0.f, sh = nok ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * C::WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int orow = s_orow[row];
                if (orow >= 0 && nok) {
                    float v = acc[i][j][r] * sc + sh;
                    if (p.residual) v += p.residual[(size_t)orow * p.ldr + n];
                    const bool do_relu = p.relu == 1 || (p.relu == 2 && p.relu_if_neg[m0 + row] < 0);
                    if (do_relu) v = fmaxf(v, 0.f);
                    p.out[(size_t)orow * p.ldo + n] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// host side
// =====================================================================================================
template <class C>
static int launch_rows(const RowsArgs& a, hipStream_t st) {
    const size_t lds = C::LDS_FLOATS * sizeof(float) + (size_t)C::BM * (a.taps + 1) * sizeof(int);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_conv_rows<C>), lds), "k_conv_rows: cannot reserve %zu B of LDS", lds);
    RowsArgs p = a;
    p.ntn = ceil_div(a.cout, C::BN);
    const int mt8 = round_up(ceil_div(a.m_cap, C::BM), kXcds);
    const unsigned grid = (unsigned)mt8 * p.ntn;
    hipLaunchKernelGGL(k_conv_rows<C>, dim3(grid), dim3(256), lds, st, p);
    LDN_CHECK_LAUNCH("k_conv_rows");
    return LDN_OK;
}

}  // namespace ldn

using namespace ldn;

extern "C" int ldn_conv_rows(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count,
                             int m_cap, const float* w, int cin, int cout, const float* scale, const float* shift,
                             int relu, const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual,
                             int ldr, float* out, int ldo, void* stream) {
    LDN_REQUIRE(a && w && scale && shift && out, "ldn_conv_rows: null pointer");
    LDN_REQUIRE(taps == 1 || taps == 9, "ldn_conv_rows: taps must be 1 or 9 (got %d)", taps);
    LDN_REQUIRE(a_rows || taps == 1, "ldn_conv_rows: a_rows required when taps > 1");
    LDN_REQUIRE(cin > 0 && cout > 0 && cin % 4 == 0, "ldn_conv_rows: cin must be a positive multiple of 4 (got %d)", cin);
    LDN_REQUIRE(lda % 4 == 0 && lda >= cin, "ldn_conv_rows: lda must be a multiple of 4 and >= cin");
    LDN_REQUIRE(ldo >= cout, "ldn_conv_rows: ldo < cout");
    LDN_REQUIRE(relu >= 0 && relu <= 2 && (relu != 2 || relu_if_neg), "ldn_conv_rows: bad relu mode");
    LDN_REQUIRE(!residual || ldr >= cout, "ldn_conv_rows: ldr < cout");
    LDN_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), "ldn_conv_rows: a/w must be 16-byte aligned");
    if (m_cap <= 0) return LDN_OK;
    RowsArgs p{a, lda, a_rows, taps, m_count, m_cap, w, cin, cout, scale, shift, relu,
               relu_if_neg, out_rows, residual, ldr, out, ldo, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cout > 64) return launch_rows<TileCfg<128, 128, 2, 2>>(p, st);
    if (cout > 32) return launch_rows<TileCfg<128, 64, 2, 2>>(p, st);
    return launch_rows<TileCfg<128, 32, 4, 1>>(p, st);
}
