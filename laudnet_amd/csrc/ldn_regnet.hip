// LAD-RegNet-specific kernels of the hot path (gfx950): grouped 3x3 convolution over packed pixel lists and the
// squeeze-excitation of a packed (layer-skip) batch.  All are HBM-bound VALU kernels: a RegNet-Y grouped conv has
// 9*gw MACs per output element (gw = 8..24), far below the fp32 ridge, so no MFMA here.
#include "ldn_common.h"

namespace ldn {

// out[r, c] = act(scale[c] * sum_{t<9} sum_{i<gw} a[nbr[r,t], g*gw + i] * w[c, t, i] + shift[c]),  g = c / gw
// one thread per (row, 4 consecutive output channels); the 4 channels share the group's activation loads.
template <int GW4>   // gw / 4 (2, 4 or 6): compile-time inner extent; 0 = run-time loop
__global__ __launch_bounds__(256) void k_grouped3x3_rows(const float* __restrict__ a, int lda,
                                                          const int32_t* __restrict__ nbr,
                                                          const int32_t* __restrict__ m_count, int m_cap,
                                                          const float* __restrict__ w, int C, int gw,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int relu, float* __restrict__ out, int ldo) {
    const int M = m_count ? min(m_count[0], m_cap) : m_cap;
    const int C4 = C >> 2;
    const long total = (long)M * C4;
    const int g4 = GW4 ? GW4 : (gw >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C4), c = (int)(i - (long)r * C4) * 4;
        const int g0 = (c / gw) * gw;   // first input channel of this output channel's group (c..c+3 share it: gw % 4 == 0)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ar = nbr[(size_t)r * 9 + t];
            if (ar < 0) continue;
            const float* ap = a + (size_t)ar * lda + g0;
            const float* wp = w + ((size_t)c * 9 + t) * gw;
            const int qn = GW4 ? GW4 : g4;
#pragma unroll
            for (int q = 0; q < qn; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)e * 9 * gw + q * 4);
                    acc[e] += av[0] * wv[0] + av[1] * wv[1] + av[2] * wv[2] + av[3] * wv[3];
                }
            }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 v = acc * sc + sh;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + (size_t)r * ldo + c) = v;
    }
}

// v2: one block = one channel group x a run of packed rows; the group's [gw][9][gw] weights sit in LDS (all lanes of a
// channel quad read the same address: broadcast), each thread produces 4 output channels of one row.
template <int GW>
__global__ __launch_bounds__(256) void k_grouped3x3_lds(const float* __restrict__ a, int lda,
                                                         const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ m_count, int m_cap,
                                                         const float* __restrict__ w, int C,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int relu, float* __restrict__ out, int ldo) {
    constexpr int QUADS = GW / 4, ROWS = 256 / QUADS;
    __shared__ __attribute__((aligned(16))) float s_w[GW * 9 * GW];
    const int g = blockIdx.y, tid = threadIdx.x;
    const int M = m_count ? min(m_count[0], m_cap) : m_cap;
    for (int i = tid * 4; i < GW * 9 * GW; i += 1024)
        *reinterpret_cast<f32x4*>(s_w + i) = *reinterpret_cast<const f32x4*>(w + (size_t)g * GW * 9 * GW + i);
    __syncthreads();
    const int q = tid % QUADS, rl = tid / QUADS;
    if (rl >= ROWS) return;
    const int c = g * GW + q * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    for (int row = blockIdx.x * ROWS + rl; row < M; row += gridDim.x * ROWS) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ar = nbr[(size_t)row * 9 + t];
            if (ar < 0) continue;      // (the branch also keeps hipcc from hoisting all 144 weight reads of a row: 512 VGPRs and spills, measured 7x slower)
            const float* ap = a + (size_t)ar * lda + g * GW;
#pragma unroll
            for (int qi = 0; qi < QUADS; ++qi) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + qi * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(s_w + ((q * 4 + e) * 9 + t) * GW + qi * 4);
                    acc[e] += av[0] * wv[0] + av[1] * wv[1] + av[2] * wv[2] + av[3] * wv[3];
                }
            }
        }
        f32x4 v = acc * sc + sh;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + (size_t)row * ldo + c) = v;
    }
}

// per-image channel sums over the image's packed rows [prefix[b], prefix[b+1]); grid (splits, B), deterministic
__global__ __launch_bounds__(256) void k_rows_gap(const float* __restrict__ a, int lda, const int32_t* __restrict__ prefix,
                                                   int C, int splits, float* __restrict__ partial) {
    // thread = (row lane, channel quad): with few channels (a 64-wide layer has 16 quads) the other threads of the workgroup take
    // interleaved rows; eight rows in flight per thread; fixed order of additions (row lanes reduced through LDS in lane order)
    __shared__ f32x4 s_red[256];
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int r0 = prefix[b], n = prefix[b + 1] - r0;
    const int per = ceil_div(max(n, 1), splits);
    const int lo = r0 + s * per, hi = min(r0 + n, lo + per);
    const int Q = C >> 2;
    const int RL = Q >= 256 ? 1 : 256 / Q;
    const int rl = Q >= 256 ? 0 : tid / Q;
    for (int q0 = 0; q0 < Q; q0 += 256) {
        const int q = q0 + (Q >= 256 ? tid : tid % Q);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (q < Q && rl < RL) {
            const float* src = a + q * 4;
            int r = lo + rl;
            for (; r + 7 * RL < hi; r += 8 * RL) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(r + k * RL) * lda);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
            for (; r < hi; r += RL) acc += *reinterpret_cast<const f32x4*>(src + (size_t)r * lda);
        }
        if (RL > 1) {
            __syncthreads();
            s_red[tid] = acc;
            __syncthreads();
            if (rl == 0 && q < Q)
                for (int k = 1; k < RL; ++k) acc += s_red[k * Q + q];
        }
        if (q < Q && rl == 0) *reinterpret_cast<f32x4*>(partial + ((size_t)b * splits + s) * C + q * 4) = acc;
    }
}

// squeeze-excitation head per image: mean -> fc1 + ReLU -> fc2 -> sigmoid  (torchvision SqueezeExcitation)
// With a channel list (channel mode: column j of image b is channel ch_idx[b, j], j < ch_cnt[b]) the weights are gathered
// through it; masked channels are exact zeros in the reference (post-activation mask) and contribute nothing to fc1.
template <int NT>   // threads per image: the three phases are chains of dependent global reads, so the more waves share them the fewer round trips
__global__ __launch_bounds__(NT) void k_se_head(const float* __restrict__ partial, const int32_t* __restrict__ prefix,
                                                  int C, int S, int splits, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, const int32_t* __restrict__ ch_idx,
                                                  const int32_t* __restrict__ ch_cnt, float* __restrict__ gate,
                                                  int slot_rows, const int32_t* __restrict__ m_count) {
    extern __shared__ __attribute__((aligned(16))) float s_f[];
    float* s_mean = s_f;        // [C]
    float* s_hid = s_f + C;     // [S]
    int* s_ch = reinterpret_cast<int*>(s_f + C + S);   // [C] channel of column j
    constexpr int NW = NT / 64;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // slot form (slot_rows > 0): b is the b-th KEPT image, whole images of slot_rows rows each, *m_count rows in all
    const int n = slot_rows > 0 ? ((long)b * slot_rows < (long)m_count[0] ? slot_rows : 0) : prefix[b + 1] - prefix[b];
    if (n == 0) return;         // skipped image: its gate is never read
    const int Cb = ch_idx ? ch_cnt[b] : C;
    const float inv = 1.f / (float)n;
    // (a chain of dependent global reads per image -- partial sums, fc1 rows, fc2 rows: every loop below keeps several
    // independent loads in flight; per output the order of additions is the plain loop's)
    for (int c = tid; c < C; c += NT) {
        float s = 0.f;
        if (c < Cb) {
            int k = 0;
            for (; k + 4 <= splits; k += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = partial[((size_t)b * splits + k + u) * C + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) s += v[u];
            }
            for (; k < splits; ++k) s += partial[((size_t)b * splits + k) * C + c];
        }
        s_mean[c] = s * inv;
        s_ch[c] = c < Cb ? (ch_idx ? ch_idx[(size_t)b * C + c] : c) : 0;
    }
    __syncthreads();
    constexpr int OB = 8;                              // fc1 outputs per wave and pass: their weight loads fly together
    for (int o0 = wave; o0 < S; o0 += NW * OB) {
        float acc[OB];
#pragma unroll
        for (int u = 0; u < OB; ++u) acc[u] = 0.f;
        for (int c = lane; c < Cb; c += 64) {
            const int ch = s_ch[c];
            const float m = s_mean[c];
            float wv[OB];
#pragma unroll
            for (int u = 0; u < OB; ++u) wv[u] = o0 + NW * u < S ? w1[(size_t)(o0 + NW * u) * C + ch] : 0.f;
#pragma unroll
            for (int u = 0; u < OB; ++u) acc[u] += wv[u] * m;
        }
#pragma unroll
        for (int u = 0; u < OB; ++u) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[u] += __shfl_xor(acc[u], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < OB; ++u)
                if (o0 + NW * u < S) s_hid[o0 + NW * u] = fmaxf(acc[u] + b1[o0 + NW * u], 0.f);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        float g = 0.f;
        if (c < Cb) {
            const int ch = s_ch[c];
            float acc = b2[ch];
            const float* wr = w2 + (size_t)ch * S;
            int j = 0;
            for (; j + 16 <= S; j += 16) {      // sixteen weights in flight, added in j order
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = wr[j + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += wv[u] * s_hid[j + u];
            }
            for (; j + 8 <= S; j += 8) {
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = wr[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += wv[u] * s_hid[j + u];
            }
            for (; j < S; ++j) acc += wr[j] * s_hid[j];
            g = 1.f / (1.f + __expf(-acc));
        }
        gate[(size_t)b * C + c] = g;
    }
}

// ---- channel mode (laud_regnet.py:160-189): grouped 3x3 conv + BN + ReLU over a dense image whose channels are LEFT-PACKED per
// image (column j of image b = channel ch_idx[b, j], ascending, j < ch_cnt[b]).  Output column j (channel c, group g = c / gw)
// sums over the ACTIVE input channels of its group only -- they occupy a contiguous run of packed columns -- so skipped
// channels cost nothing: out[b,p,j] = act(scale[c] * sum_t sum_{q in run(g)} a[b, pix(p,t), q] * w[c, t, ch(q) - g*gw] + shift[c]).
// Columns j >= ch_cnt[b] are written as zeros.  One workgroup = one image x a run of output pixels; thread = (pixel, column).
__global__ __launch_bounds__(256) void k_grouped3x3_chan(const float* __restrict__ a, int lda, int Hi, int Wi, int stride,
                                                          int Ho, int Wo, const float* __restrict__ w, int C, int gw,
                                                          const int32_t* __restrict__ ch_idx, const int32_t* __restrict__ ch_cnt,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int relu, float* __restrict__ out, int ldo) {
    extern __shared__ int s_i[];
    int* s_ch = s_i;                 // [C] channel of packed column j
    int* s_lo = s_i + C;             // [C / gw] first packed column of group g
    int* s_hi = s_lo + C / gw;       // [C / gw] one past its last packed column
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Cb = min(ch_cnt[b], C), G = C / gw;
    for (int g = tid; g < G; g += 256) { s_lo[g] = 0; s_hi[g] = 0; }
    for (int j = tid; j < Cb; j += 256) {
        s_ch[j] = ch_idx[(size_t)b * C + j];
        LDN_DCHECK(s_ch[j] >= 0 && s_ch[j] < C, 401);                              // channel list entries
        LDN_DCHECK(j == 0 || s_ch[j] > ch_idx[(size_t)b * C + j - 1], 402);        // ascending (active runs per group are contiguous)
    }
    LDN_DCHECK(ch_cnt[b] >= 0 && ch_cnt[b] <= C, 403);
    __syncthreads();
    for (int j = tid; j < Cb; j += 256) {
        const int g = s_ch[j] / gw;
        if (j == 0 || s_ch[j - 1] / gw != g) s_lo[g] = j;
        if (j == Cb - 1 || s_ch[j + 1] / gw != g) s_hi[g] = j + 1;
    }
    __syncthreads();
    const int HWo = Ho * Wo;
    const long total = (long)HWo * C;
    for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
        const int p = (int)(i / C), j = (int)(i - (long)p * C);
        float v = 0.f;
        if (j < Cb) {
            const int c = s_ch[j], g = c / gw, g0 = g * gw, lo = s_lo[g], hi = s_hi[g];
            const int oy = p / Wo, ox = p - oy * Wo;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy * stride + t / 3 - 1, ix = ox * stride + t % 3 - 1;
                if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
                const float* ap = a + ((size_t)(b * Hi + iy) * Wi + ix) * lda;
                const float* wp = w + ((size_t)c * 9 + t) * gw - g0;
                for (int q = lo; q < hi; ++q) acc += ap[q] * wp[s_ch[q]];
            }
            v = acc * scale[c] + shift[c];
            if (relu) v = fmaxf(v, 0.f);
        }
        out[((size_t)b * HWo + p) * ldo + j] = v;
    }
}

// a[r, :] *= gate[image(r), :] for the packed rows of every image; grid (chunks, B)
__global__ __launch_bounds__(256) void k_rows_scale(float* __restrict__ a, int lda, const int32_t* __restrict__ prefix,
                                                     int C, const float* __restrict__ gate) {
    const int b = blockIdx.y;
    const int r0 = prefix[b], n = prefix[b + 1] - r0;
    const int C4 = C >> 2;
    const long total = (long)n * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C4), c = (int)(i - (long)r * C4) * 4;
        f32x4* p = reinterpret_cast<f32x4*>(a + (size_t)(r0 + r) * lda + c);
        *p = *p * *reinterpret_cast<const f32x4*>(gate + (size_t)b * C + c);
    }
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_regnet)

}  // namespace ldn

using namespace ldn;

extern "C" int ldn_grouped_conv3x3_rows(const float* a, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap,
                                        const float* w, int C, int group_width, const float* scale, const float* shift,
                                        int relu, float* out, int ldo, void* stream) {
    LDN_REQUIRE(a && nbr && w && scale && shift && out, "ldn_grouped_conv3x3_rows: null pointer");
    LDN_REQUIRE(C > 0 && group_width > 0 && C % group_width == 0 && group_width % 4 == 0,
                "ldn_grouped_conv3x3_rows: channels must be a multiple of the group width, group width a multiple of 4");
    LDN_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && lda >= C && ldo >= C, "ldn_grouped_conv3x3_rows: strides must be multiples of 4");
    if (m_cap <= 0) return LDN_OK;
    long blocks = ((long)m_cap * (C / 4) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define LDN_LAUNCH_G(G4)                                                                                              \
    hipLaunchKernelGGL(k_grouped3x3_rows<G4>, dim3((unsigned)blocks), dim3(256), 0, st, a, lda, nbr, m_count, m_cap, w, C, \
                       group_width, scale, shift, relu, out, ldo)
#define LDN_LAUNCH_L(GW)                                                                                              \
    {                                                                                                                 \
        constexpr int rows_per_block = 256 / (GW / 4);                                                               \
        long bx = (m_cap + rows_per_block - 1) / rows_per_block;                                                      \
        const long cap = (256L * 16) / (C / GW) + 1;                                                                  \
        if (bx > cap) bx = cap;                                                                                       \
        hipLaunchKernelGGL(k_grouped3x3_lds<GW>, dim3((unsigned)bx, C / GW), dim3(256), 0, st, a, lda, nbr, m_count,   \
                           m_cap, w, C, scale, shift, relu, out, ldo);                                                \
    }
    switch (group_width) {
        case 8: LDN_LAUNCH_L(8); break;
        case 16: LDN_LAUNCH_L(16); break;
        case 24: LDN_LAUNCH_L(24); break;
        default: LDN_LAUNCH_G(0); break;
    }
#undef LDN_LAUNCH_L
#undef LDN_LAUNCH_G
    LDN_CHECK_LAUNCH("k_grouped3x3_rows");
    return LDN_OK;
}

extern "C" size_t ldn_se_packed_workspace_bytes(int B, int C, int max_rows_per_image) {
    return (size_t)B * (ldn_channel_masker_splits(max_rows_per_image) + 1) * C * sizeof(float);
}

extern "C" int ldn_grouped_conv3x3_image(const float* a, int lda, int B, int Hi, int Wi, int stride, int Ho, int Wo,
                                         const float* w, int C, int group_width, const int32_t* ch_idx, const int32_t* ch_cnt,
                                         const float* scale, const float* shift, int relu, float* out, int ldo, void* stream) {
    LDN_REQUIRE(a && w && ch_idx && ch_cnt && scale && shift && out, "ldn_grouped_conv3x3_image: null pointer");
    LDN_REQUIRE(C > 0 && group_width > 0 && C % group_width == 0, "ldn_grouped_conv3x3_image: channels must be a multiple of the group width");
    LDN_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && stride >= 1 && (Ho - 1) * stride < Hi && (Wo - 1) * stride < Wi,
                "ldn_grouped_conv3x3_image: bad geometry");
    LDN_REQUIRE(lda >= 1 && ldo >= C, "ldn_grouped_conv3x3_image: bad strides");
    long bx = ((long)Ho * Wo * C + 255) / 256;
    if (bx > 64) bx = 64;
    const size_t lds = (size_t)(C + 2 * (C / group_width)) * sizeof(int);
    hipLaunchKernelGGL(k_grouped3x3_chan, dim3((unsigned)bx, B), dim3(256), lds, static_cast<hipStream_t>(stream), a, lda, Hi, Wi,
                       stride, Ho, Wo, w, C, group_width, ch_idx, ch_cnt, scale, shift, relu, out, ldo);
    LDN_CHECK_LAUNCH("k_grouped3x3_chan");
    return LDN_OK;
}

extern "C" int ldn_se_packed(float* a, int lda, const int32_t* row_prefix, int B, int C, int S, const float* w1,
                             const float* b1, const float* w2, const float* b2, const int32_t* ch_idx, const int32_t* ch_cnt,
                             int max_rows_per_image, float* work, void* stream) {
    LDN_REQUIRE((ch_idx == nullptr) == (ch_cnt == nullptr), "ldn_se_packed: channel list and count must be given together");
    LDN_REQUIRE(a && row_prefix && w1 && b1 && w2 && b2 && work, "ldn_se_packed: null pointer");
    LDN_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && S > 0 && lda % 4 == 0, "ldn_se_packed: bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int splits = ldn_channel_masker_splits(max_rows_per_image);
    float* partial = work;                           // [B][splits][C]
    float* gate = work + (size_t)B * splits * C;     // [B][C]
    hipLaunchKernelGGL(k_rows_gap, dim3(splits, B), dim3(256), 0, st, a, lda, row_prefix, C, splits, partial);
    LDN_CHECK_LAUNCH("k_rows_gap");
    hipLaunchKernelGGL(k_se_head<1024>, dim3(B), dim3(1024), (size_t)(2 * C + S) * sizeof(float), st, partial, row_prefix, C, S, splits,
                       w1, b1, w2, b2, ch_idx, ch_cnt, gate, 0, nullptr);
    LDN_CHECK_LAUNCH("k_se_head");
    int chunks = (max_rows_per_image * (C / 4) + 255) / 256;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_rows_scale, dim3(chunks, B), dim3(256), 0, st, a, lda, row_prefix, C, gate);
    LDN_CHECK_LAUNCH("k_rows_scale");
    return LDN_OK;
}

extern "C" int ldn_se_gate_slots(const float* gap_partial, int splits, const int32_t* m_count, int images_cap, int rows_per_image, int C, int S,
                                 const float* w1, const float* b1, const float* w2, const float* b2, float* gate, void* stream) {
    LDN_REQUIRE(gap_partial && m_count && w1 && b1 && w2 && b2 && gate, "ldn_se_gate_slots: null pointer");
    LDN_REQUIRE(images_cap >= 0 && splits > 0 && rows_per_image > 0 && C > 0 && S > 0, "ldn_se_gate_slots: bad shape");
    if (images_cap == 0) return LDN_OK;
    hipLaunchKernelGGL(k_se_head<1024>, dim3(images_cap), dim3(1024), (size_t)(2 * C + S) * sizeof(float), static_cast<hipStream_t>(stream),
                       gap_partial, nullptr, C, S, splits, w1, b1, w2, b2, nullptr, nullptr, gate, rows_per_image, m_count);
    LDN_CHECK_LAUNCH("k_se_head");
    return LDN_OK;
}
