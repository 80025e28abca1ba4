// k_packed_mha -- multi-head self-attention over the KEPT tokens of every image (token skipping, BASELINE config 5: AdaViT /
// DeiT-S shaped blocks; the reference holds only the latency model of this operator, DyNetSimulator/adavit/simulate_adavit.py:77-131:
// q/k/v for all tokens, attention [B, heads, L_select, d] on the selected ones).  gfx950, bf16x3 arithmetic, fp32 softmax.
//
// One 512-thread workgroup per (image, head).  The image's kept tokens (<= 256, listed by flat row) are gathered ONCE: K as
// [key][64] fp32 and V transposed as [d][key] fp32 in LDS; wave w owns the 32 queries [32 w, 32 w + 32).  Everything is computed
// TRANSPOSED (lane = query), as in k_tail:
//   S^T[key][query] = K_chunk . Q^T        A operand = 32 keys x 16 d (two ds_read_b128 of the key's row), B operand = Q^T (registers)
//   the C layout of S^T (lane = query, 16 registers = keys) is, up to the fixed K-order permutation, the B layout of the next GEMM:
//   O^T[d][query] += V^T_chunk . P^T       A operand = 32 d x 16 keys (two ds_read_b128 of row d of V^T, keys in the permuted order)
// so the probabilities never leave the registers; the online softmax (running max / sum, rescaling of O^T) is a per-lane affair
// because a query is a lane (its 32 keys of a chunk sit in the two half-waves: one cross-half exchange per chunk).
#include "ldn_common.h"

namespace ldn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int A_D = 64;                 // head dimension
constexpr int A_KS = 68;                // row stride of K in LDS (floats): 16-byte aligned, 4 banks of shift per row
constexpr int A_MAXTOK = 256;

struct MhaArgs {
    const float* qkv; int ld;           // dense token rows [rows][ld]: q | k | v, each [heads][64]
    const int32_t* tok_rows;            // [N] flat row of every kept token
    const int32_t* prefix;              // [B + 1]
    int B, heads, dim;
    float scale;
    float* out; int ldo;                // packed rows [N][ldo]
    int vs;                             // row stride of V^T in LDS (floats): multiple of 4, = 4 mod 32
    const float* head_keep;             // optional [B][heads] {0,1}: head skipping (simulate_adavit.py:81-88) -- a dropped head's output is 0
};

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? a[e] : b[e - 4];
        const __bf16 hb = (__bf16)v;
        hi[e] = hb;
        lo[e] = (__bf16)(v - (float)hb);
    }
}

__global__ __launch_bounds__(512, 2) void k_packed_mha(const MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / p.heads, hd = blockIdx.x - b * p.heads;
    const int n0 = p.prefix[b];
    const int Lb = min(p.prefix[b + 1] - n0, A_MAXTOK);
    if (Lb <= 0) return;
    if (p.head_keep && p.head_keep[(size_t)b * p.heads + hd] < 0.5f) {   // head skipped for this image: its 64 output columns are zero
        for (int i = threadIdx.x; i < Lb * 16; i += 512)
            *reinterpret_cast<f32x4*>(p.out + (size_t)(n0 + (i >> 4)) * p.ldo + hd * A_D + (i & 15) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int Lp = round_up(Lb, 32);
    float* const s_k = reinterpret_cast<float*>(smem);                   // [Lp][A_KS]
    float* const s_vt = s_k + (size_t)Lp * A_KS;                         // [64][vs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    // ---- gather K rows and V^T of this (image, head): thread = (key, 4 d-values)
    for (int i = tid; i < Lp * 16; i += 512) {
        const int key = i >> 4, q4 = i & 15;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (key < Lb) {
            const float* row = p.qkv + (size_t)p.tok_rows[n0 + key] * p.ld + hd * A_D + q4 * 4;
            kv = *reinterpret_cast<const f32x4*>(row + p.dim);
            vv = *reinterpret_cast<const f32x4*>(row + 2 * p.dim);
        }
        *reinterpret_cast<f32x4*>(s_k + key * A_KS + q4 * 4) = kv;
#pragma unroll
        for (int e = 0; e < 4; ++e) s_vt[(q4 * 4 + e) * p.vs + key] = vv[e];
    }
    // ---- Q^T fragments of this wave's 32 queries: lane (query l31, half h), K16 step s: d = 16 s + 8 h .. + 7
    const int qi = wave * 32 + l31;
    const bool qvalid = qi < Lb;
    bf16x8 qh[4], ql[4];
    {
        const float* qrow = p.qkv + (size_t)p.tok_rows[n0 + (qvalid ? qi : 0)] * p.ld + hd * A_D;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h);
            const f32x4 c = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h + 4);
            split8(a, c, qh[s], ql[s]);
        }
    }
    __syncthreads();
    if (wave * 32 >= Lb) return;                                          // no queries (no further barriers below)

    f32x16 o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nchunk = Lp / 32;
    for (int c = 0; c < nchunk; ++c) {
        // S^T chunk: rows = keys 32 c + ..., columns = this wave's queries
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const float* krow = s_k + (32 * c + l31) * A_KS + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(krow + 16 * s);
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(krow + 16 * s + 4);
            bf16x8 kh, kl;
            split8(a, c4, kh, kl);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], sacc, 0, 0, 0);
        }
        // register r of the lane = key 32 c + (r & 3) + 8 (r >> 2) + 4 h of its query: scale, mask the padding keys, online softmax
        float mc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = key < Lb ? sacc[r] * p.scale : -INFINITY;
            sacc[r] = v;
            mc = fmaxf(mc, v);
        }
        mc = fmaxf(mc, __shfl_xor(mc, 32, 64));                          // the query's other 16 keys live in the partner half-wave
        const float m_new = fmaxf(m_run, mc);                            // finite: key 32 c is always a real key
        const float alpha = __expf(m_run - m_new);                       // 0 for the first chunk (m_run = -inf)
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(sacc[r] - m_new);                     // exp(-inf) = 0 on the padding keys
            sacc[r] = e;
            ls += e;
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
        // O^T += V^T_chunk . P^T: the K16 step t takes registers 8 t .. 8 t + 7 of P as its B operand (k-slot e <-> register 8 t + e
        // <-> key 16 t + (e & 3) + 8 (e >> 2) + 4 h); the A operand reads V^T in the same key order: two groups of four keys
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = sacc[8 * t + e];
                const __bf16 hb = (__bf16)v;
                ph[e] = hb;
                pl[e] = (__bf16)(v - (float)hb);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* vrow = s_vt + (32 * j + l31) * p.vs + 32 * c + 16 * t + 4 * h;
                const f32x4 a = *reinterpret_cast<const f32x4*>(vrow);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(vrow + 8);
                bf16x8 vh, vl;
                split8(a, c4, vh, vl);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[j], 0, 0, 0);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[j], 0, 0, 0);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[j], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (!qvalid) return;
    float* orow = p.out + (size_t)(n0 + qi) * p.ldo + hd * A_D;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {                                  // registers 4 q4 .. 4 q4 + 3 = d 32 j + 8 q4 + 4 h + {0..3}
            const f32x4 v = {o[j][4 * q4] * inv, o[j][4 * q4 + 1] * inv, o[j][4 * q4 + 2] * inv, o[j][4 * q4 + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + 32 * j + 8 * q4 + 4 * h) = v;
        }
}

}  // namespace ldn

using namespace ldn;

static int packed_mha_impl(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                           int head_dim, int max_tokens, float scale, const float* head_keep, float* out, int ldo, void* stream);
extern "C" int ldn_packed_mha(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                              int head_dim, int max_tokens, float scale, float* out, int ldo, void* stream) {
    return packed_mha_impl(qkv, ld_qkv, tok_rows, img_prefix, B, heads, head_dim, max_tokens, scale, nullptr, out, ldo, stream);
}
extern "C" int ldn_packed_mha_heads(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                                    int head_dim, int max_tokens, float scale, const float* head_keep, float* out, int ldo, void* stream) {
    LDN_REQUIRE(head_keep, "ldn_packed_mha_heads: null head_keep");
    return packed_mha_impl(qkv, ld_qkv, tok_rows, img_prefix, B, heads, head_dim, max_tokens, scale, head_keep, out, ldo, stream);
}
static int packed_mha_impl(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                           int head_dim, int max_tokens, float scale, const float* head_keep, float* out, int ldo, void* stream) {
    LDN_REQUIRE(qkv && tok_rows && img_prefix && out, "ldn_packed_mha: null pointer");
    LDN_REQUIRE(head_dim == A_D, "ldn_packed_mha: head_dim must be 64 (got %d)", head_dim);
    LDN_REQUIRE(B > 0 && heads > 0 && max_tokens > 0 && max_tokens <= A_MAXTOK, "ldn_packed_mha: at most %d kept tokens per image (got %d)", A_MAXTOK, max_tokens);
    const int dim = heads * head_dim;
    LDN_REQUIRE(ld_qkv >= 3 * dim && ld_qkv % 4 == 0 && ldo >= dim && ldo % 4 == 0, "ldn_packed_mha: bad row strides");
    LDN_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 16 == 0, "ldn_packed_mha: qkv / out must be 16-byte aligned");
    MhaArgs a{};
    a.qkv = qkv; a.ld = ld_qkv; a.tok_rows = tok_rows; a.prefix = img_prefix; a.B = B; a.heads = heads; a.dim = dim;
    a.scale = scale; a.out = out; a.ldo = ldo; a.head_keep = head_keep;
    const int Lp = round_up(max_tokens, 32);
    a.vs = Lp + 4;                                                        // = 4 mod 32: the 32 d-rows of a fragment read hit distinct banks
    const size_t lds = ((size_t)Lp * A_KS + (size_t)A_D * a.vs) * 4;
    LDN_REQUIRE(lds <= 160 * 1024, "ldn_packed_mha: %zu B of LDS exceed 160 KiB", lds);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_packed_mha), lds), "k_packed_mha: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k_packed_mha, dim3((unsigned)B * heads), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    LDN_CHECK_LAUNCH("k_packed_mha");
    return LDN_OK;
}
