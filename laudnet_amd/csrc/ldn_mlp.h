// Device-side pieces shared by the mask / index kernels (csrc/ldn_index.hip) and the chained stage kernel (csrc/ldn_tail.hip).
#pragma once
#include "ldn_common.h"

namespace ldn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// rank of this lane among the set lanes of `flag` in its wave, and the wave total
__device__ __forceinline__ int wave_rank(bool flag, int& total) {
    const unsigned long long m = __ballot(flag);
    total = __popcll(m);
    const int lane = threadIdx.x & 63;
    return __popcll(m & ((1ull << lane) - 1ull));
}

// block_rank for a block of NW waves
template <int NW>
__device__ __forceinline__ int block_rank_n(bool flag, int* s_w, int& chunk_total) {
    int wtot;
    const int r = wave_rank(flag, wtot);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect s_w from the previous call
    if ((threadIdx.x & 63) == 0) s_w[wave] = wtot;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int v = s_w[i];
        before += i < wave ? v : 0;
        total += v;
    }
    chunk_total = total;
    return before + r;
}

// Channel masker of ONE image b (models/utils.py:92-131 in eval mode): GAP from split partials -> MLP -> keep iff
// logit_keep >= logit_drop -> ordered list of the active channels.  Called by k_channel_mlp (one workgroup per image) and by
// k_chain (csrc/ldn_tail.hip).  Per output the summation order does not depend on NT: decisions are identical for any NT.
// s_f: C + max(hidden, 1) + 2 G floats of LDS; s_w: NT / 64 ints.
template <int NT>
__device__ __forceinline__ void channel_mlp_body(const int b, const float* __restrict__ partial, int HW, int C, int splits,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      int hidden, int G, int gran, const float* __restrict__ mask_in,
                                                      float* __restrict__ mask, float* __restrict__ logits,
                                                      int32_t* __restrict__ ch_idx, int32_t* __restrict__ ch_cnt,
                                                      float* s_f, int* s_w, float* s_part = nullptr,     // s_part: optional NT * 4 floats
                                                      bool wide_latency = false) {                       // the latency form for hidden <= 8 NW (k_chain_ld)
    float* s_gap = s_f;                 // [C]
    float* s_hid = s_gap + C;           // [max(hidden,1)]
    float* s_log = s_hid + (hidden > 0 ? hidden : 1);  // [2G]
    constexpr int NW = NT / 64;          // waves per block (one block per image)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G2 = 2 * G;
    if (!mask_in) {
        const float inv = 1.f / (float)HW;
        const int n1 = hidden > 0 ? hidden : G2;     // outputs of the first (or only) layer
        float* dst1 = hidden > 0 ? s_hid : s_log;
        // LATENCY form (the masker is a chain of dependent global reads: GAP partials -> layer 1 weights -> layer 2 weights -> list;
        // measured inside k_chain: 36 k cycles per block, 6 % of the run): when a wave owns at most two layer-1 outputs and a lane's
        // share of a weight row is at most 8 quads, ALL global loads of the phase -- the lane's layer-1 weights, its layer-2 row, the
        // GAP partials -- are issued before the first of them is waited for.  Per output the summation order is unchanged.
        const bool lat = hidden > 0 && hidden <= 2 * NW && (C & 255) == 0 && C <= 2048 && (hidden & 3) == 0 && hidden <= 16 &&
                         (reinterpret_cast<uintptr_t>(w1) & 15) == 0 && (reinterpret_cast<uintptr_t>(w2) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
        // LATENCY form for the wider hidden layer of stage 3 (hidden = 64 = 8 outputs per wave; round 6, inside k_chain_ld the masker is a serial
        // gap between two blocks' matrix phases: 28 k cycles per block as three dependent rounds of global reads): every global load of the phase --
        // the wave's eight layer-1 rows (four quads per lane each), the thread's layer-2 row, the GAP partials -- is issued before the first is
        // waited for.  Per output the summation order (and every expression) is the general form's below: identical decisions.
        const bool lat2 = wide_latency && !lat && hidden > 0 && hidden <= 8 * NW && (C & 255) == 0 && C <= 1024 && (hidden & 3) == 0 && hidden <= 64 &&
                          G2 <= NT && splits <= 8 && (reinterpret_cast<uintptr_t>(w1) & 15) == 0 && (reinterpret_cast<uintptr_t>(w2) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
        if (lat2) {
            const int nq = C >> 8;                                   // quads of a weight row per lane (<= 4)
            f32x4 wq[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int o = wave + NW * k;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wq[k][q] = (o < n1 && q < nq) ? *reinterpret_cast<const f32x4*>(w1 + (size_t)o * C + lane * 4 + 256 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x4 w2q[16];
            const int o2 = tid;                                      // layer-2 output of this thread
#pragma unroll
            for (int q = 0; q < 16; ++q)
                w2q[q] = (o2 < G2 && 4 * q < hidden) ? *reinterpret_cast<const f32x4*>(w2 + (size_t)o2 * hidden + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float b2v = o2 < G2 ? b2[o2] : 0.f;
            float b1v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) b1v[k] = wave + NW * k < n1 ? b1[wave + NW * k] : 0.f;
            // GAP: four channels per thread, the partials of the (<= 8) splits in flight together, added in split order
            for (int c = tid * 4; c < C; c += NT * 4) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
                f32x4 pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    pv[k] = k < splits ? *reinterpret_cast<const f32x4*>(partial + ((size_t)b * splits + k) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < splits) sacc += pv[k];
                *reinterpret_cast<f32x4*>(s_gap + c) = sacc * inv;
            }
            __syncthreads();
            float a1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a1[k] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(s_gap + lane * 4 + 256 * q);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (wave + NW * k < n1) a1[k] += wq[k][q][0] * g[0] + wq[k][q][1] * g[1] + wq[k][q][2] * g[2] + wq[k][q][3] * g[3];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) a1[k] = wave_sum(a1[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int o = wave + NW * k;
                    if (o < n1) dst1[o] = fmaxf(a1[k] + b1v[k], 0.f);
                }
            }
            __syncthreads();
            if (o2 < G2) {
                float a = b2v;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (4 * q < hidden) {
                        const int j = 4 * q;
                        a += w2q[q][0] * s_hid[j];
                        a += w2q[q][1] * s_hid[j + 1];
                        a += w2q[q][2] * s_hid[j + 2];
                        a += w2q[q][3] * s_hid[j + 3];
                    }
                }
                s_log[o2] = a;
            }
            __syncthreads();
        } else
        if (lat) {
            const int nq = C >> 8;                                   // quads of a weight row per lane (<= 8)
            f32x4 wq[2][8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = wave + NW * k;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    wq[k][q] = (o < n1 && q < nq) ? *reinterpret_cast<const f32x4*>(w1 + (size_t)o * C + lane * 4 + 256 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x4 w2q[4];
            const int o2 = tid;                                      // layer-2 output of this thread (G2 <= NT in this form, else looped below)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w2q[q] = (o2 < G2 && 4 * q < hidden) ? *reinterpret_cast<const f32x4*>(w2 + (size_t)o2 * hidden + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float b2v = o2 < G2 ? b2[o2] : 0.f;
            float b1v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) b1v[k] = wave + NW * k < n1 ? b1[wave + NW * k] : 0.f;
            // GAP.  Many splits on a narrow layer (the fused hand-off of stages 1 / 2: 112 / 32 partials of 256 / 512 channels -- a
            // quarter-wave to a wave of threads walking them one after the other was 35 us of pure latency per block): the NT / (C / 4)
            // thread groups each sum a contiguous range of splits, the group sums are added in group order (deterministic).  Only
            // for splits > 16 and with scratch from the caller: k_chain (8 splits) and every 8-split hand-off keep the plain order.
            const int nact = C >> 2;
            if (s_part && splits > 16 && nact < NT && NT % nact == 0) {
                const int R = NT / nact, r = tid / nact, c = (tid - r * nact) * 4;
                const int per = (splits + R - 1) / R;
                const int k_lo = r * per, k_hi = min(splits, k_lo + per);
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = k_lo; k0 < k_hi; k0 += 8) {
                    f32x4 pv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        pv[k] = k0 + k < k_hi ? *reinterpret_cast<const f32x4*>(partial + ((size_t)b * splits + k0 + k) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k0 + k < k_hi) sacc += pv[k];
                }
                *reinterpret_cast<f32x4*>(s_part + (size_t)r * C + c) = sacc;
                __syncthreads();
                for (int c2 = tid * 4; c2 < C; c2 += NT * 4) {
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
                    for (int r2 = 0; r2 < R; ++r2) t += *reinterpret_cast<const f32x4*>(s_part + (size_t)r2 * C + c2);
                    *reinterpret_cast<f32x4*>(s_gap + c2) = t * inv;
                }
            } else
            // four channels per thread, the partials of up to 8 splits in flight together, added in split order
            for (int c = tid * 4; c < C; c += NT * 4) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < splits; k0 += 8) {
                    f32x4 pv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        pv[k] = k0 + k < splits ? *reinterpret_cast<const f32x4*>(partial + ((size_t)b * splits + k0 + k) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k0 + k < splits) sacc += pv[k];
                }
                *reinterpret_cast<f32x4*>(s_gap + c) = sacc * inv;
            }
            __syncthreads();
            float a1[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < nq) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(s_gap + lane * 4 + 256 * q);
#pragma unroll
                    for (int k = 0; k < 2; ++k) a1[k] += wq[k][q][0] * g[0] + wq[k][q][1] * g[1] + wq[k][q][2] * g[2] + wq[k][q][3] * g[3];
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) a1[k] = wave_sum(a1[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int o = wave + NW * k;
                    if (o < n1) dst1[o] = fmaxf(a1[k] + b1v[k], 0.f);
                }
            }
            __syncthreads();
            for (int o = tid; o < G2; o += NT) {
                float a = o == o2 ? b2v : b2[o];
                const float* wr = w2 + (size_t)o * hidden;
                for (int j = 0; j < hidden; j += 4) {
                    const f32x4 v = o == o2 ? w2q[j >> 2] : *reinterpret_cast<const f32x4*>(wr + j);
                    a += v[0] * s_hid[j];
                    a += v[1] * s_hid[j + 1];
                    a += v[2] * s_hid[j + 2];
                    a += v[3] * s_hid[j + 3];
                }
                s_log[o] = a;
            }
            __syncthreads();
        } else {
        if (s_part && splits > 16 && C <= NT && NT % C == 0) {
            // many partials of a narrow map (the fused stem's hand-off: 56 tiles x 64 channels): NT / C thread groups each sum a contiguous
            // range of splits with all loads in flight, the group sums are added in group order (deterministic) -- one wave walking 56
            // dependent loads was 30 us of pure latency in front of the first block
            const int R = NT / C, r = tid / C, c = tid - r * C;
            const int per = (splits + R - 1) / R;
            const int k_lo = r * per, k_hi = min(splits, k_lo + per);
            float sacc = 0.f;
            for (int k0 = k_lo; k0 < k_hi; k0 += 8) {
                float pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = k0 + k < k_hi ? partial[((size_t)b * splits + k0 + k) * C + c] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k0 + k < k_hi) sacc += pv[k];
            }
            s_part[(size_t)r * C + c] = sacc;
            __syncthreads();
            if (tid < C) {
                float t = 0.f;
                for (int r2 = 0; r2 < R; ++r2) t += s_part[(size_t)r2 * C + tid];
                s_gap[tid] = t * inv;
            }
        } else
        for (int c = tid; c < C; c += NT) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += partial[((size_t)b * splits + k) * C + c];
            s_gap[c] = s * inv;
        }
        __syncthreads();
        // each wave owns outputs wave, wave + NW, ...; eight of them per pass so that their weight loads are in flight
        // together and their cross-lane reductions interleave (per output the summation order is unchanged)
        constexpr int OB = 8;
        for (int o0 = wave; o0 < n1; o0 += NW * OB) {
            float acc[OB];
#pragma unroll
            for (int k = 0; k < OB; ++k) acc[k] = 0.f;
            if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(w1) & 15) == 0) {
                // 16 bytes per lane: a wave instruction covers 256 consecutive weights of a row (1 KiB instead of 256 B)
                for (int c = lane * 4; c < C; c += 256) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(s_gap + c);
#pragma unroll
                    for (int k = 0; k < OB; ++k) {
                        const int o = o0 + NW * k;
                        if (o < n1) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + (size_t)o * C + c);
                            acc[k] += wv[0] * g[0] + wv[1] * g[1] + wv[2] * g[2] + wv[3] * g[3];
                        }
                    }
                }
            } else {
                for (int c = lane; c < C; c += 64) {
                    const float g = s_gap[c];
#pragma unroll
                    for (int k = 0; k < OB; ++k) {
                        const int o = o0 + NW * k;
                        if (o < n1) acc[k] += w1[(size_t)o * C + c] * g;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < OB; ++k) acc[k] = wave_sum(acc[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < OB; ++k) {
                    const int o = o0 + NW * k;
                    if (o < n1) {
                        const float a = acc[k] + b1[o];
                        dst1[o] = hidden > 0 ? fmaxf(a, 0.f) : a;
                    }
                }
            }
        }
        __syncthreads();
        if (hidden > 0) {
            // one thread per output; its weight row is read as 16-byte vectors (consecutive instructions of a lane stay
            // inside the same cache lines), s_hid reads are LDS broadcasts
            for (int o = tid; o < G2; o += NT) {
                float a = b2[o];
                const float* wr = w2 + (size_t)o * hidden;
                if ((hidden & 3) == 0 && (reinterpret_cast<uintptr_t>(wr) & 15) == 0) {
                    for (int j = 0; j < hidden; j += 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(wr + j);
                        a += v[0] * s_hid[j];
                        a += v[1] * s_hid[j + 1];
                        a += v[2] * s_hid[j + 2];
                        a += v[3] * s_hid[j + 3];
                    }
                } else {
                    for (int j = 0; j < hidden; ++j) a += wr[j] * s_hid[j];
                }
                s_log[o] = a;
            }
            __syncthreads();
        }
        }   // !lat
        for (int j = tid; j < G; j += NT) mask[(size_t)b * G + j] = s_log[j] >= s_log[G + j] ? 1.f : 0.f;
        if (logits)
            for (int o = tid; o < G2; o += NT) logits[(size_t)b * G2 + o] = s_log[o];
    } else {
        for (int j = tid; j < G; j += NT) mask[(size_t)b * G + j] = mask_in[(size_t)b * G + j];
    }
    __syncthreads();
    // ordered compaction of the active channels (group j owns [j*gran, (j+1)*gran))
    const int width = G * gran;
    int running = 0;
    for (int c0 = 0; c0 < width; c0 += NT) {
        const int c = c0 + tid;
        bool f = false;
        if (c < width) {
            const int j = c / gran;
            f = mask_in ? mask_in[(size_t)b * G + j] > 0.5f : s_log[j] >= s_log[G + j];
#ifdef LDN_MASK_HASH   // tuning only (ablation builds whose activations are garbage): data-independent decisions at a fixed density (per mille)
            f = ((unsigned)(j * 2654435761u + (unsigned)b * 40503u) >> 12) % 1000u < (unsigned)(LDN_MASK_HASH);
#endif
        }
        int tot;
        const int r = block_rank_n<NW>(f, s_w, tot);
        if (f) ch_idx[(size_t)b * width + running + r] = c;
        running += tot;
    }
    if (tid == 0) ch_cnt[b] = running;
}

}  // namespace ldn
