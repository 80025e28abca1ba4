// Mask / index kernels of the LAUDNet hot path (gfx950): maskers, mask -> packed index lists
// (wave64 ballot + popcount prefix sums), stand-alone row gather and masked scatter-add.
#include <stdarg.h>

#include <map>
#include <mutex>

#include "ldn_common.h"
#include "ldn_mlp.h"

namespace ldn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Raise a kernel's dynamic-LDS ceiling above the 64 KiB default when a launch needs it (gfx950: 160 KiB/CU).
bool allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    // the attribute is set once per (kernel, device) and raised when a launch needs more -- not on every launch
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    static const bool always = getenv("LDN_LDS_ATTR_ALWAYS") != nullptr;      // A/B: the attribute call in front of every launch
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{kernel, dev}];
    if (have >= bytes && !always) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    have = bytes;
    return true;
}

// Ordered compaction step for one chunk of 256 candidates (one per thread) inside a 256-thread block.
// Returns the rank of this thread's element among the chunk's set flags (valid if flag), and adds the
// chunk total to `running` (uniform across the block).  s_w: 4 ints of LDS.
__device__ __forceinline__ int block_rank(bool flag, int* s_w, int& chunk_total) {
    int wtot;
    const int r = wave_rank(flag, wtot);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect s_w from the previous call
    if ((threadIdx.x & 63) == 0) s_w[wave] = wtot;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) before += i < wave ? s_w[i] : 0;
    chunk_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return before + r;
}

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) {   // one fixed contraction (shared by the masker heads of k_spatial_masker and k_plan: the same floats from the same means)
    return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}

// ---------------------------------------------------------------------------------------- a1
// one wave per (image, patch); lanes stride over channels.
__global__ __launch_bounds__(256) void k_spatial_masker(const float* __restrict__ x, int B, int Hi, int Wi, int C,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         int g, int S, float* __restrict__ mask,
                                                         float* __restrict__ logits, float* __restrict__ pool_work,
                                                         const float* __restrict__ carry_mask) {
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool pooled = S < Hi;  // models/utils.py:48
    const int Sy = pooled ? S : Hi, Sx = pooled ? S : Wi;
    if (job >= B * Sy * Sx) return;
    const int b = job / (Sy * Sx), pp = job - b * Sy * Sx;
    const int py = pp / Sx, px = pp - py * Sx;
    int y0 = py, y1 = py + 1, x0 = px, x1 = px + 1;
    if (pooled) {  // adaptive pool bins: floor(i*H/S) .. ceil((i+1)*H/S)
        y0 = (py * Hi) / S; y1 = ((py + 1) * Hi + S - 1) / S;
        x0 = (px * Wi) / S; x1 = ((px + 1) * Wi + S - 1) / S;
    }
    const float inv = 1.f / float((y1 - y0) * (x1 - x0));
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    const int G2 = 2 * g;
    // Patch carry: pool_work [B][Sy][Sx][C] keeps every patch's pooled channel means.  A patch the PREVIOUS block of the same
    // residual stream left untouched (carry_mask[b][patch] == 0: its conv3 wrote no pixel of it) has the means that block's masker
    // stored -- they are reused instead of re-reading the window (the same floats: bit-identical decisions).
    float* const pw_row = pool_work ? pool_work + ((size_t)b * Sy * Sx + pp) * C : nullptr;
    const bool reuse = pw_row && carry_mask && carry_mask[(size_t)b * Sy * Sx + pp] < 0.5f;
    if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        // 16 bytes per lane: one wave instruction covers 256 channels of a pixel; the pixel loop is unrolled so that several
        // KiB per wave are in flight (this kernel is one HBM pass over x)
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (reuse) {
                s = *reinterpret_cast<const f32x4*>(pw_row + c);
            } else {
            // the window's pixels in row-major order, eight loads in flight per lane, added in that order (same sums as the plain loop)
            const int pw = x1 - x0, np = (y1 - y0) * pw;
            const float* base = x + ((size_t)(b * Hi + y0) * Wi + x0) * C + c;
            int pi = 0;
            for (; pi + 8 <= np; pi += 8) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int py = (pi + k) / pw, px = (pi + k) - py * pw;
                    v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + ((size_t)py * Wi + px) * C));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            for (; pi < np; ++pi) {
                const int py = pi / pw, px = pi - py * pw;
                s += *reinterpret_cast<const f32x4*>(base + ((size_t)py * Wi + px) * C);
            }
            s *= inv;
            if (pw_row) *reinterpret_cast<f32x4*>(pw_row + c) = s;
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < G2) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + o * C + c);
                    acc[o] += dot4(wv, s);
                }
        }
    } else {
        for (int c = lane; c < C; c += 64) {
            float s = 0.f;
            if (reuse) {
                s = pw_row[c];
            } else {
                for (int y = y0; y < y1; ++y)
                    for (int xx = x0; xx < x1; ++xx) s += x[((size_t)(b * Hi + y) * Wi + xx) * C + c];
                s *= inv;
                if (pw_row) pw_row[c] = s;
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < G2) acc[o] += w[o * C + c] * s;
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = wave_sum(acc[o]);
    if (lane < g) {
        float lk = 0.f, ld = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (o == lane) lk = acc[o] + bias[o];
            if (o == lane + g) ld = acc[o] + bias[o];
        }
        const size_t plane = (size_t)Sy * Sx;
        mask[((size_t)b * g + lane) * plane + pp] = lk >= ld ? 1.f : 0.f;  // ties keep (utils.py:60)
        if (logits) {
            logits[((size_t)b * G2 + lane) * plane + pp] = lk;
            logits[((size_t)b * G2 + g + lane) * plane + pp] = ld;
        }
    }
}

// Per-PIXEL masks (mask_size == the map: models/utils.py:48 takes x itself, BASELINE config 1's granularity 1-1-1-1): k_spatial_masker gives
// every pixel its own wave with ONE 16-byte load per lane in flight -- latency-bound at 1.9 TB/s, 30 % of a LAUD-ResNet50 g = 1 forward
// (round 6, profiles/r06_spatial_g1_kernel_stats.txt).  Here a wave walks PX consecutive pixels with all of their rows requested before the
// first dot product; per pixel the arithmetic (lane l: channels 4 l + 256 k, dot4, the xor-butterfly wave sum, + bias, l_keep >= l_drop) is
// k_spatial_masker's: identical logits and decisions.  C % 4 == 0.
// Sixteen values per lane summed over the wave with 18 exchanges instead of 16 x 6: at the level that pairs lanes l and l ^ M a lane hands over
// the half of its values its partner keeps.  Per value the additions are the xor-butterfly's (levels 32, 16, 8, 4, 2, 1, own + partner at each):
// bit-identical to wave_sum.  Returns the total of value index ((lane >> 2) & 15) (bit 5 of the lane = bit 3 of the index, ...).
__device__ __forceinline__ float wave_sum16(float (&v)[16]) {
    const int lane = threadIdx.x & 63;
    float u8[8], u4[4], u2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float recv = __shfl_xor(b5 ? v[i] : v[i + 8], 32, 64);
        u8[i] = (b5 ? v[i + 8] : v[i]) + recv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float recv = __shfl_xor(b4 ? u8[i] : u8[i + 4], 16, 64);
        u4[i] = (b4 ? u8[i + 4] : u8[i]) + recv;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float recv = __shfl_xor(b3 ? u4[i] : u4[i + 2], 8, 64);
        u2[i] = (b3 ? u4[i + 2] : u4[i]) + recv;
    }
    float r = (b2 ? u2[1] : u2[0]) + __shfl_xor(b2 ? u2[0] : u2[1], 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

template <int PX, int G2>      // PX * G2 == 16: (8 pixels, one mask group) or (4 pixels, two mask groups)
__global__ __launch_bounds__(256) void k_pixel_masker(const float* __restrict__ x, int B, int HW, int C, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ mask, float* __restrict__ logits) {
    static_assert(PX * G2 == 16, "sixteen sums per wave");
    constexpr int g = G2 / 2;
    const int lane = threadIdx.x & 63;
    const long job = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per_img = (HW + PX - 1) / PX;
    if (job >= (long)B * per_img) return;
    const int b = (int)(job / per_img), p0 = (int)(job - (long)b * per_img) * PX;
    const int nk = (C + 255) >> 8;
    float acc[16];      // value index = pixel * G2 + output
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = 0; k < nk; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c >= C) break;                                   // (C % 4 == 0; lanes beyond a narrow map's channels add nothing, as in the general kernel)
        f32x4 v[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int pp = min(p0 + i, HW - 1);
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + pp) * C + c));
        }
        f32x4 wv[G2];
#pragma unroll
        for (int o = 0; o < G2; ++o) wv[o] = *reinterpret_cast<const f32x4*>(w + o * C + c);
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int o = 0; o < G2; ++o) acc[i * G2 + o] += dot4(wv[o], v[i]);
    }
    const float tot = wave_sum16(acc);                       // this lane: value (lane >> 2) & 15 = (pixel, output)
    const int vi = (lane >> 2) & 15, pix = vi / G2, o = vi % G2;
    const float mine = tot + bias[o];
    const float other = __shfl_xor(mine, 4 * g, 64);         // output o ^ g of the same pixel: the other logit of the (keep, drop) pair
    const int pp = p0 + pix;
    if ((lane & 3) == 0 && o < g && pp < HW) {
        const float lk = mine, ld = other;
        mask[((size_t)b * g + o) * HW + pp] = lk >= ld ? 1.f : 0.f;  // ties keep (utils.py:60)
        if (logits) {
            logits[((size_t)b * G2 + o) * HW + pp] = lk;
            logits[((size_t)b * G2 + g + o) * HW + pp] = ld;
        }
    }
}

// ---------------------------------------------------------------------------------------- a4 / a11
struct IdxGeom {
    int B, S, Sx, Ho, Wo, stride, Hi, Wi;      // patch mask [B][S][Sx]
};

__device__ __forceinline__ int nearest_src(int i, float scale, int S) {
    // ATen nearest: min(int(floorf(dst * scale)), in - 1), scale = float(in) / float(out)
    const int s = (int)floorf((float)i * scale);
    return s < S - 1 ? s : S - 1;
}

// fills s_m3[Ho*Wo] (bytes) for image b; returns nothing. all threads participate.
__device__ __forceinline__ void fill_mask3(const float* __restrict__ patch, const IdxGeom g, int b,
                                           unsigned char* s_m3) {
    const float sh = (float)g.S / (float)g.Ho, sw = (float)g.Sx / (float)g.Wo;
    const int n = g.Ho * g.Wo;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / g.Wo, x = i - y * g.Wo;
        const int sy = nearest_src(y, sh, g.S), sx = nearest_src(x, sw, g.Sx);
        s_m3[i] = patch[((size_t)b * g.S + sy) * g.Sx + sx] > 0.5f ? 1 : 0;
    }
}

__device__ __forceinline__ bool mask1_at(const unsigned char* s_m3, const IdxGeom g, int iy, int ix) {
    // set output pixels (oy,ox) with |iy - oy*s| <= 1 and |ix - ox*s| <= 1
    const int s = g.stride;
    int oy0 = (iy - 1 + s - 1) / s, oy1 = (iy + 1) / s;   // ceil((iy-1)/s) for iy-1 >= 0 ; clamp below
    if (iy - 1 < 0) oy0 = 0;
    int ox0 = (ix - 1 + s - 1) / s, ox1 = (ix + 1) / s;
    if (ix - 1 < 0) ox0 = 0;
    oy1 = oy1 < g.Ho - 1 ? oy1 : g.Ho - 1;
    ox1 = ox1 < g.Wo - 1 ? ox1 : g.Wo - 1;
    bool any = false;
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) any |= s_m3[oy * g.Wo + ox] != 0;
    return any;
}

__global__ __launch_bounds__(256) void k_mask_count(const float* __restrict__ patch, const IdxGeom g,
                                                     int32_t* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bytes[];
    unsigned char* s_m3 = s_bytes;
    __shared__ int s_red[3];
    const int b = blockIdx.x;
    if (threadIdx.x < 3) s_red[threadIdx.x] = 0;
    fill_mask3(patch, g, b, s_m3);
    __syncthreads();
    int c3 = 0, c1 = 0, cp = 0;
    for (int i = threadIdx.x; i < g.Ho * g.Wo; i += 256) c3 += s_m3[i];
    for (int i = threadIdx.x; i < g.Hi * g.Wi; i += 256) {
        const int iy = i / g.Wi, ix = i - iy * g.Wi;
        c1 += mask1_at(s_m3, g, iy, ix) ? 1 : 0;
    }
    for (int i = threadIdx.x; i < g.S * g.Sx; i += 256) cp += patch[(size_t)b * g.S * g.Sx + i] > 0.5f ? 1 : 0;
    atomicAdd(&s_red[0], c3);   // integer LDS atomics: order-independent
    atomicAdd(&s_red[1], c1);
    atomicAdd(&s_red[2], cp);
    __syncthreads();
    if (threadIdx.x < 3) work[threadIdx.x * g.B + b] = s_red[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_mask_index(const float* __restrict__ patch, const IdxGeom g,
                                                     const int32_t* __restrict__ work, int32_t* __restrict__ idx3,
                                                     int32_t* __restrict__ pos3, int32_t* __restrict__ idx1,
                                                     int32_t* __restrict__ pos1, int32_t* __restrict__ nbr,
                                                     int32_t* __restrict__ cnt, int32_t* __restrict__ pre3,
                                                     int32_t* __restrict__ pre1, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bytes[];
    const int HWo = g.Ho * g.Wo, HWi = g.Hi * g.Wi;
    int* s_pos3 = reinterpret_cast<int*>(s_bytes);
    int* s_pos1 = s_pos3 + HWo;
    unsigned char* s_m3 = reinterpret_cast<unsigned char*>(s_pos1 + HWi);
    __shared__ int s_w[4];
    __shared__ int s_base[3];
    const int b = blockIdx.x, tid = threadIdx.x;

    // exclusive prefix over the images before b (and, in the last block, the grand totals)
    if (tid < 3) s_base[tid] = 0;
    __syncthreads();
    {
        int a3 = 0, a1 = 0, ap = 0;
        const bool last = b == g.B - 1;
        const int upto = last ? g.B : b;
        for (int i = tid; i < upto; i += 256) {
            const bool own = i == b;  // only in the last block
            a3 += own ? 0 : work[i];
            a1 += own ? 0 : work[g.B + i];
            ap += work[2 * g.B + i] * ((last || i < b) ? 1 : 0);
        }
        atomicAdd(&s_base[0], a3);
        atomicAdd(&s_base[1], a1);
        atomicAdd(&s_base[2], ap);
    }
    fill_mask3(patch, g, b, s_m3);
    __syncthreads();
    const int base3 = s_base[0], base1 = s_base[1];
    if (tid == 0) {
        pre3[b] = base3;
        pre1[b] = base1;
        if (b == g.B - 1) {
            const int tot3 = base3 + work[b], tot1 = base1 + work[g.B + b];
            pre3[g.B] = tot3;
            pre1[g.B] = tot1;
            cnt[0] = tot3;
            cnt[1] = tot1;
            stats[0] = (float)s_base[2] / (float)((long)g.B * g.S * g.Sx);
            stats[1] = (float)tot3 / (float)((long)g.B * HWo);
            stats[2] = (float)tot1 / (float)((long)g.B * HWi);
        }
    }

    int running = 0;
    for (int i0 = 0; i0 < HWo; i0 += 256) {
        const int i = i0 + tid;
        const bool f = i < HWo && s_m3[i];
        int tot;
        const int r = block_rank(f, s_w, tot);
        if (i < HWo) {
            const int p = f ? base3 + running + r : -1;
            s_pos3[i] = p;
            pos3[(size_t)b * HWo + i] = p;
            LDN_DCHECK(!f || (p >= 0 && p < g.B * HWo), 201);                  // packed position inside the list capacity
            if (f) idx3[p] = b * HWo + i;
        }
        running += tot;
    }
    running = 0;
    for (int i0 = 0; i0 < HWi; i0 += 256) {
        const int i = i0 + tid;
        bool f = false;
        if (i < HWi) {
            const int iy = i / g.Wi, ix = i - iy * g.Wi;
            f = mask1_at(s_m3, g, iy, ix);
        }
        int tot;
        const int r = block_rank(f, s_w, tot);
        if (i < HWi) {
            const int p = f ? base1 + running + r : -1;
            s_pos1[i] = p;
            pos1[(size_t)b * HWi + i] = p;
            LDN_DCHECK(!f || (p >= 0 && p < g.B * HWi), 202);
            if (f) idx1[p] = b * HWi + i;
        }
        running += tot;
    }
    __syncthreads();
    for (int i = tid; i < HWo; i += 256) {
        const int p = s_pos3[i];
        if (p < 0) continue;
        const int oy = i / g.Wo, ox = i - oy * g.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy * g.stride - 1 + t / 3, ix = ox * g.stride - 1 + t % 3;
            const bool inb = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
            LDN_DCHECK(!inb || s_pos1[iy * g.Wi + ix] >= 0, 203);              // every in-bounds tap of an active pixel is in the dilated list
            nbr[(size_t)p * 9 + t] = inb ? s_pos1[iy * g.Wi + ix] : -1;
        }
    }
}

// ---- round 4: decision + counts + prefix over the images + lists in ONE launch (k_plan) -------------------------------------------
// k_mask_count + k_mask_index were two launches because image b's first packed position is the sum of the counts of the images in
// front of it.  Here workgroup b publishes its three counts (+ 1, so that 0 means "not yet") and waits for the counts of the
// workgroups in front of it, which were dispatched before it; the waiting loop is bounded in TIME (never a hang).  A failed wait is
// safe and loud: the workgroup raises the launch's failure word sync[0] BEFORE anything else and counts the event
// in g_plan_timeouts (sticky, ldn_plan_timeouts()); every workgroup re-reads sync[0] at its very end, after its own writes of the
// counts / prefixes have been acknowledged (write-through stores + vmcnt(0)), and a workgroup that finds it raised -- as well as the
// failing one -- zeroes cnt, stats and EVERY prefix.  Whatever the interleaving, the last write to each of those words is a zero
// (a successful workgroup either checked after the flag went up and zeroes behind its own values, or finished its writes before the
// flag went up and the failing workgroup zeroes behind them): consumers see empty lists, never uninitialised rows or prefixes.
// Two extras of the fused spatial path (models/utils.py:47-65 behind conv3's epilogue):
//   * decide mode (patch == nullptr): the patch decisions come from the POOLED CHANNEL MEANS [B][S*Sx][C] (the stand-alone masker's
//     `pool_work`, refreshed by the previous block's conv3 epilogue) -- k_spatial_masker's head, same arithmetic, no read of x;
//   * patch_major: idx3 lists the kept pixels patch by patch (row-major inside a patch) instead of row-major over the image, so
//     that the rows of one patch are contiguous in the packed tensors (conv3's epilogue then owns whole patches).  Same SET of
//     pixels, same counts / prefixes / statistics; pos3 / nbr follow the order.  Even grids only (Ho % S == 0, Wo % Sx == 0).
__device__ unsigned g_plan_timeouts = 0u;      // launches' failed prefix waits since the last reset (release builds too)
__device__ int* g_fault_dev = nullptr;          // the process's host-visible fault word (ldn_fault_flag), set by launch_plan once per device
#ifdef LDN_DEBUG
__device__ int g_plan_stall = -1;              // test hook (ldn_debug_plan_stall): this image never publishes its counts
#endif
struct PlanArgs {
    IdxGeom g;
    int patch_major;
    const float* patch;
    const float* pool; int C; const float* w; const float* bias; float* mask_out; float* logits;
    int32_t* sync;                 // [2 + 3 B], ZERO in front of the launch and left zero by it: failure word; per image count3 + 1, count1 + 1, patches + 1;
                                   // workgroups that have left (the last one out zeroes all of it for the next launch on this buffer)
    int32_t *idx3, *pos3, *idx1, *pos1, *nbr, *cnt, *pre3, *pre1;
    float* stats;
    long long timeout_ticks;       // bound of the prefix wait in wall_clock64() ticks (100 MHz); set by launch_plan
};

// (the flags are read and written with relaxed agent-scope atomics: the flag IS the payload, nothing else is published through it --
// an acquire load per poll would invalidate the L2 under the workgroups that are still computing)
__device__ __forceinline__ int wave_isum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// i / d for 0 <= i < 2^22 through the float reciprocal, corrected to the exact quotient
__device__ __forceinline__ int idiv(int i, int d, float inv) {
    int q = (int)(((float)i + 0.5f) * inv);
    const int r = i - q * d;
    q += r >= d ? 1 : (r < 0 ? -1 : 0);
    return q;
}

__global__ __launch_bounds__(256) void k_zero_i32(int32_t* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

constexpr int kPlanThreads = 512;
__global__ __launch_bounds__(kPlanThreads) void k_plan(const PlanArgs a) {
    constexpr int NT = kPlanThreads, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bytes[];
    const IdxGeom g = a.g;
    const int HWo = g.Ho * g.Wo, HWi = g.Hi * g.Wi, SS = g.S * g.Sx;
    int* s_pos3 = reinterpret_cast<int*>(s_bytes);           // [HWo] pixel -> packed position or -1
    int* s_pos1 = s_pos3 + HWo;                                // [HWi]
    int* s_list = s_pos1 + HWi;                                // [HWo] k-th kept output pixel of this image
    unsigned char* s_m3 = reinterpret_cast<unsigned char*>(s_list + HWo);   // [HWo]
    unsigned char* s_m1 = s_m3 + HWo;                          // [HWi] dilated mask
    unsigned char* s_patch = s_m1 + HWi;                       // [SS]
    const float inv_wo = 1.f / (float)g.Wo, inv_wi = 1.f / (float)g.Wi;
    __shared__ int s_w[NW];
    __shared__ int s_red[3], s_base[3], s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // image = blockIdx.x: workgroups are dispatched in index order (per XCD), so the lowest unfinished image is always resident and
    // waits on nothing unfinished -- progress by induction.  (A ticket from one atomic counter would make that independent of the
    // dispatch order, but 256 device-scope atomics on one address serialise across the 8 XCDs: measured ~40 us per launch.)
    if (tid == 0) s_fail = 0;
    if (tid < 3) { s_red[tid] = 0; s_base[tid] = 0; }
    __syncthreads();
    const int b = blockIdx.x;

    // 1. the patch decisions of this image
    if (a.patch) {
        for (int i = tid; i < SS; i += NT) s_patch[i] = a.patch[(size_t)b * SS + i] > 0.5f ? 1 : 0;
    } else {
        // a wave takes four patches at a time: their rows are in flight together (one patch after the other is one memory latency per
        // patch and wave: 25 patches x ~2 us at stage 1); per patch the arithmetic order is k_spatial_masker's
        const float b0 = a.bias[0], b1 = a.bias[1];
        // (round 6: the channel loop in blocks of four steps whose 16 row loads are issued together -- one step at a time the loop was a chain
        // of C / 256 memory latencies per patch group, ~30 us per launch at 1024 channels on the serial path between two blocks; same sums in the
        // same order)
        for (int p0 = wave * 4; p0 < SS; p0 += NW * 4) {
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            const float* rowp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rowp[u] = a.pool + ((size_t)b * SS + min(p0 + u, SS - 1)) * a.C;
            for (int c0 = lane * 4; c0 < a.C; c0 += 1024) {
                f32x4 sv[4][4], w0[4], w1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = min(c0 + 256 * k, a.C - 4);            // (steps beyond C re-read the last one; their sums are not taken)
#pragma unroll
                    for (int u = 0; u < 4; ++u) sv[k][u] = *reinterpret_cast<const f32x4*>(rowp[u] + c);
                    w0[k] = *reinterpret_cast<const f32x4*>(a.w + c);
                    w1[k] = *reinterpret_cast<const f32x4*>(a.w + a.C + c);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + 256 * k < a.C) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a0[u] += dot4(w0[k], sv[k][u]); a1[u] += dot4(w1[k], sv[k][u]); }
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float lk = wave_sum(a0[u]) + b0, ld = wave_sum(a1[u]) + b1;
                const int pp = p0 + u;
                if (lane == 0 && pp < SS) {
                    const bool keep = lk >= ld;                  // ties keep (utils.py:60)
                    s_patch[pp] = keep ? 1 : 0;
                    a.mask_out[(size_t)b * SS + pp] = keep ? 1.f : 0.f;
                    if (a.logits) {
                        a.logits[((size_t)b * 2) * SS + pp] = lk;
                        a.logits[((size_t)b * 2 + 1) * SS + pp] = ld;
                    }
                }
            }
        }
    }
    __syncthreads();
    // 2. mask3 (nearest interpolation of the patch decisions), the three counts of this image
    {
        const float sh = (float)g.S / (float)g.Ho, sw = (float)g.Sx / (float)g.Wo;
        for (int i = tid; i < HWo; i += NT) {
            const int y = idiv(i, g.Wo, inv_wo), x = i - y * g.Wo;
            s_m3[i] = s_patch[nearest_src(y, sh, g.S) * g.Sx + nearest_src(x, sw, g.Sx)];
        }
    }
    __syncthreads();
    {
        int c3 = 0, c1 = 0, cp = 0;
        for (int i = tid; i < HWo; i += NT) c3 += s_m3[i];
        for (int i = tid; i < HWi; i += NT) {
            const int iy = idiv(i, g.Wi, inv_wi), ix = i - iy * g.Wi;
            const unsigned char f = mask1_at(s_m3, g, iy, ix) ? 1 : 0;
            s_m1[i] = f;                                         // (the compaction pass reads the byte instead of dilating again)
            c1 += f;
        }
        for (int i = tid; i < SS; i += NT) cp += s_patch[i];
        c3 = wave_isum(c3); c1 = wave_isum(c1); cp = wave_isum(cp);
        if (lane == 0) {            // one integer LDS atomic per wave (all lanes on one address serialise): order-independent
            atomicAdd(&s_red[0], c3);
            atomicAdd(&s_red[1], c1);
            atomicAdd(&s_red[2], cp);
        }
    }
    __syncthreads();
#ifdef LDN_DEBUG
    const bool publish = b != g_plan_stall;
#else
    constexpr bool publish = true;
#endif
    if (tid < 3 && publish) __hip_atomic_store(&a.sync[1 + tid * g.B + b], s_red[tid] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // 3. exclusive prefix over the images in front of this one
    {
        int a3 = 0, a1 = 0, ap = 0, failed = 0;
        long long t_start = 0;
        for (int i = tid; i < b; i += NT) {
            int v3, v1, vp, n = 0;
            for (;;) {             // the three loads are in flight together (each is a trip to the coherence point)
                v3 = __hip_atomic_load(&a.sync[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v1 = __hip_atomic_load(&a.sync[1 + g.B + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vp = __hip_atomic_load(&a.sync[1 + 2 * g.B + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v3 != 0 && v1 != 0 && vp != 0) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++n & 255) == 0) {           // bounded in wall-clock time (a constant 100 MHz counter), looked at every 256 polls
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > a.timeout_ticks) { failed = 1; v3 = v1 = vp = 1; break; }
                }
            }
            a3 += v3 - 1; a1 += v1 - 1; ap += vp - 1;
        }
        if (failed) s_fail = 1;
        a3 = wave_isum(a3); a1 = wave_isum(a1); ap = wave_isum(ap);
        if (lane == 0) {
            atomicAdd(&s_base[0], a3);
            atomicAdd(&s_base[1], a1);
            atomicAdd(&s_base[2], ap);
        }
    }
    __syncthreads();
    // every word a consumer sizes its work by (cnt, stats, all prefixes) zeroed by one wave: the failure path (see the header comment)
    auto poison = [&]() {
        for (int i = lane; i <= g.B; i += 64) {
            __hip_atomic_store(&a.pre3[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.pre1[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < 2) __hip_atomic_store(&a.cnt[lane], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 3) __hip_atomic_store(reinterpret_cast<int*>(a.stats) + lane, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Leaving (the LAST thing a workgroup does, on every path): count out; the last workgroup out -- nobody reads a flag word any more -- zeroes
    // them all, so that the next launch on this buffer (stream order) finds them clean without a zeroing launch in front of it.
    auto leave = [&]() {
        __syncthreads();
        if (wave == 0) {
            int last = 0;
            if (lane == 0) last = __hip_atomic_fetch_add(&a.sync[1 + 3 * g.B], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.B - 1;
            if (__builtin_amdgcn_readfirstlane(last))
                for (int i = lane; i < 2 + 3 * g.B; i += 64) __hip_atomic_store(&a.sync[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (s_fail) {                   // (never observed outside the test hook: a predecessor that did not publish within the bound)
        if (wave == 0) {
            if (lane == 0) {        // flag first, then the zeroes
                __hip_atomic_store(&a.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&g_plan_timeouts, 1u);
                if (g_fault_dev) __hip_atomic_store(g_fault_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // loud: the caller's next check() raises
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            poison();
        }
        leave();
        return;
    }
    const int base3 = s_base[0], base1 = s_base[1], own3 = s_red[0];
    if (tid == 0) {                 // write-through (agent-scope) stores: acknowledged at the coherence point before the end check below
        __hip_atomic_store(&a.pre3[b], base3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.pre1[b], base1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b == g.B - 1) {
            const int tot3 = base3 + own3, tot1 = base1 + s_red[1];
            __hip_atomic_store(&a.pre3[g.B], tot3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.pre1[g.B], tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.cnt[0], tot3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.cnt[1], tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.stats[0], (float)(s_base[2] + s_red[2]) / (float)((long)g.B * SS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.stats[1], (float)tot3 / (float)((long)g.B * HWo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.stats[2], (float)tot1 / (float)((long)g.B * HWi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // 4. ordered compaction of the kept output pixels (row-major, or patch by patch) and of the dilated input pixels
    const int gy = g.Ho / g.S, gx = g.Wo / g.Sx, PP = gy * gx;
    const float inv_pp = 1.f / (float)max(PP, 1), inv_sx = 1.f / (float)g.Sx, inv_gx = 1.f / (float)max(gx, 1);
    int running = 0;
    for (int i0 = 0; i0 < HWo; i0 += NT) {
        const int e = i0 + tid;
        int i = e;
        if (a.patch_major && e < HWo) {
            const int q = idiv(e, PP, inv_pp), l = e - q * PP;
            const int py = idiv(q, g.Sx, inv_sx), px = q - py * g.Sx, ly = idiv(l, gx, inv_gx), lx = l - ly * gx;
            i = (py * gy + ly) * g.Wo + px * gx + lx;
        }
        const bool f = e < HWo && s_m3[i];
        int tot;
        const int r = block_rank_n<NW>(f, s_w, tot);
        if (e < HWo) {
            s_pos3[i] = f ? base3 + running + r : -1;
            if (f) s_list[running + r] = i;
        }
        running += tot;
    }
    running = 0;
    for (int i0 = 0; i0 < HWi; i0 += NT) {
        const int i = i0 + tid;
        const bool f = i < HWi && s_m1[i];
        int tot;
        const int r = block_rank_n<NW>(f, s_w, tot);
        if (i < HWi) {
            const int p = f ? base1 + running + r : -1;
            s_pos1[i] = p;
            a.pos1[(size_t)b * HWi + i] = p;
            LDN_DCHECK(!f || (p >= 0 && p < g.B * HWi), 202);
            if (f) a.idx1[p] = b * HWi + i;
        }
        running += tot;
    }
    __syncthreads();
    // 5. the lists, in packed order (coalesced; no thread idles on a dropped pixel)
    for (int i = tid; i < HWo; i += NT) a.pos3[(size_t)b * HWo + i] = s_pos3[i];
    for (int k = tid; k < own3; k += NT) {
        LDN_DCHECK(base3 + k < g.B * HWo, 201);
        a.idx3[base3 + k] = b * HWo + s_list[k];
    }
    for (int e = tid; e < own3 * 9; e += NT) {
        const int k = e / 9, t = e - 9 * k;
        const int i = s_list[k];
        const int oy = idiv(i, g.Wo, inv_wo), ox = i - oy * g.Wo;
        const int iy = oy * g.stride - 1 + t / 3, ix = ox * g.stride - 1 + t % 3;
        const bool inb = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
        LDN_DCHECK(!inb || s_pos1[iy * g.Wi + ix] >= 0, 203);
        a.nbr[(size_t)base3 * 9 + e] = inb ? s_pos1[iy * g.Wi + ix] : -1;
    }
    // 6. end check (wave 0, behind the acknowledgement of its own stores): did ANY workgroup of this launch fail its prefix wait?
    if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_load(&a.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) poison();
    }
    leave();
}

// ---- S == Sx == 1 (layer skip: one decision per image): every list is a closed form of the number of kept images in front of
// image b -- no count pass, no LDS tables, grid = chunks x B.  Same lists, counts, prefixes and statistics as k_mask_index.
__global__ __launch_bounds__(256) void k_layer_index(const float* __restrict__ patch, const IdxGeom g, int32_t* __restrict__ idx3,
                                                      int32_t* __restrict__ pos3, int32_t* __restrict__ idx1,
                                                      int32_t* __restrict__ pos1, int32_t* __restrict__ nbr,
                                                      int32_t* __restrict__ cnt, int32_t* __restrict__ pre3,
                                                      int32_t* __restrict__ pre1, float* __restrict__ stats, int tgy, int tgx) {
    __shared__ int s_cnt[2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int HWo = g.Ho * g.Wo, HWi = g.Hi * g.Wi;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = tid; i < g.B; i += 256) {
        const int k = patch[i] > 0.5f ? 1 : 0;
        total += k;
        before += i < b ? k : 0;
    }
    atomicAdd(&s_cnt[0], before);   // integer LDS atomics: order-independent
    atomicAdd(&s_cnt[1], total);
    __syncthreads();
    const int nb = s_cnt[0], tot = s_cnt[1];
    const bool kept = patch[b] > 0.5f;
    const int base3 = nb * HWo, base1 = nb * HWi;
    if (blockIdx.x == 0 && tid == 0) {
        pre3[b] = base3;
        pre1[b] = base1;
        if (b == g.B - 1) {
            const int tot3 = tot * HWo, tot1 = tot * HWi;
            pre3[g.B] = tot3;
            pre1[g.B] = tot1;
            cnt[0] = tot3;
            cnt[1] = tot1;
            stats[0] = (float)tot / (float)((long)g.B * g.S * g.Sx);
            stats[1] = (float)tot3 / (float)((long)g.B * HWo);
            stats[2] = (float)tot1 / (float)((long)g.B * HWi);
        }
    }
    const int step = gridDim.x * 256, first = blockIdx.x * 256 + tid;
    // tgy > 0: the image's pixels tile by tile (tgy x tgx pixels, row-major inside a tile, tiles row-major) instead of row-major --
    // tgy * tgx consecutive packed rows are then one tile (ldn_conv_rows_pool leaves the tiles' means: the fused layer masker)
    const int PP = tgy * tgx, Tx = tgy ? g.Wo / tgx : 1;
    auto pixel_of = [&](int i) {
        if (!tgy) return i;
        const int q = i / PP, l = i - q * PP;
        const int ty = q / Tx, tx = q - ty * Tx, ly = l / tgx, lx = l - ly * tgx;
        return (ty * tgy + ly) * g.Wo + tx * tgx + lx;
    };
    for (int i = first; i < HWo; i += step) {
        const int pix = pixel_of(i);
        pos3[(size_t)b * HWo + pix] = kept ? base3 + i : -1;
        if (kept) idx3[base3 + i] = b * HWo + pix;
    }
    for (int i = first; i < HWi; i += step) {
        pos1[(size_t)b * HWi + i] = kept ? base1 + i : -1;
        if (kept) idx1[base1 + i] = b * HWi + i;
    }
    if (!kept) return;
    for (int e = first; e < HWo * 9; e += step) {
        const int i = e / 9, t = e - 9 * i;
        const int pix = pixel_of(i);
        const int oy = pix / g.Wo, ox = pix - oy * g.Wo;
        const int iy = oy * g.stride - 1 + t / 3, ix = ox * g.stride - 1 + t % 3;
        const bool inb = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
        nbr[(size_t)base3 * 9 + e] = inb ? base1 + iy * g.Wi + ix : -1;
    }
}

// ---- the same lists built by BANDS of output rows (grid = B x bands) for maps whose per-image tables do not fit one workgroup's
// LDS (detection-size inputs).  Band j of image b owns the output rows [y0, y1) and the input rows [y0 s, y1 s);
// packed positions are row-major inside an image and images follow each other, so band (b, j) starts at the sum of the counts of
// all (image, band) pairs before it -- the lists are identical to the whole-image kernel's.  What a band needs from its
// neighbours it recomputes: two halo rows of mask3 on each side (mask1 of the halo input rows) and the packed positions of the
// input row just above / below its own rows (the 3x3 neighbour table): the row above ends right before the band's first
// position, the row below starts right after its last one.
struct BandGeom {
    int R, nb;        // output rows per band, bands per image
};

__device__ __forceinline__ void fill_mask3_rows(const float* __restrict__ patch, const IdxGeom g, int b, int ya, int yb,
                                                unsigned char* s_m3) {   // rows [ya, yb) -> s_m3[(y - ya) * Wo + x]; rows outside the map = 0
    const float sh = (float)g.S / (float)g.Ho, sw = (float)g.Sx / (float)g.Wo;
    const int n = (yb - ya) * g.Wo;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = ya + i / g.Wo, x = i % g.Wo;
        unsigned char v = 0;
        if (y >= 0 && y < g.Ho) {
            const int sy = nearest_src(y, sh, g.S), sx = nearest_src(x, sw, g.Sx);
            v = patch[((size_t)b * g.S + sy) * g.Sx + sx] > 0.5f ? 1 : 0;
        }
        s_m3[i] = v;
    }
}

// mask1 of input pixel (iy, ix) from the band-local mask3 rows starting at output row ya
__device__ __forceinline__ bool mask1_rows(const unsigned char* s_m3, const IdxGeom g, int ya, int iy, int ix) {
    const int s = g.stride;
    int oy0 = (iy - 1 + s - 1) / s, oy1 = (iy + 1) / s;
    if (iy - 1 < 0) oy0 = 0;
    int ox0 = (ix - 1 + s - 1) / s, ox1 = (ix + 1) / s;
    if (ix - 1 < 0) ox0 = 0;
    oy1 = oy1 < g.Ho - 1 ? oy1 : g.Ho - 1;
    ox1 = ox1 < g.Wo - 1 ? ox1 : g.Wo - 1;
    bool any = false;
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) any |= s_m3[(oy - ya) * g.Wo + ox] != 0;
    return any;
}

__global__ __launch_bounds__(256) void k_mask_count_band(const float* __restrict__ patch, const IdxGeom g, const BandGeom bg,
                                                          int32_t* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bytes[];
    unsigned char* s_m3 = s_bytes;
    __shared__ int s_red[3];
    const int b = blockIdx.x / bg.nb, j = blockIdx.x - b * bg.nb;
    const int y0 = j * bg.R, y1 = min(y0 + bg.R, g.Ho);
    const int ya = y0 - 2;
    if (threadIdx.x < 3) s_red[threadIdx.x] = 0;
    fill_mask3_rows(patch, g, b, ya, y1 + 2, s_m3);
    __syncthreads();
    int c3 = 0, c1 = 0, cp = 0;
    for (int i = threadIdx.x; i < (y1 - y0) * g.Wo; i += 256) c3 += s_m3[2 * g.Wo + i];
    const int i0 = y0 * g.stride * g.Wi, i1 = y1 * g.stride * g.Wi;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const int iy = i / g.Wi, ix = i - iy * g.Wi;
        c1 += mask1_rows(s_m3, g, ya, iy, ix) ? 1 : 0;
    }
    if (j == 0)
        for (int i = threadIdx.x; i < g.S * g.Sx; i += 256) cp += patch[(size_t)b * g.S * g.Sx + i] > 0.5f ? 1 : 0;
    atomicAdd(&s_red[0], c3);
    atomicAdd(&s_red[1], c1);
    atomicAdd(&s_red[2], cp);
    __syncthreads();
    const int nent = g.B * bg.nb;
    if (threadIdx.x < 3) work[threadIdx.x * nent + blockIdx.x] = s_red[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_mask_index_band(const float* __restrict__ patch, const IdxGeom g, const BandGeom bg,
                                                          const int32_t* __restrict__ work, int32_t* __restrict__ idx3,
                                                          int32_t* __restrict__ pos3, int32_t* __restrict__ idx1,
                                                          int32_t* __restrict__ pos1, int32_t* __restrict__ nbr,
                                                          int32_t* __restrict__ cnt, int32_t* __restrict__ pre3,
                                                          int32_t* __restrict__ pre1, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bytes[];
    const int HWo = g.Ho * g.Wo, HWi = g.Hi * g.Wi;
    const int b = blockIdx.x / bg.nb, j = blockIdx.x - b * bg.nb, tid = threadIdx.x;
    const int y0 = j * bg.R, y1 = min(y0 + bg.R, g.Ho);
    const int ya = y0 - 2;
    const int r0 = y0 * g.stride, r1 = y1 * g.stride;            // own input rows [r0, r1); s_pos1 covers [r0 - 1, r1 + 1)
    int* s_pos3 = reinterpret_cast<int*>(s_bytes);               // [(y1 - y0) * Wo]
    int* s_pos1 = s_pos3 + bg.R * g.Wo;                          // [(R s + 2) * Wi]
    unsigned char* s_m3 = reinterpret_cast<unsigned char*>(s_pos1 + (bg.R * g.stride + 2) * g.Wi);   // [(R + 4) * Wo]
    __shared__ int s_w[4];
    __shared__ int s_base[3];
    const int nent = g.B * bg.nb;
    if (tid < 3) s_base[tid] = 0;
    __syncthreads();
    {
        int a3 = 0, a1 = 0, ap = 0;
        const bool last = (int)blockIdx.x == nent - 1;
        for (int i = tid; i < (last ? nent : (int)blockIdx.x); i += 256) {
            const bool own = i == (int)blockIdx.x;   // only in the last block: totals of the patch count
            a3 += own ? 0 : work[i];
            a1 += own ? 0 : work[nent + i];
            ap += work[2 * nent + i];
        }
        atomicAdd(&s_base[0], a3);
        atomicAdd(&s_base[1], a1);
        atomicAdd(&s_base[2], ap);
    }
    fill_mask3_rows(patch, g, b, ya, y1 + 2, s_m3);
    __syncthreads();
    const int base3 = s_base[0], base1 = s_base[1];
    const int own1 = work[nent + blockIdx.x];                    // mask1 pixels of this band
    if (tid == 0) {
        if (j == 0) { pre3[b] = base3; pre1[b] = base1; }
        if ((int)blockIdx.x == nent - 1) {
            const int tot3 = base3 + work[blockIdx.x], tot1 = base1 + own1;
            pre3[g.B] = tot3;
            pre1[g.B] = tot1;
            cnt[0] = tot3;
            cnt[1] = tot1;
            stats[0] = (float)s_base[2] / (float)((long)g.B * g.S * g.Sx);
            stats[1] = (float)tot3 / (float)((long)g.B * HWo);
            stats[2] = (float)tot1 / (float)((long)g.B * HWi);
        }
    }
    // own output pixels
    int running = 0;
    const int n3 = (y1 - y0) * g.Wo;
    for (int i0 = 0; i0 < n3; i0 += 256) {
        const int i = i0 + tid;
        const bool f = i < n3 && s_m3[2 * g.Wo + i];
        int tot;
        const int r = block_rank(f, s_w, tot);
        if (i < n3) {
            const int pp = f ? base3 + running + r : -1;
            s_pos3[i] = pp;
            pos3[(size_t)b * HWo + y0 * g.Wo + i] = pp;
            LDN_DCHECK(!f || (pp >= 0 && pp < g.B * HWo), 201);
            if (f) idx3[pp] = b * HWo + y0 * g.Wo + i;
        }
        running += tot;
    }
    // own input pixels -> s_pos1 rows 1 .. (r1 - r0)
    running = 0;
    const int n1 = (r1 - r0) * g.Wi;
    for (int i0 = 0; i0 < n1; i0 += 256) {
        const int i = i0 + tid;
        bool f = false;
        if (i < n1) f = mask1_rows(s_m3, g, ya, r0 + i / g.Wi, i % g.Wi);
        int tot;
        const int r = block_rank(f, s_w, tot);
        if (i < n1) {
            const int pp = f ? base1 + running + r : -1;
            s_pos1[g.Wi + i] = pp;
            pos1[(size_t)b * HWi + (size_t)r0 * g.Wi + i] = pp;
            LDN_DCHECK(!f || (pp >= 0 && pp < g.B * HWi), 202);
            if (f) idx1[pp] = b * HWi + r0 * g.Wi + i;
        }
        running += tot;
    }
    // the input rows just above (ends right before base1) and just below (starts at base1 + own1) the band: positions only
    for (int side = 0; side < 2; ++side) {
        const int iy = side == 0 ? r0 - 1 : r1;
        int* dst = s_pos1 + (side == 0 ? 0 : (r1 - r0 + 1) * g.Wi);
        const bool inmap = iy >= 0 && iy < g.Hi;
        // total of the row first (the row above is addressed from its end)
        int rowtot = 0;
        if (side == 0 && inmap) {
            for (int i0 = 0; i0 < g.Wi; i0 += 256) {
                const int i = i0 + tid;
                const bool f = i < g.Wi && mask1_rows(s_m3, g, ya, iy, i);
                int tot;
                (void)block_rank(f, s_w, tot);
                rowtot += tot;
            }
        }
        running = 0;
        for (int i0 = 0; i0 < g.Wi; i0 += 256) {
            const int i = i0 + tid;
            const bool f = inmap && i < g.Wi && mask1_rows(s_m3, g, ya, iy, i);
            int tot;
            const int r = block_rank(f, s_w, tot);
            if (i < g.Wi) dst[i] = !f ? -1 : (side == 0 ? base1 - rowtot + running + r : base1 + own1 + running + r);
            running += tot;
        }
    }
    __syncthreads();
    for (int i = tid; i < n3; i += 256) {
        const int pp = s_pos3[i];
        if (pp < 0) continue;
        const int oy = y0 + i / g.Wo, ox = i % g.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy * g.stride - 1 + t / 3, ix = ox * g.stride - 1 + t % 3;
            const bool inb = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
            const int v = inb ? s_pos1[(iy - (r0 - 1)) * g.Wi + ix] : -1;
            LDN_DCHECK(!inb || v >= 0, 203);
            nbr[(size_t)pp * 9 + t] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------- K2 / K5
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, int ld_src,
                                                      const int32_t* __restrict__ rows,
                                                      const int32_t* __restrict__ count, int cap, int C4,
                                                      float* __restrict__ packed, int ld_packed) {
    const int n = count ? min(*count, cap) : cap;
    const long total = (long)n * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C4), q = (int)(i - (long)r * C4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)rows[r] * ld_src + q * 4);
        *reinterpret_cast<f32x4*>(packed + (size_t)r * ld_packed + q * 4) = v;
    }
}

__global__ __launch_bounds__(256) void k_scatter_add_relu(const float* __restrict__ packed, int ld_packed,
                                                           const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ count, int cap, int C4,
                                                           const float* identity, int ld_id, float* out,
                                                           int ld_out) {
    const int n = count ? min(*count, cap) : cap;
    const long total = (long)n * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C4), q = (int)(i - (long)r * C4);
        const size_t row = (size_t)rows[r];
        f32x4 v = *reinterpret_cast<const f32x4*>(packed + (size_t)r * ld_packed + q * 4);
        const f32x4 id = *reinterpret_cast<const f32x4*>(identity + row * ld_id + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + id[e], 0.f);
        *reinterpret_cast<f32x4*>(out + row * ld_out + q * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------- a2
// partial[b][s][c] = sum over rows of split s of x[b][row][c]   (deterministic two-stage GAP)
__global__ __launch_bounds__(256) void k_gap_partial(const float* __restrict__ x, int HW, int C, int splits,
                                                      float* __restrict__ partial, const int32_t* __restrict__ unchanged_prefix) {
    extern __shared__ __attribute__((aligned(16))) float s_f[];
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    // layer skip: an image the previous block skipped is unchanged -> its partial sums in `partial` (left by the previous call) stand
    if (unchanged_prefix && unchanged_prefix[b + 1] == unchanged_prefix[b]) return;
    const int Q = C >> 2;
    const int per = ceil_div(HW, splits);
    const int r0 = s * per, r1 = min(r0 + per, HW);
    const int RL = Q >= 256 ? 1 : 256 / Q;            // row lanes
    const int rl = Q >= 256 ? 0 : tid / Q;
    const float* xb = x + (size_t)b * HW * C;
    for (int q0 = 0; q0 < Q; q0 += 256) {
        const int q = q0 + (Q >= 256 ? tid : tid % Q);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (q < Q && rl < RL) {
            // eight rows in flight per lane (this kernel is one pass over x: it runs at the rate its loads are outstanding); they are
            // ADDED in row order, so the sums are the same as a row-by-row loop's
            const float* src = xb + q * 4;
            int r = r0 + rl;
            for (; r + 7 * RL < r1; r += 8 * RL) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(r + k * RL) * C));
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
            for (; r < r1; r += RL) acc += *reinterpret_cast<const f32x4*>(src + (size_t)r * C);
        }
        if (RL > 1) {
            __syncthreads();
            if (rl < RL) *reinterpret_cast<f32x4*>(s_f + ((size_t)rl * Q + q) * 4) = acc;
            __syncthreads();
            if (rl == 0) {
                for (int k = 1; k < RL; ++k) acc += *reinterpret_cast<const f32x4*>(s_f + ((size_t)k * Q + q) * 4);
            }
        }
        if (q < Q && rl == 0)
            *reinterpret_cast<f32x4*>(partial + ((size_t)b * splits + s) * C + q * 4) = acc;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_channel_mlp(const float* __restrict__ partial, int HW, int C, int splits,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      int hidden, int G, int gran, const float* __restrict__ mask_in,
                                                      float* __restrict__ mask, float* __restrict__ logits,
                                                      int32_t* __restrict__ ch_idx, int32_t* __restrict__ ch_cnt) {
    extern __shared__ __attribute__((aligned(16))) float s_f[];
    __shared__ int s_w[NT / 64];
    __shared__ __attribute__((aligned(16))) float s_part[NT == 1024 ? NT * 4 : 4];   // group sums of the many-split GAP (ldn_mlp.h)
    channel_mlp_body<NT>(blockIdx.x, partial, HW, C, splits, w1, b1, w2, b2, hidden, G, gran, mask_in, mask, logits, ch_idx, ch_cnt,
                         s_f, s_w, NT == 1024 ? s_part : nullptr);
}

// head of the whole-image spatial masker: one workgroup (four waves) per image; gap = sum of the split partials / HW, then 2g dots.
// A chain of dependent global reads per image: every thread keeps the partials of four splits and the 2g weights of its channel in
// flight; the four waves' sums are combined through LDS in wave order (deterministic).
__global__ __launch_bounds__(256) void k_spatial_head(const float* __restrict__ partial, int B, int HW, int C, int splits,
                                                       const float* __restrict__ w, const float* __restrict__ bias, int g,
                                                       float* __restrict__ mask, float* __restrict__ logits) {
    __shared__ float s_acc[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float inv = 1.f / (float)HW;
    const int G2 = 2 * g;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    for (int c = tid; c < C; c += 256) {
        float wv[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) wv[o] = o < G2 ? w[o * C + c] : 0.f;
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = partial[((size_t)b * splits + k + u) * C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; k < splits; ++k) s += partial[((size_t)b * splits + k) * C + c];
        s *= inv;
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] += wv[o] * s;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = wave_sum(acc[o]);
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) s_acc[wave][o] = acc[o];
    }
    __syncthreads();
    if (tid < g) {
        float lk = bias[tid], ld = bias[tid + g];
        float sk = 0.f, sd = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { sk += s_acc[wv][tid]; sd += s_acc[wv][tid + g]; }
        lk += sk; ld += sd;
        mask[(size_t)b * g + tid] = lk >= ld ? 1.f : 0.f;
        if (logits) {
            logits[(size_t)b * G2 + tid] = lk;
            logits[(size_t)b * G2 + g + tid] = ld;
        }
    }
}

// The layer-skip decision from many equal tile means per image (ldn_layer_head): pool [B][nparts][C] -> mean over the parts -> 1x1 conv.
// One workgroup per image; 16 bytes per lane, row lanes over the parts with eight loads in flight, fixed-order reductions.
__global__ __launch_bounds__(256) void k_layer_head(const float* __restrict__ pool, int nparts, int C, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int g, float* __restrict__ mask,
                                                     float* __restrict__ logits) {
    __shared__ f32x4 s_part[256];
    __shared__ float s_acc[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int Q = C >> 2, G2 = 2 * g;
    const int RL = Q >= 256 ? 1 : 256 / Q, rl = Q >= 256 ? 0 : tid / Q;
    const float inv = 1.f / (float)nparts;
    const float* pb = pool + (size_t)b * nparts * C;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    for (int q0 = 0; q0 < Q; q0 += 256) {
        const int q = q0 + (Q >= 256 ? tid : tid % Q);
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (q < Q && rl < RL) {
            const float* src = pb + q * 4;
            int r = rl;
            for (; r + 7 * RL < nparts; r += 8 * RL) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(r + k * RL) * C);
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += v[k];
            }
            for (; r < nparts; r += RL) sum += *reinterpret_cast<const f32x4*>(src + (size_t)r * C);
        }
        if (RL > 1) {
            __syncthreads();
            s_part[tid] = sum;
            __syncthreads();
            if (rl == 0)
                for (int k = 1; k < RL; ++k) sum += s_part[k * Q + tid];
        }
        if (q < Q && rl == 0) {
            sum *= inv;
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < G2) acc[o] += dot4(*reinterpret_cast<const f32x4*>(w + (size_t)o * C + q * 4), sum);
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = wave_sum(acc[o]);
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) s_acc[wave][o] = acc[o];
    }
    __syncthreads();
    if (tid < g) {
        float sk = 0.f, sd = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { sk += s_acc[wv][tid]; sd += s_acc[wv][tid + g]; }
        const float lk = bias[tid] + sk, ld = bias[tid + g] + sd;
        mask[(size_t)b * g + tid] = lk >= ld ? 1.f : 0.f;
        if (logits) {
            logits[(size_t)b * G2 + tid] = lk;
            logits[(size_t)b * G2 + g + tid] = ld;
        }
    }
}

LDN_DEFINE_TU_VIOLATIONS(tu_violations_index)

}  // namespace ldn

using namespace ldn;

extern "C" const char* ldn_last_error(void) { return g_err; }
extern "C" int ldn_version(void) { return 100; }
extern "C" int ldn_debug_violations(int* count, int* first_code, int reset) {
    LDN_REQUIRE(count != nullptr, "ldn_debug_violations: null pointer");
#ifndef LDN_DEBUG
    *count = -1;   // not a debug build: the checks are compiled away
    if (first_code) *first_code = 0;
    (void)reset;
    return LDN_OK;
#else
    if (hipDeviceSynchronize() != hipSuccess) { set_error("ldn_debug_violations: device error"); return LDN_EHIP; }
    unsigned c = 0, code = 0;
    int rc = tu_violations_conv(&c, &code, reset);
    if (!rc) rc = tu_violations_index(&c, &code, reset);
    if (!rc) rc = tu_violations_regnet(&c, &code, reset);
    if (!rc) rc = tu_violations_tail(&c, &code, reset);
    if (!rc) rc = tu_violations_dense(&c, &code, reset);
    if (!rc) rc = tu_violations_small(&c, &code, reset);
    if (!rc) rc = tu_violations_rows3(&c, &code, reset);
    if (rc) { set_error("ldn_debug_violations: cannot read the counters"); return rc; }
    *count = (int)c;
    if (first_code) *first_code = (int)code;
    return LDN_OK;
#endif
}

namespace ldn {
// One pinned, device-mapped int per process.  Kernels write 1 into it when a bounded wait fails; the host reads it without synchronising.
int* fault_word_host() {
    static int* word = [] {
        void* q = nullptr;
        if (hipHostMalloc(&q, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return static_cast<int*>(nullptr); }
        *static_cast<volatile int*>(q) = 0;
        return static_cast<int*>(q);
    }();
    return word;
}
int* fault_word_dev() {
    int* h = fault_word_host();
    void* d = nullptr;
    if (!h || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<int*>(d);
}
// once per device: the kernels of this translation unit learn the word's address
static void arm_fault_word_index(hipStream_t st) {
    static bool armed[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || armed[dev]) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;       // (not inside a graph capture: the symbol copy is a synchronous call)
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    int* d = fault_word_dev();
    if (d && hipMemcpyToSymbol(HIP_SYMBOL(g_fault_dev), &d, sizeof(d)) == hipSuccess) armed[dev] = true;
    else (void)hipGetLastError();
}
}  // namespace ldn

// The fault word (see include/ldn_hip.h): NULL when pinned memory could not be had (no device).
extern "C" const int* ldn_fault_flag(void) { return ldn::fault_word_host(); }

// Prefix waits of ldn_mask_plan / ldn_mask_to_index launches that ran into their time bound since the last reset (0 on a healthy
// device).  Such a launch leaves EMPTY lists (counts, prefixes and statistics zero), never uninitialised ones; this counter is how a
// caller finds out.  Synchronises the device.
extern "C" int ldn_plan_timeouts(int* count, int reset) {
    LDN_REQUIRE(count != nullptr, "ldn_plan_timeouts: null pointer");
    if (hipDeviceSynchronize() != hipSuccess) { set_error("ldn_plan_timeouts: device error"); return LDN_EHIP; }
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_plan_timeouts), sizeof(v)) != hipSuccess) { set_error("ldn_plan_timeouts: cannot read the counter"); return LDN_EHIP; }
    unsigned stalls = 0;      // + the loader / consumer hand-off waits of the chained kernel that ran into their bound (csrc/ldn_chain_ld.h)
    if (tu_chain_stalls(&stalls, reset) != LDN_OK) { set_error("ldn_plan_timeouts: cannot read the chained kernel's counter"); return LDN_EHIP; }
    *count = (int)(v + stalls);
    if (reset) {
        if (int* fw = fault_word_host()) *static_cast<volatile int*>(fw) = 0;      // re-arm the loud path
    }
    if (reset && v) {
        const unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_plan_timeouts), &z, sizeof(z)) != hipSuccess) { set_error("ldn_plan_timeouts: cannot reset the counter"); return LDN_EHIP; }
    }
    return LDN_OK;
}

#ifdef LDN_DEBUG
// test hook of the debug build: workgroup `image` of every following k_plan launch never publishes its counts, so the workgroups
// behind it run into the time bound (-1 = off)
extern "C" int ldn_debug_plan_stall(int image) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_plan_stall), &image, sizeof(image)) == hipSuccess ? LDN_OK : LDN_EHIP;
}
#endif

extern "C" int ldn_device_cus(int* cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("ldn_device_cus: no HIP device");
        return LDN_EHIP;
    }
    *cus = prop.multiProcessorCount;
    return LDN_OK;
}

extern "C" size_t ldn_spatial_masker_workspace_bytes(int B, int Hi, int Wi, int C, int S) {
    if (!(S < Hi)) return 0;             // one logit per pixel: nothing pooled, nothing to keep
    if (S > 1) return (size_t)B * S * S * C * sizeof(float);     // the patches' pooled channel means (optional: patch carry)
    return (size_t)B * ldn_channel_masker_splits(Hi * Wi) * C * sizeof(float);   // whole-image window (layer skip): two-stage reduction
}
constexpr size_t kWholeImageLds = 150 * 1024;   // per-image tables up to this size are built by one workgroup per image
constexpr size_t kBandLds = 64 * 1024;          // LDS budget of a band (several workgroups per CU)

// rows per band / bands per image / LDS bytes of the banded index build; false if not even one row fits
static bool index_bands(int Ho, int Wo, int stride, int* R, int* nb, size_t* lds) {
    const size_t Wi = (size_t)Wo * stride;
    auto bytes = [&](int r) { return (size_t)r * Wo * 4 + ((size_t)r * stride + 2) * Wi * 4 + (size_t)(r + 4) * Wo; };
    const char* env = getenv("LDN_INDEX_BAND_LDS");             // tests: a small budget forces many bands on small maps
    const size_t budget = env ? (size_t)atol(env) : kBandLds;
    int r = Ho;
    while (r > 1 && bytes(r) > budget) r = (r + 1) / 2;
    if (bytes(r) > 150 * 1024) return false;
    const int n = (Ho + r - 1) / r;
    r = (Ho + n - 1) / n;                        // balance the bands
    *R = r; *nb = (Ho + r - 1) / r; *lds = bytes(r);
    return true;
}

extern "C" size_t ldn_mask_to_index_workspace_bytes(int B, int Ho, int Wo, int stride) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || stride <= 0) return 0;
    int R = 0, nb = 1;
    size_t lds = 0;
    const size_t whole = (size_t)Ho * Wo * 5 + (size_t)Ho * stride * Wo * stride * 4;
    if ((whole > kWholeImageLds || getenv("LDN_INDEX_BANDS")) && !index_bands(Ho, Wo, stride, &R, &nb, &lds)) nb = 1;
    return ((size_t)3 * B * nb + 4) * sizeof(int32_t);     // (+ a spare word: the flag array of the one-launch build, k_plan, starts at work[1])
}
extern "C" size_t ldn_channel_masker_workspace_bytes(int B, int HW, int C) {
    return (size_t)B * ldn_channel_masker_splits(HW) * C * sizeof(float);
}

extern "C" int ldn_spatial_masker(const float* x, int B, int Hi, int Wi, int C, const float* w, const float* bias,
                                  int g, int S, float* mask, float* logits, float* work, const int32_t* carry_prefix,
                                  const float* carry_mask, void* stream) {
    LDN_REQUIRE(x && w && bias && mask, "ldn_spatial_masker: null pointer");
    LDN_REQUIRE(g >= 1 && g <= 4, "ldn_spatial_masker: mask groups must be 1..4 (got %d)", g);
    LDN_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0 && S > 0, "ldn_spatial_masker: bad shape");
    const bool pooled = S < Hi;
    if (pooled && S == 1 && Hi * Wi >= 32 && C % 4 == 0) {       // (>= 32 pixels: 7 x 7 maps included -- one wave per image walking 49 x 2048 values took 58 us)
        // layer-skip masks (mask_size 1): the pooling window is the whole image -> two-stage deterministic GAP over
        // many workgroups (one wave per image would walk 3136 pixels serially), then one wave per image for the head
        LDN_REQUIRE(work, "ldn_spatial_masker: mask_size 1 needs the work buffer (B*splits*C floats)");
        const int HW = Hi * Wi, splits = ldn_channel_masker_splits(HW), Q = C / 4;
        const size_t lds = Q >= 256 ? 0 : (size_t)(256 / Q) * Q * 4 * sizeof(float);
        hipLaunchKernelGGL(k_gap_partial, dim3(splits, B), dim3(256), lds, static_cast<hipStream_t>(stream), x, HW, C,
                           splits, work, carry_prefix);
        LDN_CHECK_LAUNCH("k_gap_partial");
        hipLaunchKernelGGL(k_spatial_head, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), work, B, HW,
                           C, splits, w, bias, g, mask, logits);
        LDN_CHECK_LAUNCH("k_spatial_head");
        return LDN_OK;
    }
    if (!pooled && g <= 2 && C % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && !getenv("LDN_PIXEL_MASKER_OLD")) {
        // per-pixel masks: several pixels' rows in flight per wave, one shared reduction (k_pixel_masker; same logits and decisions as the
        // general kernel; the patch carry does not exist at this granularity: there are no pooled means to keep)
        hipStream_t st_ = static_cast<hipStream_t>(stream);
        if (g == 1) {
            const long jobs8 = (long)B * ceil_div(Hi * Wi, 8);
            hipLaunchKernelGGL((k_pixel_masker<8, 2>), dim3((unsigned)((jobs8 + 3) / 4)), dim3(256), 0, st_, x, B, Hi * Wi, C, w, bias, mask, logits);
        } else {
            const long jobs4 = (long)B * ceil_div(Hi * Wi, 4);
            hipLaunchKernelGGL((k_pixel_masker<4, 4>), dim3((unsigned)((jobs4 + 3) / 4)), dim3(256), 0, st_, x, B, Hi * Wi, C, w, bias, mask, logits);
        }
        LDN_CHECK_LAUNCH("k_pixel_masker");
        return LDN_OK;
    }
    const long jobs = (long)B * (pooled ? S * S : Hi * Wi);
    LDN_REQUIRE(!carry_mask || work, "ldn_spatial_masker: carry_mask needs the work buffer of the previous call");
    // The pooling bins floor(i*H/S) .. ceil((i+1)*H/S) overlap when H % S != 0 while the previous block's conv3 writes pixels by the
    // nearest mapping floor(y*S/H): a bin can then hold pixels of a NEIGHBOURING patch that block did rewrite, so an untouched patch's
    // stored means are stale.  The carry is only exact on even grids.
    LDN_REQUIRE(!carry_mask || !pooled || (Hi % S == 0 && Wi % S == 0),
                "ldn_spatial_masker: the patch carry needs H %% S == 0 and W %% S == 0 (got %dx%d, S=%d)", Hi, Wi, S);
    LDN_REQUIRE((uintptr_t)work % 16 == 0, "ldn_spatial_masker: work must be 16-byte aligned");
    hipLaunchKernelGGL(k_spatial_masker, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, B, Hi, Wi, C, w, bias, g, S, mask, logits, pooled && S > 1 ? work : nullptr,
                       pooled && S > 1 ? carry_mask : nullptr);
    LDN_CHECK_LAUNCH("k_spatial_masker");
    return LDN_OK;
}

// cell means on a 2S x 2S grid -> the means of the S x S grid of 2 x 2 groups of them (equal cells: the mean of the four means); the head of a
// stage whose masker pools coarser cells than its predecessor's (LAUD-ResNet spatial 4-4-2-1: 7 x 7 cells of 8 x 8 pixels behind 14 x 14 of 4 x 4)
__global__ __launch_bounds__(256) void k_coarsen_cells(const float* __restrict__ fine, int S, int C4, float* __restrict__ coarse, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int x = (int)(r % S); r /= S;
        const int y = (int)(r % S);
        const long b = r / S;
        const f32x4* f = reinterpret_cast<const f32x4*>(fine) + ((b * 2 * S + 2 * y) * 2 * S + 2 * x) * C4 + c;
        const f32x4 v = ((f[0] + f[C4]) + (f[(long)2 * S * C4] + f[(long)2 * S * C4 + C4])) * 0.25f;
        reinterpret_cast<f32x4*>(coarse)[i] = v;
    }
}

extern "C" int ldn_coarsen_cell_means(const float* fine, int B, int S, int C, float* coarse, void* stream) {
    LDN_REQUIRE(fine && coarse, "ldn_coarsen_cell_means: null pointer");
    LDN_REQUIRE(B > 0 && S > 0 && C > 0 && C % 4 == 0, "ldn_coarsen_cell_means: bad shape (C must be a multiple of 4)");
    LDN_REQUIRE((uintptr_t)fine % 16 == 0 && (uintptr_t)coarse % 16 == 0, "ldn_coarsen_cell_means: pointers must be 16-byte aligned");
    const long total = (long)B * S * S * (C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_coarsen_cells, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), fine, S, C / 4, coarse, total);
    LDN_CHECK_LAUNCH("k_coarsen_cells");
    return LDN_OK;
}

// one launch (k_plan): patch decisions (given, or taken from pooled channel means) -> counts, prefixes, lists
static size_t plan_lds(int S, int Sx, int Ho, int Wo, int stride) {
    return (size_t)Ho * Wo * 9 + (size_t)Ho * stride * Wo * stride * 5 + (size_t)round_up(S * Sx, 16);
}
// The caller's word that the `work` buffer of the NEXT one-launch list build of this thread is all zero (a buffer it zeroed once and has only
// ever handed to such builds on one stream: every build leaves it zero) -- the zeroing launch in front of k_plan is then skipped.  Consumed by
// the next ldn_mask_plan / ldn_mask_to_index call of the thread whatever path it takes.
static thread_local int g_plan_work_zero = 0;
extern "C" int ldn_plan_work_zeroed(int yes) { g_plan_work_zero = yes ? 1 : 0; return LDN_OK; }
//  A caller that vouched for a zeroed `work` gets it back zeroed WHATEVER path the call took: the two-launch and banded builds
// leave their counts in it, so they clear it behind themselves -- the caller's mirror of the library's path choice (environment switches included)
// no longer has to be exact for a shared, zeroed-once buffer to stay valid.
static int rezero_vouched_work(bool vouched, int32_t* work, int B, int Ho, int Wo, int stride, hipStream_t st) {
    if (!vouched) return LDN_OK;
    const int n = (int)(ldn_mask_to_index_workspace_bytes(B, Ho, Wo, stride) / sizeof(int32_t));
    hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, work, n);
    LDN_CHECK_LAUNCH("k_zero_i32");
    return LDN_OK;
}

static int launch_plan(PlanArgs& a, int32_t* work, hipStream_t st, bool work_is_zero) {
    const IdxGeom& g = a.g;
    const size_t lds = plan_lds(g.S, g.Sx, g.Ho, g.Wo, g.stride);
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_plan), lds), "k_plan: cannot reserve %zu B of LDS", lds);
    a.sync = work;
    arm_fault_word_index(st);
    // bound of the prefix wait: 2 s by default (a predecessor workgroup of the SAME launch publishes within microseconds of its start;
    // only a queue pre-empted for that long could exceed it); LDN_PLAN_TIMEOUT_MS overrides (tests)
    static const long timeout_ms = getenv("LDN_PLAN_TIMEOUT_MS") ? atol(getenv("LDN_PLAN_TIMEOUT_MS")) : 2000;
    a.timeout_ticks = (long long)timeout_ms * 100000ll;
    // the flag words are zeroed by a kernel, not hipMemsetAsync: a memset node inside a captured hipGraph faulted on the second replay
    // of the graph (ROCm 7.2, measured: tools/experiments/repro_graph.py), a kernel node replays fine
    const int nsync = 3 * g.B + 2;
    if (!work_is_zero) {
        hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)ceil_div(nsync, 256)), dim3(256), 0, st, work, nsync);
        LDN_CHECK_LAUNCH("k_zero_i32");
    }
    hipLaunchKernelGGL(k_plan, dim3((unsigned)g.B), dim3(kPlanThreads), lds, st, a);
    LDN_CHECK_LAUNCH("k_plan");
    return LDN_OK;
}

extern "C" int ldn_mask_plan_fits(int S, int Sx, int Ho, int Wo, int stride) {
    return S > 0 && Sx > 0 && Ho > 0 && Wo > 0 && stride >= 1 && plan_lds(S, Sx, Ho, Wo, stride) <= kWholeImageLds ? 1 : 0;
}

extern "C" int ldn_mask_plan(const float* patch_mask, const float* pool, int C, const float* w, const float* bias, float* mask_out,
                             float* logits, int B, int S, int Sx, int Ho, int Wo, int stride, int patch_major, int32_t* idx3,
                             int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt, int32_t* img_prefix3,
                             int32_t* img_prefix1, float* stats, int32_t* work, void* stream) {
    const bool work_zero = g_plan_work_zero != 0;     // (consumed by this call whatever happens below)
    g_plan_work_zero = 0;
    LDN_REQUIRE(idx3 && pos3 && idx1 && pos1 && nbr && cnt && img_prefix3 && img_prefix1 && stats && work, "ldn_mask_plan: null pointer");
    LDN_REQUIRE(B > 0 && S > 0 && Sx > 0 && Ho > 0 && Wo > 0 && stride >= 1, "ldn_mask_plan: bad shape");
    LDN_REQUIRE((long)B * Ho * stride * Wo * stride < (1l << 31), "ldn_mask_plan: index space exceeds int32");
    LDN_REQUIRE(patch_mask || (pool && w && bias && mask_out && C > 0 && C % 4 == 0),
                "ldn_mask_plan: either the patch mask, or pooled means + masker weights (C a multiple of 4) + the mask output");
    LDN_REQUIRE(patch_mask || ((uintptr_t)pool % 16 == 0 && (uintptr_t)w % 16 == 0), "ldn_mask_plan: pool / w must be 16-byte aligned");
    LDN_REQUIRE(!patch_major || (Ho % S == 0 && Wo % Sx == 0), "ldn_mask_plan: patch-major lists need an even grid (%dx%d map, %dx%d patches)", Ho, Wo, S, Sx);
    LDN_REQUIRE(ldn_mask_plan_fits(S, Sx, Ho, Wo, stride), "ldn_mask_plan: the per-image tables of a %dx%d map exceed one workgroup's LDS (use ldn_mask_to_index)", Ho * stride, Wo * stride);
    PlanArgs a{IdxGeom{B, S, Sx, Ho, Wo, stride, Ho * stride, Wo * stride}, patch_major ? 1 : 0, patch_mask, pool, C, w, bias, mask_out, logits,
               nullptr, idx3, pos3, idx1, pos1, nbr, cnt, img_prefix3, img_prefix1, stats};
    return launch_plan(a, work, static_cast<hipStream_t>(stream), work_zero);
}

// layer skip (one decision per image) with the kept images' pixels listed TILE BY TILE (tile_gy x tile_gx pixels; 0, 0 = row-major,
// i.e. ldn_mask_to_index with S = 1) -- the lists of the fused layer masker
extern "C" int ldn_layer_index(const float* image_mask, int B, int Ho, int Wo, int stride, int tile_gy, int tile_gx, int32_t* idx3,
                               int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt, int32_t* img_prefix3,
                               int32_t* img_prefix1, float* stats, void* stream) {
    LDN_REQUIRE(image_mask && idx3 && pos3 && idx1 && pos1 && nbr && cnt && img_prefix3 && img_prefix1 && stats, "ldn_layer_index: null pointer");
    LDN_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && stride >= 1, "ldn_layer_index: bad shape");
    LDN_REQUIRE((long)B * Ho * stride * Wo * stride < (1l << 31), "ldn_layer_index: index space exceeds int32");
    LDN_REQUIRE((tile_gy == 0 && tile_gx == 0) || (tile_gy > 0 && tile_gx > 0 && Ho % tile_gy == 0 && Wo % tile_gx == 0),
                "ldn_layer_index: the %dx%d map must split evenly into %dx%d tiles", Ho, Wo, tile_gy, tile_gx);
    IdxGeom g{B, 1, 1, Ho, Wo, stride, Ho * stride, Wo * stride};
    const long items = (long)Ho * Wo * 9 > (long)g.Hi * g.Wi ? (long)Ho * Wo * 9 : (long)g.Hi * g.Wi;
    const unsigned chunks = (unsigned)(items / 2048 < 1 ? 1 : (items / 2048 > 32 ? 32 : items / 2048));
    hipLaunchKernelGGL(k_layer_index, dim3(chunks, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), image_mask, g, idx3, pos3,
                       idx1, pos1, nbr, cnt, img_prefix3, img_prefix1, stats, tile_gy, tile_gx);
    LDN_CHECK_LAUNCH("k_layer_index");
    return LDN_OK;
}

// the layer-skip decision from pooled TILE MEANS (models/utils.py:47-65 with mask_size 1): pool [B][nparts][C] holds the channel means
// of nparts equal tiles of every image (ldn_spatial_masker's patch means with S*S = nparts, refreshed by ldn_conv_rows_pool), the global
// average is their mean; mask [B][g], logits [B][2g] optional.  x is not read.
extern "C" int ldn_layer_head(const float* pool, int B, int nparts, int C, const float* w, const float* bias, int g, float* mask,
                              float* logits, void* stream) {
    LDN_REQUIRE(pool && w && bias && mask, "ldn_layer_head: null pointer");
    LDN_REQUIRE(B > 0 && nparts > 0 && C > 0 && g >= 1 && g <= 4, "ldn_layer_head: bad shape");
    if (C % 4 == 0 && (C / 4 >= 256 || 256 % (C / 4) == 0) && (uintptr_t)pool % 16 == 0 && (uintptr_t)w % 16 == 0) {
        hipLaunchKernelGGL(k_layer_head, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), pool, nparts, C, w, bias, g, mask, logits);
        LDN_CHECK_LAUNCH("k_layer_head");
        return LDN_OK;
    }
    hipLaunchKernelGGL(k_spatial_head, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), pool, B, nparts, C, nparts, w, bias, g, mask, logits);
    LDN_CHECK_LAUNCH("k_spatial_head");
    return LDN_OK;
}

extern "C" int ldn_mask_to_index(const float* patch_mask, int B, int S, int Sx, int Ho, int Wo, int stride, int32_t* idx3,
                                 int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt,
                                 int32_t* img_prefix3, int32_t* img_prefix1, float* stats, int32_t* work,
                                 void* stream) {
    const bool work_zero = g_plan_work_zero != 0;     // (consumed by this call whatever path it takes)
    g_plan_work_zero = 0;
    LDN_REQUIRE(patch_mask && idx3 && pos3 && idx1 && pos1 && nbr && cnt && img_prefix3 && img_prefix1 && stats && work,
                "ldn_mask_to_index: null pointer");
    LDN_REQUIRE(B > 0 && S > 0 && Sx > 0 && Ho > 0 && Wo > 0 && stride >= 1, "ldn_mask_to_index: bad shape");
    LDN_REQUIRE((long)B * Ho * stride * Wo * stride < (1l << 31), "ldn_mask_to_index: index space exceeds int32");
    IdxGeom g{B, S, Sx, Ho, Wo, stride, Ho * stride, Wo * stride};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (S == 1 && Sx == 1 && !getenv("LDN_INDEX_GENERIC") && !getenv("LDN_INDEX_BANDS")) {   // one decision per image (layer skip): closed-form lists, one launch
        const long items = (long)Ho * Wo * 9 > (long)g.Hi * g.Wi ? (long)Ho * Wo * 9 : (long)g.Hi * g.Wi;
        const unsigned chunks = (unsigned)(items / 2048 < 1 ? 1 : (items / 2048 > 32 ? 32 : items / 2048));
        hipLaunchKernelGGL(k_layer_index, dim3(chunks, (unsigned)B), dim3(256), 0, st, patch_mask, g, idx3, pos3, idx1, pos1, nbr, cnt,
                           img_prefix3, img_prefix1, stats, 0, 0);
        LDN_CHECK_LAUNCH("k_layer_index");
        return LDN_OK;
    }
    const char* plan_env = getenv("LDN_INDEX_PLAN");                 // "0": the two-launch build (A/B, tests); read per call
    const bool use_plan = !(plan_env && atoi(plan_env) == 0);
    if (use_plan && !getenv("LDN_INDEX_BANDS") && ldn_mask_plan_fits(S, Sx, Ho, Wo, stride)) {   // whole image in one workgroup's LDS: one launch
        PlanArgs a{g, 0, patch_mask, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, idx3, pos3, idx1, pos1, nbr, cnt,
                   img_prefix3, img_prefix1, stats};
        return launch_plan(a, work, st, work_zero);
    }
    const size_t lds1 = (size_t)Ho * Wo;
    const size_t lds2 = (size_t)Ho * Wo * 5 + (size_t)g.Hi * g.Wi * 4;
    if (lds2 <= kWholeImageLds && !getenv("LDN_INDEX_BANDS")) {   // whole image in one workgroup's LDS
        LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_mask_index), lds2), "k_mask_index: cannot reserve %zu B of LDS", lds2);
        hipLaunchKernelGGL(k_mask_count, dim3(B), dim3(256), lds1, st, patch_mask, g, work);
        LDN_CHECK_LAUNCH("k_mask_count");
        hipLaunchKernelGGL(k_mask_index, dim3(B), dim3(256), lds2, st, patch_mask, g, work, idx3, pos3, idx1, pos1, nbr,
                           cnt, img_prefix3, img_prefix1, stats);
        LDN_CHECK_LAUNCH("k_mask_index");
        return rezero_vouched_work(work_zero, work, B, Ho, Wo, stride, st);
    }
    // bands of output rows (large / detection-size maps)
    BandGeom bg{};
    size_t ldsb = 0;
    LDN_REQUIRE(index_bands(Ho, Wo, stride, &bg.R, &bg.nb, &ldsb), "ldn_mask_to_index: a row of the %dx%d map does not fit the LDS", g.Hi, g.Wi);
    LDN_REQUIRE((long)B * bg.nb < (1l << 24), "ldn_mask_to_index: too many bands");
    LDN_REQUIRE(allow_dynamic_lds(reinterpret_cast<const void*>(&k_mask_index_band), ldsb), "k_mask_index_band: cannot reserve %zu B of LDS", ldsb);
    hipLaunchKernelGGL(k_mask_count_band, dim3((unsigned)B * bg.nb), dim3(256), (size_t)(bg.R + 4) * Wo, st, patch_mask, g, bg, work);
    LDN_CHECK_LAUNCH("k_mask_count_band");
    hipLaunchKernelGGL(k_mask_index_band, dim3((unsigned)B * bg.nb), dim3(256), ldsb, st, patch_mask, g, bg, work, idx3, pos3, idx1,
                       pos1, nbr, cnt, img_prefix3, img_prefix1, stats);
    LDN_CHECK_LAUNCH("k_mask_index_band");
    return rezero_vouched_work(work_zero, work, B, Ho, Wo, stride, st);
}

static unsigned stream_grid(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int ldn_gather_rows(const float* src, int ld_src, const int32_t* rows, const int32_t* count, int cap,
                               int C, float* packed, int ld_packed, void* stream) {
    LDN_REQUIRE(src && rows && packed, "ldn_gather_rows: null pointer");
    LDN_REQUIRE(C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_packed % 4 == 0, "ldn_gather_rows: C and strides must be multiples of 4");
    if (cap <= 0) return LDN_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(stream_grid((long)cap * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, ld_src, rows, count, cap, C / 4, packed, ld_packed);
    LDN_CHECK_LAUNCH("k_gather_rows");
    return LDN_OK;
}

extern "C" int ldn_scatter_add_relu(const float* packed, int ld_packed, const int32_t* rows, const int32_t* count,
                                    int cap, int C, const float* identity, int ld_id, float* out, int ld_out,
                                    void* stream) {
    LDN_REQUIRE(packed && rows && identity && out, "ldn_scatter_add_relu: null pointer");
    LDN_REQUIRE(C > 0 && C % 4 == 0 && ld_packed % 4 == 0 && ld_id % 4 == 0 && ld_out % 4 == 0,
                "ldn_scatter_add_relu: C and strides must be multiples of 4");
    if (cap <= 0) return LDN_OK;
    hipLaunchKernelGGL(k_scatter_add_relu, dim3(stream_grid((long)cap * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), packed, ld_packed, rows, count, cap, C / 4, identity, ld_id,
                       out, ld_out);
    LDN_CHECK_LAUNCH("k_scatter_add_relu");
    return LDN_OK;
}

namespace ldn {
// FLOPs bookkeeping of a whole forward in ONE launch (laud_resnet.py:112-147 per block, :321-356 for the static parts): per block j
//   sparse_j = t[j][0] + t[j][1] cs s1 + t[j][2] cs^2 s2 + t[j][3] cs s3 + t[j][4],   perc_j = sparse_j / sum_i t[j][i],
//   flops = sum_j sparse_j + static          (fp64 inside, as laudnet_amd's flops_from_sparsities; summed in block order)
// with (s3, s2, s1, cs) = st_in[j] where given, and cs = sum_b cnt[j][b] / denom[j] for channel-mode blocks (denom[j] > 0).
__global__ __launch_bounds__(1024) void k_forward_stats(const int32_t* cnt, int B, const float* denom, const float* st_in, int st_cols,
                                                       const double* terms, double static_flops, int n, float* st_out, float* perc,
                                                       float* flops) {
    __shared__ double s_sparse[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < n; j += 16) {         // one wave per block (sixteen waves: the per-block chains of dependent loads overlap)
        float s3 = 1.f, s2 = 1.f, s1 = 1.f, cs = 1.f;
        if (st_in) {
            s3 = st_in[j * st_cols]; s2 = st_in[j * st_cols + 1]; s1 = st_in[j * st_cols + 2];
            if (st_cols > 3) cs = st_in[j * st_cols + 3];
        }
        if (cnt && denom && denom[j] > 0.f) {
            long acc = 0;                        // integer sum: exact, order-free
            for (int b = lane; b < B; b += 64) acc += cnt[(size_t)j * B + b];
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            cs = (float)acc / denom[j];          // = torch: cnt.sum().float() / denom  (sum exact in int64, one fp32 division)
        }
        if (lane == 0) {
            const double* t = terms + (size_t)j * 5;
            const double dcs = (double)cs;
            double sp = t[0] + t[1] * dcs * (double)s1;
            sp = sp + t[2] * (dcs * dcs) * (double)s2;
            sp = sp + t[3] * dcs * (double)s3;
            sp = sp + t[4];
            const double tot = (((t[0] + t[1]) + t[2]) + t[3]) + t[4];
            s_sparse[j] = sp;
            perc[j] = (float)(sp / tot);
            st_out[j * 4] = s3; st_out[j * 4 + 1] = s2; st_out[j * 4 + 2] = s1; st_out[j * 4 + 3] = cs;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double f = 0.0;
        for (int j = 0; j < n; ++j) f += s_sparse[j];
        flops[0] = (float)(f + static_flops);
    }
}
}  // namespace ldn

extern "C" int ldn_forward_stats(const int32_t* cnt, int B, const float* denom, const float* st_in, int st_cols, const double* terms,
                                 double static_flops, int n_blocks, float* st_out, float* perc, float* flops, void* stream) {
    LDN_REQUIRE(terms && st_out && perc && flops, "ldn_forward_stats: null pointer");
    LDN_REQUIRE(n_blocks > 0 && n_blocks <= 512, "ldn_forward_stats: 1 .. 512 blocks (got %d)", n_blocks);
    LDN_REQUIRE((cnt == nullptr) == (denom == nullptr) && (!cnt || B > 0), "ldn_forward_stats: cnt and denom go together");
    LDN_REQUIRE(!st_in || st_cols == 3 || st_cols == 4, "ldn_forward_stats: st_in has 3 or 4 columns");
    hipLaunchKernelGGL(ldn::k_forward_stats, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), cnt, B, denom, st_in, st_cols, terms,
                       static_flops, n_blocks, st_out, perc, flops);
    LDN_CHECK_LAUNCH("k_forward_stats");
    return LDN_OK;
}

extern "C" int ldn_channel_masker_splits(int HW) {
    int s = HW / 256;
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    return s;
}

extern "C" int ldn_channel_masker(const float* x, int B, int HW, int C, const float* w1, const float* b1,
                                  const float* w2, const float* b2, int hidden, int G, int gran, const float* mask_in,
                                  float* mask, float* logits, int32_t* ch_idx, int32_t* ch_cnt, float* work,
                                  const float* gap_partial, int gap_splits, void* stream) {
    LDN_REQUIRE(mask && ch_idx && ch_cnt, "ldn_channel_masker: null output pointer");
    LDN_REQUIRE(B > 0 && G > 0 && gran > 0, "ldn_channel_masker: bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int splits = ldn_channel_masker_splits(HW);
    if (!mask_in && gap_partial) {
        // the producer of x (ldn_conv_image with colsum) already left per-subtile channel sums: no pass over x
        LDN_REQUIRE(w1 && b1 && gap_splits > 0 && HW > 0 && C > 0, "ldn_channel_masker: bad fused-GAP arguments");
        LDN_REQUIRE(hidden == 0 || (w2 && b2), "ldn_channel_masker: two-layer masker needs w2/b2");
        splits = gap_splits;
        work = const_cast<float*>(gap_partial);
    } else if (!mask_in) {
        LDN_REQUIRE(x && w1 && b1 && work, "ldn_channel_masker: null pointer");
        LDN_REQUIRE(hidden == 0 || (w2 && b2), "ldn_channel_masker: two-layer masker needs w2/b2");
        LDN_REQUIRE(HW > 0 && C > 0 && C % 4 == 0, "ldn_channel_masker: C must be a positive multiple of 4");
        const int Q = C / 4;
        const size_t lds = Q >= 256 ? 0 : (size_t)(256 / Q) * Q * 4 * sizeof(float);
        hipLaunchKernelGGL(k_gap_partial, dim3(splits, B), dim3(256), lds, st, x, HW, C, splits, work, (const int32_t*)nullptr);
        LDN_CHECK_LAUNCH("k_gap_partial");
    }
    const size_t lds2 = (size_t)(C + (hidden > 0 ? hidden : 1) + 2 * G) * sizeof(float);
    // one block per image; 16 waves when there is an MLP to evaluate (its phases are latency chains), 4 for a pure list build
    if (mask_in) hipLaunchKernelGGL(k_channel_mlp<256>, dim3(B), dim3(256), lds2, st, work, HW, C, splits, w1, b1, w2, b2, hidden, G, gran,
                                    mask_in, mask, logits, ch_idx, ch_cnt);
    else hipLaunchKernelGGL(k_channel_mlp<1024>, dim3(B), dim3(1024), lds2, st, work, HW, C, splits, w1, b1, w2, b2, hidden, G, gran,
                       mask_in, mask, logits, ch_idx, ch_cnt);
    LDN_CHECK_LAUNCH("k_channel_mlp");
    return LDN_OK;
}
