// k_chain_ld -- the chained stage kernel (k_chain, csrc/ldn_tail.hip) re-decomposed as a LOADER / CONSUMER pipeline (round 6).
//
// What round 5's traces said about k_chain: every phase of a block runs at 2-3 x its matrix time because all eight waves walk the same
// barrier-to-barrier chain per K chunk -- wait for the chunk's LDS-DMA, s_barrier, issue the next chunk's DMA (index look-ups, address arithmetic,
// M0 set-ups), read fragments, multiply -- so the two waves of a SIMD multiply together and then idle the matrix pipe together.  A 14 x 14 map is 196
// pixels = seven 32-pixel wave tiles: the EIGHTH wave of the workgroup has no pixels.  Here it becomes the workgroup's loader:
//   * wave 7 issues ALL LDS-DMA of the operands the waves share (the image's gathered W1 rows, W2 tiles, h1 slices, W3 chunks) into rings of 3-4
//     slots, from per-lane source offsets it computes once per block / K slice, and publishes "chunks landed" in an LDS word behind its own vmcnt;
//   * waves 0-6 (the consumers) never issue a shared DMA, never look an index up and never meet a workgroup barrier inside a phase: they poll the
//     landed word (normally already ahead), multiply, and post their own progress word, which the loader polls before it refills a slot.  Nothing
//     re-synchronises the two waves of a SIMD chunk by chunk, so they drift out of phase and one wave's fragment reads / epilogue run under the
//     other's MFMAs;
//   * conv1's activation rows stay private to the wave that owns the pixels (its own DMA, its own vmcnt, ring of 2-3 K32 steps: DESIGN.md 4u).
// Same products in the same order per accumulator as head_body / tail_body: results are bit-identical to k_chain and to the block-by-block kernels
// (tests/test_hip_chain.py).  Every spin is bounded: a wait that runs into its bound is counted (ldn_plan_timeouts) and poisons nothing but the
// values -- the launch always terminates.
//
// Included by ldn_tail.hip behind head_body / tail_body (their DMA / fragment helpers are used as they are).
#pragma once

namespace ldn {

#ifndef LDN_LD_PRIO
#define LDN_LD_PRIO 0        // tuning: s_setprio of the younger consumer waves (4-6); the loader gets 3
#endif
#ifndef LDN_LD_R2MAX
#define LDN_LD_R2MAX 4       // tuning: deepest W2 ring
#endif
#ifndef LDN_LD3_ABLATE
#define LDN_LD3_ABLATE 0      // tuning only (wrong results): conv3 of the chained kernel without 1 = its whole epilogue, 2 = the residual loads and output stores only, 4 = its MFMAs
#endif
#ifndef LDN_LD3_DEFER
#define LDN_LD3_DEFER 0        // conv3's per-chunk epilogue deferred into the next chunk's K loop (0 = epilogue behind its own K loop)
#endif
#ifndef LDN_LD3_STORE_STEP
#define LDN_LD3_STORE_STEP 1     // K step (n-subtile) of the next chunk behind which a deferred epilogue waits for its residual tile and stores
#endif
#ifndef LDN_LD2_STAGED
#define LDN_LD2_STAGED 0     // conv2's weight tiles through the loader's registers as well: measured SLOWER -- 45 tiles of 20 KB per block are more than one wave can shuffle (4.3 k cycles per tile against the consumers' 2.8 k: conv2 149 k -> 195 k cycles per block); conv2 keeps LDS-DMA into the dense pair layout + per-wave shuffles
#endif
#ifndef LDN_LD3_STAGED
#define LDN_LD3_STAGED 1     // conv3's weights through the loader's registers, shuffled once into fragment order (0 = LDS-DMA + per-wave shuffles: A/B measurements)
#endif

constexpr int LD_SYNC_OFF = T_KIDX_BYTES;         // the sync words live behind the channel list in every phase
constexpr int LD_SYNC_BYTES = 256;
constexpr int LD_LOADER = 7;                      // the wave without pixels
constexpr int LD_SPIN_LIMIT = 1 << 20;            // polls (each >= ~150 cycles with its s_sleep): ~0.1 s

__device__ unsigned g_ld_stalls = 0u;             // hand-off waits that ran into LD_SPIN_LIMIT since the last reset
__device__ int* g_ld_fault_dev = nullptr;         // the process's host-visible fault word (ldn_fault_flag), set by launch_chain once per device
#ifdef LDN_TRACE   // tuning only (tools/trace_chain.py): [B][8 waves][8] cycles summed over the run -- consumers: conv1 loop, conv1 epilogue, conv2 loop,
                   // table build + conversion, conv3 loop, waits for the loader (all phases); loader: the three streams, waits for the consumers
__device__ unsigned long long* g_ld_trace = nullptr;
#define LT(x) x = __builtin_amdgcn_s_memtime();
#define LD_TIMED(slot, stmt) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); stmt; q.t[slot] += __builtin_amdgcn_s_memtime() - t0_; }
#define LD_SPAN(slot, a, b) q.t[slot] += (b) - (a);
#else
#define LT(x)
#define LD_TIMED(slot, stmt) { stmt; }
#define LD_SPAN(slot, a, b)
#endif

struct LdSync {                                   // LDS, 256 bytes
    unsigned landed;                              // chunks (sequence numbers < landed) whose LDS-DMA has landed; written by the loader only
    unsigned pad0[15];
    unsigned done[16];                            // done[w] = sequence number + 1 of the last chunk consumer wave w has finished READING
    unsigned pad1[32];
};
static_assert(sizeof(LdSync) == LD_SYNC_BYTES, "LdSync layout");

// per-wave state of the hand-off protocol (wave-uniform)
struct LdSeq {
    unsigned base;                                // sequence number of the next phase's first chunk (same arithmetic in every wave)
    unsigned seen;                                // consumers: the last value of sync->landed this wave has read
#ifdef LDN_TRACE
    unsigned long long t[8];
#endif
};

// ds_read_b32 of a sync word with its wait in ONE statement (hipcc neither counts nor waits for an asm load: cdna_hip_programming.md 5.7)
__device__ __forceinline__ unsigned ld_lds_read(const unsigned* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_off(p)) : "memory");
    return v;
}
__device__ __forceinline__ void ld_lds_write(unsigned* p, unsigned v) {
    asm volatile("ds_write_b32 %0, %1\n\ts_nop 0" ::"v"(lds_off(p)), "v"(v) : "memory");
}

// consumer: chunk `need - 1` has landed (LDS is one in-order memory: the loader's ds_write of the word sits behind its covering vmcnt, this wave's
// fragment reads sit behind its read of the word)
__device__ __forceinline__ void ld_wait_landed(LdSync* sy, unsigned need, unsigned& seen) {
    if (seen >= need) return;
    for (int spin = 0;; ++spin) {
        seen = __builtin_amdgcn_readfirstlane(ld_lds_read(&sy->landed));
        if (seen >= need) break;
        if (spin >= LD_SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&g_ld_stalls, 1u);
                if (g_ld_fault_dev) __hip_atomic_store(g_ld_fault_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            seen = 0x7fffffffu;                   // sticky: this wave stops waiting (values are lost, the launch terminates)
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// consumer: every read of chunk seq's slot has been ISSUED (LDS executes a wave's operations in order: the word becomes visible behind them)
__device__ __forceinline__ void ld_post_done(LdSync* sy, int wave, unsigned seq_plus_1) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) ld_lds_write(&sy->done[wave], seq_plus_1);
    asm volatile("" ::: "memory");
}
// loader: every consumer wave has finished chunk need - 1
__device__ __forceinline__ void ld_wait_done(LdSync* sy, int ncomp, unsigned need, bool& dead) {
    if (dead) return;
    const int lane = threadIdx.x & 63;
    for (int spin = 0;; ++spin) {
        const unsigned v = ld_lds_read(&sy->done[lane & 15]);
        if (__ballot(lane < ncomp && v < need) == 0ull) break;
        if (spin >= LD_SPIN_LIMIT) {
            if (lane == 0) {
                atomicAdd(&g_ld_stalls, 1u);
                if (g_ld_fault_dev) __hip_atomic_store(g_ld_fault_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
#ifdef LDN_DEBUG
__device__ int g_chain_stall = -1;             // test hook (ldn_debug_chain_stall): the loader of this image's workgroup never publishes "landed"
#endif
__device__ __forceinline__ void ld_publish(LdSync* sy, unsigned landed) {
#ifdef LDN_DEBUG
    if ((int)blockIdx.x == g_chain_stall) return;      // (the DMA itself still runs: consumers time out ONCE, stop waiting, read whatever has landed)
#endif
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) ld_lds_write(&sy->landed, landed);
    asm volatile("" ::: "memory");
}

// counted wait with a run-time, wave-uniform count (0 .. 63: the counter has six bits)
__device__ __forceinline__ void wait_vm_rt63(int n) {
#define LDN_WV(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        LDN_WV(0) LDN_WV(1) LDN_WV(2) LDN_WV(3) LDN_WV(4) LDN_WV(5) LDN_WV(6) LDN_WV(7) LDN_WV(8) LDN_WV(9) LDN_WV(10) LDN_WV(11) LDN_WV(12)
        LDN_WV(13) LDN_WV(14) LDN_WV(15) LDN_WV(16) LDN_WV(17) LDN_WV(18) LDN_WV(19) LDN_WV(20) LDN_WV(21) LDN_WV(22) LDN_WV(23) LDN_WV(24)
        LDN_WV(25) LDN_WV(26) LDN_WV(27) LDN_WV(28) LDN_WV(29) LDN_WV(30) LDN_WV(31) LDN_WV(32) LDN_WV(33) LDN_WV(34) LDN_WV(35) LDN_WV(36)
        LDN_WV(37) LDN_WV(38) LDN_WV(39) LDN_WV(40) LDN_WV(41) LDN_WV(42) LDN_WV(43) LDN_WV(44) LDN_WV(45) LDN_WV(46) LDN_WV(47) LDN_WV(48)
        LDN_WV(49) LDN_WV(50) LDN_WV(51) LDN_WV(52) LDN_WV(53) LDN_WV(54) LDN_WV(55) LDN_WV(56) LDN_WV(57) LDN_WV(58) LDN_WV(59) LDN_WV(60)
        LDN_WV(61) LDN_WV(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef LDN_WV
}

// 16 bytes from sbase (wave-uniform) + off (per lane): a GLOBAL load in the saddr + 32-bit offset form (left as a generic pointer, hipcc emits
// flat loads behind 64-bit address arithmetic -- two address registers per load, 64 of them in a staged tile)
__device__ __forceinline__ u32x4 ld_global16(const unsigned char* sbase, unsigned off) {
    typedef const __attribute__((address_space(1))) u32x4* gptr;
    return *(gptr)(sbase + off);
}

__device__ __forceinline__ f32x4 ld_global_f4(const float* ptr) {
    typedef const __attribute__((address_space(1))) f32x4* gptr;
    return *(gptr)(ptr);
}

// One LDS-DMA piece (1 KB: 16 bytes per lane) from sbase (wave-uniform) + vo (per-lane byte offset) to LDS lds_base + lane * 16.
__device__ __forceinline__ void ld_dma1(unsigned vo, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(lds_base) : "memory");
}
// n (wave-uniform, a multiple of 4, <= MAXP) consecutive 1 KB pieces of a slot: piece i from sbase + vo[i] to lds_base + i * 1024.  Groups of four
// share one M0 set-up (dma16_pieces: the instruction offset moves source AND destination, so vo[i] carries the bias (3 - i % 4) * 1024 against
// sbase - 3072: ld_bias()).  All indices are compile-time: the offsets stay in registers.
__device__ __forceinline__ unsigned ld_bias(int i) { return (unsigned)(3 - (i & 3)) * 1024u; }
template <int MAXP>
__device__ __forceinline__ void ld_dma_run(const unsigned (&vo)[MAXP], int n, const unsigned char* sbase, unsigned lds_base) {
    const void* sb = uniform_cptr(sbase - 3072);
#pragma unroll
    for (int g = 0; g < MAXP / 4; ++g) {
        if (4 * g >= n) break;
        const unsigned v4[4] = {vo[4 * g], vo[4 * g + 1], vo[4 * g + 2], vo[4 * g + 3]};
        dma16_pieces<4>(v4, sb, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)g * 4096u)));
    }
}

// ================================================================================================================ conv1
// head_ld: conv1 of block i on the loader / consumer structure.  LDS: [channel list 1280 | sync 256 | sc1 sh1 ps1 3 W floats | weight ring RW x
// (8 ceil(Nb / 8) rows x 128 B) | x rings: consumer wave w owns DX x 4 KB].  A weight slot holds the image's gathered rows of one K32 chunk in list
// order (row n at n * 128, 16-byte units XOR-swizzled by (n >> 1) & 7 on the source side); an x slot the wave's 32 pixels of the chunk.
template <int NS>
__device__ __forceinline__ void head_ld(const HeadArgs& p, const int b, unsigned char* const smem, const int lds_total, const int tid, LdSeq& q) {
    constexpr int W = NS * 32;
    constexpr int MAXP = NS * 4;                                          // weight pieces of a chunk when the image keeps every channel
    int* const s_nidx = reinterpret_cast<int*>(smem);                    // [W + 32]
    LdSync* const sy = reinterpret_cast<LdSync*>(smem + LD_SYNC_OFF);
    float* const s_tab = reinterpret_cast<float*>(smem + LD_SYNC_OFF + LD_SYNC_BYTES);
    unsigned char* const s_ring = smem + LD_SYNC_OFF + LD_SYNC_BYTES + 3 * W * 4;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int npix = p.HW;                                                // whole image (<= 224 pixels)
    const long row0 = (long)b * p.HW;
    const int ncomp = ceil_div(npix, 32);

    const int Nb = min(p.n_cnt[b], W);
    const int nsub = __builtin_amdgcn_readfirstlane(ceil_div(Nb, 32));
    const int nwp = __builtin_amdgcn_readfirstlane(4 * max(nsub, 1));             // weight pieces (8 rows each) per chunk: whole n-subtiles, rows beyond the list re-read its last channel
    const int wslot = nwp * 1024;
    if (tid < W + 32) s_nidx[tid] = tid < Nb ? p.n_idx[(size_t)b * W + tid] : -1;
    if (tid < LD_SYNC_BYTES / 4) reinterpret_cast<unsigned*>(sy)[tid] = 0u;       // the block's sequence numbers start at 0
    __syncthreads();
    for (int i = tid; i < 3 * W; i += 512) {
        const int k = i / W, n = i - k * W;
        const int ch = s_nidx[n];
        const float* src = k == 0 ? p.sc1 : (k == 1 ? p.sh1 : p.ps1);
        s_tab[i] = ch >= 0 ? src[ch] : 0.f;
    }
    // ring depths: three x steps per wave where the weight ring still gets three slots, else two; weight ring 2..4 slots
    const int avail = lds_total - (LD_SYNC_OFF + LD_SYNC_BYTES + 3 * W * 4);
    int DX = 3;
    if ((avail - ncomp * DX * 4096) / wslot < 3) DX = 2;
    DX = __builtin_amdgcn_readfirstlane(DX);
    const int RW = __builtin_amdgcn_readfirstlane(min(4, (avail - ncomp * DX * 4096) / wslot));      // >= 2 for W <= 256 (ldn_bottleneck_chain_fits)
    unsigned char* const s_x = s_ring + RW * wslot;
    const int nchunks = p.cin / 32;
    const unsigned base = q.base;
    q.base = base + (unsigned)nchunks;
#ifdef LDN_TRACE
    unsigned long long ta, tb, tc;
    LT(ta)
#endif

    if (wave == LD_LOADER) {
        // ---- loader: per-lane source offsets of the image's weight pieces (fixed for the block), then the stream
        if (nsub == 0) return;
        unsigned wo[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int rw = 8 * i + (lane >> 3);
            const int ch = s_nidx[min(rw, max(Nb, 1) - 1)];                // rows beyond the list re-read its last channel (zero epilogue tables)
            wo[i] = (unsigned)(max(ch, 0) * p.cin * 4 + (((lane & 7) ^ ((rw >> 1) & 7)) << 4)) + ld_bias(i);
        }
        const unsigned lds_w = lds_off(s_ring);
        bool dead = false;
        int slot = 0;
        for (int c = 0; c < nchunks; ++c) {
            if (c >= RW) LD_TIMED(5, ld_wait_done(sy, ncomp, base + (unsigned)(c - RW + 1), dead))
            ld_dma_run<MAXP>(wo, nwp, p.w1s + (long)c * 128, lds_w + (unsigned)slot * (unsigned)wslot);
            slot = slot + 1 == RW ? 0 : slot + 1;
            if (c > 0) { wait_vm_rt63(nwp); ld_publish(sy, base + (unsigned)c); }      // chunks < c have landed
        }
        wait_vm_n<0>();
        ld_publish(sy, base + (unsigned)nchunks);
        LT(tb)
        LD_SPAN(0, ta, tb)
        return;
    }
    if (wave >= ncomp) return;

    // ---- consumer
    const unsigned lds_x = lds_off(s_x) + (unsigned)wave * (unsigned)(DX * 4096);
    unsigned char* const my_x = s_x + wave * DX * 4096;
    unsigned xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 8 * i + (lane >> 3);                                // row of the wave's tile
        const int r = wave * 32 + rl;
        xo[i] = (unsigned)((min(r, npix - 1) * p.ldx) * 4 + (((lane & 7) ^ ((rl >> 1) & 7)) << 4)) + ld_bias(i);
    }
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(p.x + row0 * p.ldx) - 3072;
    auto dma_x = [&](int c, int xs) {      // chunk c (beyond the K range: the last one again -- keeps the counted wait's arithmetic constant) into x slot xs
        const int cc = min(c, nchunks - 1);
        dma16_pieces<4>(xo, uniform_cptr(xbase + (long)cc * 128), (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_x + (unsigned)xs * 4096u)));
    };

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if (nsub == 0) return;                  // (wave-uniform per image; the loader returned as well) no active channel: no h1 column is read downstream
    for (int c = 0; c < DX - 1; ++c) dma_x(c, c);
    const unsigned xsw = ((unsigned)l31 >> 1) & 7u;
    bf16x8 bh[2], bl[2];
    int xs = 0, ws_i = 0;                   // x slot / weight slot of chunk c
    for (int c = 0; c < nchunks; ++c) {
        wait_vm_rt(4 * (DX - 2));           // this wave's x rows of chunk c have landed
        {   // B operands of both K16 steps (the wave's 32 pixels), split once for all of the image's n-subtiles
            const unsigned char* xsl = my_x + xs * 4096 + l31 * 128;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned sl = 4u * half + 2u * h;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xsl + ((sl ^ xsw) << 4));
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(xsl + (((sl + 1) ^ xsw) << 4));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? x0[e] : x1[e - 4];
                    const __bf16 hb = (__bf16)v;
                    bh[half][e] = hb;
                    bl[half][e] = (__bf16)(v - (float)hb);
                }
            }
        }
        // the x slot of chunk c - 1 (read and split one iteration ago) takes chunk c + DX - 1
        dma_x(c + DX - 1, xs == 0 ? DX - 1 : xs - 1);
        LD_TIMED(5, ld_wait_landed(sy, base + (unsigned)c + 1u, q.seen))
        const unsigned char* ws = s_ring + ws_i * wslot + l31 * 128;
        // n-subtiles in DESCENDING order, software-pipelined as in head_body (weight fragment = two ds_read_b128, double-buffered in a0 / a1)
        bf16x8 a0h, a0l, a1h, a1l;
        const unsigned sl0 = 2u * h, sl1 = 4u + 2u * h;
        auto frag0 = [&](int j) {
            a0h = *reinterpret_cast<const bf16x8*>(ws + j * 4096 + ((sl0 ^ xsw) << 4));
            a0l = *reinterpret_cast<const bf16x8*>(ws + j * 4096 + (((sl0 + 1) ^ xsw) << 4));
        };
        auto frag1 = [&](int j) {
            a1h = *reinterpret_cast<const bf16x8*>(ws + j * 4096 + ((sl1 ^ xsw) << 4));
            a1l = *reinterpret_cast<const bf16x8*>(ws + j * 4096 + (((sl1 + 1) ^ xsw) << 4));
        };
#define LDN_HEADLD_STEP(J)                                                                                \
        frag1(J);                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        LDN_K16(false, acc[J], a0h, a0l, bh[0], bl[0])                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (J > 0) frag0(J > 0 ? J - 1 : 0);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        LDN_K16(false, acc[J], a1h, a1l, bh[1], bl[1])                                                    \
        __builtin_amdgcn_sched_barrier(0);
        if (nsub > 0) {
            frag0(nsub - 1);
            switch (nsub) {
                default:
                    if constexpr (NS >= 8) { LDN_HEADLD_STEP(7) }
                    [[fallthrough]];
                case 7:
                    if constexpr (NS >= 8) { LDN_HEADLD_STEP(6) }
                    [[fallthrough]];
                case 6:
                    if constexpr (NS >= 8) { LDN_HEADLD_STEP(5) }
                    [[fallthrough]];
                case 5:
                    if constexpr (NS >= 8) { LDN_HEADLD_STEP(4) }
                    [[fallthrough]];
                case 4:
                    if constexpr (NS >= 4) { LDN_HEADLD_STEP(3) }
                    [[fallthrough]];
                case 3:
                    if constexpr (NS >= 4) { LDN_HEADLD_STEP(2) }
                    [[fallthrough]];
                case 2:
                    LDN_HEADLD_STEP(1)
                    [[fallthrough]];
                case 1:
                    LDN_HEADLD_STEP(0)
            }
        }
#undef LDN_HEADLD_STEP
        ld_post_done(sy, wave, base + (unsigned)c + 1u);
        xs = xs + 1 == DX ? 0 : xs + 1;
        ws_i = ws_i + 1 == RW ? 0 : ws_i + 1;
    }
    wait_vm_n<0>();      // no LDS-DMA of this wave may be in flight when the phase's LDS is handed on
    LT(tb)
    LD_SPAN(0, ta, tb)

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
    LT(tc)
    LD_SPAN(1, tb, tc)
}

// ================================================================================================================ conv2 -> conv3
// tail_ld: the fused tail (stride 1, whole image per workgroup) on the loader / consumer structure.
// conv2 LDS: [channel list | sync | h1 slices 2 x slice_bytes | W2 ring R2 x (16 k-pair rows x Kp * 8 B)] -- a W2 slot is DENSE (row length = the
//            image's Kp / 2 channel pairs x 16 B instead of the layer's W / 2): 20 DMA pieces per chunk at Kp = 160 instead of 32, four slots
//            instead of three.
// conv3 LDS: [channel list | sync | W3 ring R3 x (Kp / 2 k-pair rows x 32 channels x 8 B) | ... | conversion tables 18 W floats | 7 x 4 KB transpose
//            scratch] (tables and scratch at the top of the workgroup's LDS).
constexpr int LD_CW = 32;                         // output channels per conv3 chunk
template <int NS>
__device__ __forceinline__ void tail_ld(const TailArgs& p, const int b, unsigned char* const smem, const int lds_total, const int tid, LdSeq& q) {
    constexpr int W = NS * 32;
    constexpr int MAXP = NS * 4;                      // pieces of a W2 / W3 chunk when the image keeps every channel (Kp / 8)
    constexpr int MAXH = 28;                          // pieces of an h1 slice (<= 224 pixels, 8 per piece)
    constexpr int NP = W;
    constexpr int W3_ROW = LD_CW * 8;
    int* const s_kidx = reinterpret_cast<int*>(smem);
    LdSync* const sy = reinterpret_cast<LdSync*>(smem + LD_SYNC_OFF);
    unsigned char* const s_lo = smem + LD_SYNC_OFF + LD_SYNC_BYTES;
    unsigned char* const s_h1 = s_lo;                                     // conv2: 2 slice slots ...
    unsigned char* const s_w2 = s_h1 + 2 * p.slice_bytes;                 // ... and the W2 ring
    unsigned char* const s_w3 = s_lo;                                     // conv3: the W3 ring ...
    unsigned char* const s_scr = smem + lds_total - 7 * 4096;             // ... the transpose scratch and the conversion tables at the top
    float* const s_tab = reinterpret_cast<float*>(s_scr - 18 * NP * 4);

    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int l31 = lane & 31, h = lane >> 5;
    const int npix = p.Ho * p.Wo;                                         // stride 1, whole image: <= 224
    const int ncomp = ceil_div(npix, 32);
    const int NR = npix, NRp = round_up(NR, 8);
    const int ZR = NRp;
    const long pix0 = (long)b * npix;

    const int Kb = min(p.k_cnt[b], W);
    const int nsub = __builtin_amdgcn_readfirstlane(ceil_div(Kb, 32));
    const int Kp = nsub * 32;
    const int P2 = __builtin_amdgcn_readfirstlane(Kp / 8);                // DMA pieces of a W2 chunk == of a W3 chunk
    const int slot2 = Kp * 128, slot3 = Kp * 128;
    if (tid < W + 32) s_kidx[tid] = tid < Kb ? p.k_idx[(size_t)b * W + tid] : -1;
    if (tid < 64) reinterpret_cast<float*>(s_h1 + (tid >> 5) * p.slice_bytes + ZR * 128)[tid & 31] = 0.f;      // the slices' zero rows
    __syncthreads();

    const int avail2 = lds_total - (LD_SYNC_OFF + LD_SYNC_BYTES) - 2 * p.slice_bytes;
    const int R2 = __builtin_amdgcn_readfirstlane(nsub > 0 ? min(LDN_LD_R2MAX, avail2 / slot2) : LDN_LD_R2MAX);          // >= 3 for a 224-pixel map of width 256
    const int avail3 = lds_total - (LD_SYNC_OFF + LD_SYNC_BYTES) - 18 * NP * 4 - 7 * 4096;
    const int R3 = __builtin_amdgcn_readfirstlane(nsub > 0 ? min(4, avail3 / slot3) : 4);
    const int nchunks = nsub * 9;
    const int nchunk3 = p.cout / LD_CW;
    const int nq = NRp / 8;
    const unsigned base2 = q.base, base3 = base2 + (unsigned)nchunks;
    q.base = base3 + (unsigned)nchunk3;
    constexpr int TMIN = 3;                           // the pieces of slice s + 1 go out with taps TMIN .. 8 of slice s (every reader has left slice s - 1 by then: R2 <= 4)
    static_assert(TMIN == 3 && MAXH == 28, "the slice schedule below is written out for taps 3 .. 8 and <= 28 pieces");

    const bool loader = wave == LD_LOADER;
    const bool consumer = wave < ncomp;
#ifdef LDN_TRACE
    unsigned long long ta, tb, tc, td;
    LT(ta)
#endif
    bool dead = false;

    // ======================================================================================================== conv2 (3x3)
    f32x16 acc[NS];
    if (loader) {
#if LDN_LD2_STAGED
        // W2 tiles through the loader's registers (as conv3's chunks below: the shuffle into fragment order happens ONCE, not in seven waves).  A tile =
        // 16 k-pair rows x Kp / 2 channel pairs of 16 B = 2 Kp pieces = nsub per lane, staged in HALVES of two groups each: item i (< nsub / 2
        // rounded up) of a lane = e = lane + 64 i over (group-in-half gh = e / (Kp / 2), packed channel pair v = e % (Kp / 2)); half hf holds groups
        // 2 hf + gh (K16 half hf of lane half gh: rows 8 hf + 4 gh + {0 .. 3}).  One register buffer per half: while a half is shuffled and written,
        // the other half's sixteen-byte loads are in flight -- a whole tile of loads is always outstanding.  Slot layout
        // [group][hi | lo plane][Kp channels][16 B]: a consumer fragment is one conflict-free ds_read_b128.
        constexpr int NH = NS / 2;                    // items per lane and half (all channels kept)
        const int nhi = (nsub + 1) / 2;               // ... for this image
        unsigned colo[NH];                            // source offset of the item's channel pair
        unsigned dsto[NH];                            // destination offset in the slot's half (a multiple of 32) | gh;  ~0u = no such item (odd nsub)
        const int hp = max(Kp / 2, 1);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = 64 * i + lane;
            const int gh = e / hp, v = e - gh * hp;
            const int ch = 2 * v < Kb ? s_kidx[2 * v] : 0;               // columns beyond the list fetch pair 0 (their accumulator columns meet zero tables)
            colo[i] = (unsigned)((ch >> 1) * 16);
            dsto[i] = gh < 2 ? ((unsigned)(gh * Kp * 32 + v * 32) | (unsigned)gh) : ~0u;
        }
        unsigned ro[2][NH][4];                        // this slice's source offsets per half: rows through the channel list (rows beyond it meet zero h1 columns)
        auto slice_rows = [&](int s_) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int gh = (int)(dsto[i] & 1u);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int kch = s_kidx[32 * s_ + 2 * (8 * hf + 4 * gh + qq)];
                        ro[hf][i][qq] = (unsigned)((max(kch, 0) >> 1) * (W / 2) * 16) + colo[i];
                    }
                }
        };
        u32x4 LA[NH][4], LB[NH][4];                   // the two halves' pieces
        auto load_half = [&](int t_, int hf, u32x4 (&L_)[NH][4]) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(uniform_cptr(p.w2p + (long)t_ * ((W / 2) * (W / 2) * 16)));
#pragma unroll
            for (int i = 0; i < NH; ++i)
                if (i < nhi)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) L_[i][qq] = ld_global16(src, ro[hf][i][qq]);
        };
        auto store_half = [&](unsigned char* slotp, int hf, const u32x4 (&L_)[NH][4]) {
            const int lo = Kp * 16;
            unsigned char* hb = slotp + hf * (Kp * 64);
#pragma unroll
            for (int i = 0; i < NH; ++i)
                if (i < nhi && dsto[i] != ~0u) {
                    unsigned char* d = hb + (dsto[i] & ~31u);
                    *reinterpret_cast<u32x4*>(d) = u32x4{L_[i][0][0], L_[i][1][0], L_[i][2][0], L_[i][3][0]};             // channel 2 v: hi quad
                    *reinterpret_cast<u32x4*>(d + lo) = u32x4{L_[i][0][1], L_[i][1][1], L_[i][2][1], L_[i][3][1]};        //              lo quad
                    *reinterpret_cast<u32x4*>(d + 16) = u32x4{L_[i][0][2], L_[i][1][2], L_[i][2][2], L_[i][3][2]};        // channel 2 v + 1
                    *reinterpret_cast<u32x4*>(d + lo + 16) = u32x4{L_[i][0][3], L_[i][1][3], L_[i][2][3], L_[i][3][3]};
                }
        };
        const unsigned lds_h1 = lds_off(s_h1);
        const unsigned char* const h1b = p.h1 + pix0 * p.h1_row_bytes;
        // h1 slice piece i: rows 8 i + (lane >> 3), physical 16-byte slot lane & 7 (XOR swizzle on the source side)
        auto h1_piece = [&](int i, const void* sb, unsigned dst) {
            const int r = 8 * i + (lane >> 3);
            const unsigned o = (unsigned)(min(r, NR - 1) * (int)p.h1_row_bytes + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
            ld_dma1(o, sb, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)i * 1024u)));
        };
        if (nchunks > 0) {
            const void* sb0 = uniform_cptr(h1b);
            for (int i = 0; i < nq; ++i) h1_piece(i, sb0, lds_h1);       // slice 0 -> slot 0
            slice_rows(0);
            load_half(0, 0, LA);
            load_half(0, 1, LB);
        }
        int slot = 0;
        int sn = 0, tn = 1;                           // slice / tap of chunk c + 1
        for (int c = 0; c < nchunks; ++c) {
            const int sc = c / 9, tc = c - 9 * sc;
            const bool nxt = c + 1 < nchunks;
            if (c >= R2) LD_TIMED(6, ld_wait_done(sy, ncomp, base2 + (unsigned)(c - R2 + 1), dead))
            if (tc == 0) wait_vm_n<0>();              // slice sc has landed (its pieces went out with taps TMIN .. 8 of the slice before)
            unsigned char* const slotp = s_w2 + slot * slot2;
            slot = slot + 1 == R2 ? 0 : slot + 1;
            store_half(slotp, 0, LA);                 // (waits for half 0's pieces only: half 1's and nothing younger may still fly)
            if (nxt) {
                if (tn == 0) slice_rows(sn);          // (both halves' rows of the next slice: ro[1] of THIS chunk has been consumed by its loads)
                load_half(tn, 0, LA);
            }
            store_half(slotp, 1, LB);
            ld_publish(sy, base2 + (unsigned)c + 1u);                  // (LDS executes this wave's writes in order: the word lands behind the quads)
            if (nxt) load_half(tn, 1, LB);
            if (tc >= TMIN && sc + 1 < nsub) {        // slice sc + 1: piece i goes out with tap TMIN + i % 6 (every reader has left slice sc - 1 by then: R2 <= 4)
                const void* hsb = uniform_cptr(h1b + (long)(sc + 1) * 128);
                const unsigned hdst = lds_h1 + (unsigned)((sc + 1) & 1) * (unsigned)p.slice_bytes;
                for (int i = tc - TMIN; i < nq; i += 6) h1_piece(i, hsb, hdst);
            }
            if (++tn == 9) { tn = 0; ++sn; }
        }
#else
        unsigned ho[MAXH];                            // h1 slice piece i: rows 8 i + (lane >> 3), physical 16-byte slot lane & 7
#pragma unroll
        for (int i = 0; i < MAXH; ++i) {
            const int r = 8 * i + (lane >> 3);
            ho[i] = (unsigned)(min(r, NR - 1) * (int)p.h1_row_bytes + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
        }
        // W2 piece i covers bytes [1024 i, 1024 i + 1024) of the dense slot: k-pair row u = e / (Kp / 2), packed n-pair v = e % (Kp / 2), e = 64 i + lane
        unsigned un[MAXP];                            // u of the lane's piece-i element
        unsigned no[MAXP];                            // its n-pair's source offset (+ the group bias)
        const int hp = max(Kp / 2, 1);
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int e = 64 * i + lane;
            const int u = min(e / hp, 15), v = e - (e / hp) * hp;
            const int ch = 2 * v < Kb ? s_kidx[2 * v] : 0;               // columns beyond the list fetch pair 0 (their accumulator columns meet zero tables)
            un[i] = (unsigned)u;
            no[i] = (unsigned)((ch >> 1) * 16) + ld_bias(i);
        }
        const unsigned lds_h1 = lds_off(s_h1), lds_w2 = lds_off(s_w2);
        const unsigned char* const h1b = p.h1 + pix0 * p.h1_row_bytes;
        if (nchunks > 0) {
#pragma unroll
            for (int i = 0; i < MAXH; ++i)
                if (i < nq) ld_dma1(ho[i], uniform_cptr(h1b), (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_h1 + (unsigned)i * 1024u)));      // slice 0 -> slot 0
        }
        int slot = 0;
        for (int s = 0; s < nsub; ++s) {
            unsigned vo[MAXP];                        // this slice's pieces: k-pair rows through the channel list (rows beyond it meet zero h1 columns)
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int kch = s_kidx[32 * s + 2 * (int)un[i]];
                vo[i] = (unsigned)((max(kch, 0) >> 1) * (W / 2) * 16) + no[i];
            }
            const unsigned char* const hsrc = h1b + (long)(s + 1) * 128;
            const unsigned hdst = lds_h1 + (unsigned)((s + 1) & 1) * (unsigned)p.slice_bytes;
            const bool more = s + 1 < nsub;
            for (int t = 0; t < 9; ++t) {             // (a run-time loop: the loader's code stays small)
                const int c = 9 * s + t;
                if (c >= R2) LD_TIMED(6, ld_wait_done(sy, ncomp, base2 + (unsigned)(c - R2 + 1), dead))
                ld_dma_run<MAXP>(vo, P2, p.w2p + (long)t * ((W / 2) * (W / 2) * 16), lds_w2 + (unsigned)slot * (unsigned)slot2);
                slot = slot + 1 == R2 ? 0 : slot + 1;
                int pieces = P2;
                if (t >= TMIN && more) {
                    const void* hsb = uniform_cptr(hsrc);
#define LDN_LD_H1(I) if ((I) < nq) { ld_dma1(ho[(I)], hsb, (unsigned)__builtin_amdgcn_readfirstlane((int)(hdst + (unsigned)(I) * 1024u))); ++pieces; }
                    switch (t) {      // piece i goes out with tap TMIN + i % NT
                        case 3: LDN_LD_H1(0) LDN_LD_H1(6) LDN_LD_H1(12) LDN_LD_H1(18) LDN_LD_H1(24) break;
                        case 4: LDN_LD_H1(1) LDN_LD_H1(7) LDN_LD_H1(13) LDN_LD_H1(19) LDN_LD_H1(25) break;
                        case 5: LDN_LD_H1(2) LDN_LD_H1(8) LDN_LD_H1(14) LDN_LD_H1(20) LDN_LD_H1(26) break;
                        case 6: LDN_LD_H1(3) LDN_LD_H1(9) LDN_LD_H1(15) LDN_LD_H1(21) LDN_LD_H1(27) break;
                        case 7: LDN_LD_H1(4) LDN_LD_H1(10) LDN_LD_H1(16) LDN_LD_H1(22) break;
                        default: LDN_LD_H1(5) LDN_LD_H1(11) LDN_LD_H1(17) LDN_LD_H1(23) break;
                    }
#undef LDN_LD_H1
                }
                if (c > 0) { wait_vm_rt63(pieces); ld_publish(sy, base2 + (unsigned)c); }      // chunks < c (and every slice piece issued with them) have landed
            }
        }
#endif
        wait_vm_n<0>();
        ld_publish(sy, base2 + (unsigned)nchunks);
        LT(tb)
        LD_SPAN(2, ta, tb)
        // The loader's path stays apart from the consumers' to its end (its own instances of the two workgroup barriers below): merged at the
        // barriers, its offsets and the consumers' 128 accumulator registers would be live together and hipcc spills the offsets -- scratch reloads
        // inside the stream loop, whose s_waitcnt vmcnt(0) then drains the ring every chunk.
        __syncthreads();       // (1) every wave is out of conv2: the slice / W2 regions are free
#if LDN_LD3_STAGED
        // conv3's weights through the loader's REGISTERS (it idles ~100 k cycles per block in this phase): a W3 piece (16 B) holds one k-pair of two
        // channels as {hi k0k1, lo k0k1} x 2, a consumer's A fragment the hi (or lo) halves of FOUR k-pairs of ONE channel -- staged by DMA, every one of
        // the seven consumer waves rebuilt it with eight ds_read_b64 + sixteen v_mov per n-subtile.  Here the loader shuffles ONCE: item i of a lane =
        // group g = 4 i + (lane >> 4) (K16 step (j = i, t = (lane >> 5) & 1) of lane half hh = (lane >> 4) & 1: k-pair rows 16 j + 8 t + 2 hh + {0, 1,
        // 4, 5}) x channel pair lane & 15; four 16-byte loads, four ds_write_b128 into [group][hi | lo plane][channel][16 B].  A consumer fragment is
        // then ONE conflict-free ds_read_b128 and no VALU.
        unsigned so[NS][4];
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int u = 16 * i + 8 * ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + (qq & 1) + 4 * (qq >> 1);
                const int kch = 2 * u < Kp ? s_kidx[2 * u] : -1;          // rows beyond the list meet zero h2 values
                so[i][qq] = (unsigned)(((long)(max(kch, 0) >> 1) * p.cout + 2 * (lane & 15)) * 8);
            }
        u32x4 L[NS][4];
        auto load_chunk = [&](int cc) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(uniform_cptr(p.w3p + (long)cc * (LD_CW * 8)));
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (i < nsub)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) L[i][qq] = ld_global16(src, so[i][qq]);
        };
        auto store_chunk = [&](unsigned char* slotp) {
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (i < nsub) {
                    unsigned char* d = slotp + (4 * i + (lane >> 4)) * 1024 + (lane & 15) * 32;
                    *reinterpret_cast<u32x4*>(d) = u32x4{L[i][0][0], L[i][1][0], L[i][2][0], L[i][3][0]};               // channel 2 np: hi quad
                    *reinterpret_cast<u32x4*>(d + 512) = u32x4{L[i][0][1], L[i][1][1], L[i][2][1], L[i][3][1]};         //               lo quad
                    *reinterpret_cast<u32x4*>(d + 16) = u32x4{L[i][0][2], L[i][1][2], L[i][2][2], L[i][3][2]};          // channel 2 np + 1
                    *reinterpret_cast<u32x4*>(d + 528) = u32x4{L[i][0][3], L[i][1][3], L[i][2][3], L[i][3][3]};
                }
        };
        if (nchunk3 > 0 && nsub > 0) load_chunk(0);                       // chunk 0's pieces fly through the table build
        __syncthreads();       // (2) the tables are in LDS
        if (nsub > 0) {
            int slot3i = 0;
            for (int cc = 0; cc < nchunk3; ++cc) {
                if (cc >= R3) LD_TIMED(7, ld_wait_done(sy, ncomp, base3 + (unsigned)(cc - R3 + 1), dead))
                store_chunk(s_w3 + slot3i * slot3);
                slot3i = slot3i + 1 == R3 ? 0 : slot3i + 1;
                ld_publish(sy, base3 + (unsigned)cc + 1u);                // (LDS executes this wave's writes in order: the word lands behind the quads)
                if (cc + 1 < nchunk3) load_chunk(cc + 1);                 // ... and fly through the wait for the next free slot
            }
        }
#else
        const unsigned lds_w3 = lds_off(s_w3);
        unsigned w3o[MAXP];    // W3 piece i = k-pair rows 4 i .. 4 i + 3 (256 B each: 32 channels x 8 B): lane = (row 4 i + lane / 16, channel pair lane % 16)
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int u = 4 * i + (lane >> 4);
            const int kch = 2 * u < Kp ? s_kidx[2 * u] : -1;              // rows beyond the list meet zero h2 values
            w3o[i] = (unsigned)(((long)(max(kch, 0) >> 1) * p.cout + 2 * (lane & 15)) * 8) + ld_bias(i);
        }
        if (nchunk3 > 0 && nsub > 0) ld_dma_run<MAXP>(w3o, P2, p.w3p, lds_w3);      // chunk 0 flies through the table build
        __syncthreads();       // (2) the tables are in LDS
        if (nsub > 0) {
            int slot3i = 1 == R3 ? 0 : 1;
            for (int cc = 1; cc < nchunk3; ++cc) {
                if (cc >= R3) LD_TIMED(7, ld_wait_done(sy, ncomp, base3 + (unsigned)(cc - R3 + 1), dead))
                ld_dma_run<MAXP>(w3o, P2, p.w3p + (long)cc * (LD_CW * 8), lds_w3 + (unsigned)slot3i * (unsigned)slot3);
                slot3i = slot3i + 1 == R3 ? 0 : slot3i + 1;
                wait_vm_rt63(P2);
                ld_publish(sy, base3 + (unsigned)cc);
            }
            wait_vm_n<0>();
        }
#endif
        ld_publish(sy, base3 + (unsigned)nchunk3);
        LT(tc)
        LD_SPAN(4, tb, tc)
        return;
    }

    // ---- consumers: this lane's output pixel and its nine tap rows in a slice (ZR = zero row); border class for the shift table
    const int pm = wave * 32 + l31;
    const bool pvalid = pm < npix;
    const int oy = pm / p.Wo, ox = pm % p.Wo;
    const int cls = (((oy - 1 < 0) | ((oy + 1 >= p.Hi) << 1)) * 4 + ((ox - 1 < 0) | ((ox + 1 >= p.Wi) << 1)));
    if (consumer) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        int trow[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            const bool ok = pvalid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            trow[t] = ok ? iy * p.Wi + ix : ZR;
        }
        // A (weights): k-pair rows 4 h + q (K16 half 0) / 4 h + 8 + q (half 1) of the dense slot, entry l31 of n-subtile j at + j * 256
        const unsigned RL = (unsigned)Kp * 8u;
        unsigned arow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) arow[i] = (unsigned)(4 * h + (i & 3) + 8 * (i >> 2)) * RL + (unsigned)l31 * 8u;
        // staged form: slot = [group 2 half + h][hi | lo plane][Kp channels][16 B]; bq[2 half + plane]
        unsigned bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[i] = (unsigned)((2 * (i >> 1) + h) * Kp * 32 + (i & 1) * Kp * 16 + l31 * 16);
#ifndef LDN_LD2_STYLE
#define LDN_LD2_STYLE 1     // 1 = per n-subtile eight entry reads, then six MFMAs, scheduled by hipcc; 3 = a hand-pipelined sequence (entries of subtile j - 1 requested in front of the MFMAs of subtile j, the next chunk's first operands in front of the last ones): measured SLOWER (conv2 151 k vs 143 k cycles per block) -- the loop is not bound by LDS latency
#endif
#ifndef LDN_LD_ABLATE
#define LDN_LD_ABLATE 0     // tuning only, style 1 (results are wrong): 1 = entries are not shuffled into hi / lo quads, 2 = entries are not read from LDS, 4 = no MFMA
#endif
#if LDN_LD_ABLATE & 2
#define LD_ABL_RD(PTR_, I_) (u32x2{(unsigned)(I_) + (unsigned)l31, (unsigned)(I_) * 3u + (unsigned)h})
#else
#define LD_ABL_RD(PTR_, I_) (*reinterpret_cast<const u32x2*>(PTR_))
#endif
#if LDN_LD_ABLATE & 1
#define LD_ABL_HI(E_) (u32x4{E_[0][0], E_[0][1], E_[1][0], E_[1][1]})
#define LD_ABL_LO(E_) (u32x4{E_[2][0], E_[2][1], E_[3][0], E_[3][1]})
#else
#define LD_ABL_HI(E_) (u32x4{E_[0][0], E_[1][0], E_[2][0], E_[3][0]})
#define LD_ABL_LO(E_) (u32x4{E_[0][1], E_[1][1], E_[2][1], E_[3][1]})
#endif
#if LDN_LD_ABLATE & 4
#define LD_ABL_K16(ACC_, AH_, AL_, BH_, BL_) asm volatile("" ::"v"(AH_), "v"(AL_), "v"(BH_), "v"(BL_));
#else
#define LD_ABL_K16(ACC_, AH_, AL_, BH_, BL_) LDN_K16(false, ACC_, AH_, AL_, BH_, BL_)
#endif
#if LDN_LD2_STYLE == 1
        int slot = 0;
        for (int s = 0; s < nsub; ++s) {
            const unsigned char* hs = s_h1 + (s & 1) * p.slice_bytes;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int c = 9 * s + t;
                LD_TIMED(6, ld_wait_landed(sy, base2 + (unsigned)c + 1u, q.seen))
                const unsigned char* ws = s_w2 + slot * slot2;
                slot = slot + 1 == R2 ? 0 : slot + 1;
                {
                    const unsigned rbase = (unsigned)trow[t] * 128u, rx = ((unsigned)trow[t] >> 1) & 7u;
                    bf16x8 bh[2], bl[2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const unsigned sl = 2u * (2u * half + h);
                        bh[half] = *reinterpret_cast<const bf16x8*>(hs + rbase + ((sl ^ rx) << 4));
                        bl[half] = *reinterpret_cast<const bf16x8*>(hs + rbase + (((sl + 1) ^ rx) << 4));
                    }
#if LDN_LD2_STAGED
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        if (j < nsub) {      // fragment (j, K16 half, hi | lo) = one ds_read_b128: group 2 half + h, plane, channel 32 j + l31
                            const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(ws + bq[0] + j * 512);
                            const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(ws + bq[1] + j * 512);
                            const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(ws + bq[2] + j * 512);
                            const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(ws + bq[3] + j * 512);
                            LDN_K16(false, acc[j], ah0, al0, bh[0], bl[0])
                            LDN_K16(false, acc[j], ah1, al1, bh[1], bl[1])
                            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        }
                    }
                }
#else
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        if (j < nsub) {
                            u32x2 e0[4], e1[4];
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) e0[qq] = LD_ABL_RD(ws + arow[qq] + j * 256, qq);
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) e1[qq] = LD_ABL_RD(ws + arow[4 + qq] + j * 256, 4 + qq);
                            {
                                const u32x4 ahu = LD_ABL_HI(e0);
                                const u32x4 alu = LD_ABL_LO(e0);
                                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                                LD_ABL_K16(acc[j], ah, al, bh[0], bl[0])
                            }
                            {
                                const u32x4 ahu = LD_ABL_HI(e1);
                                const u32x4 alu = LD_ABL_LO(e1);
                                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                                LD_ABL_K16(acc[j], ah, al, bh[1], bl[1])
                            }
                            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        }
                    }
                }
#endif
                ld_post_done(sy, wave, base2 + (unsigned)c + 1u);
            }
        }
#else
        // One chunk = tap t of K slice s: B fragment = the tap's h1 row (per-lane LDS address), A = the staged W2 rows.  The image's n-subtiles in
        // DESCENDING order as ONE linear, software-pipelined sequence with an entry point per subtile count (a switch that falls through): the
        // eight weight entries of subtile j - 1 are requested in front of the six MFMAs of subtile j (a full step of matrix time covers their LDS
        // latency -- half a step did not: measured), and the NEXT chunk's B fragment and first entries in front of the last six MFMAs of this one
        // when that chunk has already landed (a non-blocking look at the landed word; otherwise at the top of the next chunk).  Per accumulator the
        // products keep tail_body's order (half 0, then half 1, chunk after chunk).
        struct Ent { u32x2 e0[4], e1[4]; };
        bf16x8 bh[2], bl[2];
        Ent cur;
        // (the tap row behind an optimisation barrier: left alone, hipcc hoists the nine taps' four fragment addresses out of the slice loop --
        // 36 registers that it then spills and reloads inside the chunk)
        auto rdB = [&](const unsigned char* hs_, int tr, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
            asm volatile("" : "+v"(tr));
            const unsigned rbase = (unsigned)tr * 128u, rx = ((unsigned)tr >> 1) & 7u;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned sl = 2u * (2u * half + h);
                dh[half] = *reinterpret_cast<const bf16x8*>(hs_ + rbase + ((sl ^ rx) << 4));
                dl[half] = *reinterpret_cast<const bf16x8*>(hs_ + rbase + (((sl + 1) ^ rx) << 4));
            }
        };
        auto rdBh = [&](const unsigned char* hs_, int tr, int half, bf16x8& dh, bf16x8& dl) {
            asm volatile("" : "+v"(tr));
            const unsigned rbase = (unsigned)tr * 128u, rx = ((unsigned)tr >> 1) & 7u;
            const unsigned sl = 2u * (2u * half + h);
            dh = *reinterpret_cast<const bf16x8*>(hs_ + rbase + ((sl ^ rx) << 4));
            dl = *reinterpret_cast<const bf16x8*>(hs_ + rbase + (((sl + 1) ^ rx) << 4));
        };
        auto rdE = [&](const unsigned char* ws_, int j, Ent& d) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) d.e0[qq] = *reinterpret_cast<const u32x2*>(ws_ + arow[qq] + j * 256);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) d.e1[qq] = *reinterpret_cast<const u32x2*>(ws_ + arow[4 + qq] + j * 256);
        };
#define LDN_LD2_ASM(E_, AH_, AL_)                                                                         \
            const bf16x8 AH_ = __builtin_bit_cast(bf16x8, (u32x4{E_[0][0], E_[1][0], E_[2][0], E_[3][0]})); \
            const bf16x8 AL_ = __builtin_bit_cast(bf16x8, (u32x4{E_[0][1], E_[1][1], E_[2][1], E_[3][1]}));
#define LDN_LD2_STEP(J) {                                                                                 \
            Ent nxt;                                                                                      \
            rdE(ws, J - 1, nxt);                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            { LDN_LD2_ASM(cur.e0, ah0, al0) LDN_LD2_ASM(cur.e1, ah1, al1)                                 \
              LDN_K16(false, acc[J], ah0, al0, bh[0], bl[0])                                              \
              LDN_K16(false, acc[J], ah1, al1, bh[1], bl[1]) }                                            \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            cur = nxt; }
        int slot = 0;
        const unsigned char* ws = s_w2;
        bool pre = false;                  // this chunk's B fragment and first entries were requested during the previous chunk
        for (int s = 0; s < nsub; ++s) {
            const unsigned char* hs = s_h1 + (s & 1) * p.slice_bytes;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int c = 9 * s + t;
                if (!pre) {
                    LD_TIMED(6, ld_wait_landed(sy, base2 + (unsigned)c + 1u, q.seen))
                    rdB(hs, trow[t], bh, bl);
                    rdE(ws, nsub - 1, cur);
                }
                slot = slot + 1 == R2 ? 0 : slot + 1;
                const unsigned char* wsn = s_w2 + slot * slot2;                                    // the next chunk's slot,
                const unsigned char* hsn = t == 8 ? s_h1 + ((s + 1) & 1) * p.slice_bytes : hs;     // slice
                const int trn = trow[(t + 1) % 9];                                                 // and tap row
                switch (nsub) {
                    default:
                        if constexpr (NS >= 8) LDN_LD2_STEP(7)
                        [[fallthrough]];
                    case 7:
                        if constexpr (NS >= 8) LDN_LD2_STEP(6)
                        [[fallthrough]];
                    case 6:
                        if constexpr (NS >= 8) LDN_LD2_STEP(5)
                        [[fallthrough]];
                    case 5:
                        if constexpr (NS >= 8) LDN_LD2_STEP(4)
                        [[fallthrough]];
                    case 4:
                        if constexpr (NS >= 4) LDN_LD2_STEP(3)
                        [[fallthrough]];
                    case 3:
                        if constexpr (NS >= 4) LDN_LD2_STEP(2)
                        [[fallthrough]];
                    case 2:
                        LDN_LD2_STEP(1)
                        [[fallthrough]];
                    case 1: {
                        LDN_LD2_ASM(cur.e0, ah0, al0) LDN_LD2_ASM(cur.e1, ah1, al1)
                        __builtin_amdgcn_sched_barrier(0);
                        pre = false;
                        if (c + 1 < nchunks) {      // (wave-uniform)
                            const unsigned need = base2 + (unsigned)c + 2u;
                            if (q.seen < need) q.seen = __builtin_amdgcn_readfirstlane(ld_lds_read(&sy->landed));      // one look, no spin
                            pre = q.seen >= need;
                        }
                        if (pre) rdE(wsn, nsub - 1, cur);
                        __builtin_amdgcn_sched_barrier(0);
                        LDN_K16(false, acc[0], ah0, al0, bh[0], bl[0])
                        __builtin_amdgcn_sched_barrier(0);
                        if (pre) rdBh(hsn, trn, 0, bh[0], bl[0]);      // (in place: the MFMAs that read this half have been issued)
                        __builtin_amdgcn_sched_barrier(0);
                        LDN_K16(false, acc[0], ah1, al1, bh[1], bl[1])
                        __builtin_amdgcn_sched_barrier(0);
                        if (pre) rdBh(hsn, trn, 1, bh[1], bl[1]);
                    }
                }
                ld_post_done(sy, wave, base2 + (unsigned)c + 1u);
                ws = wsn;
            }
        }
#undef LDN_LD2_ASM
#undef LDN_LD2_STEP
#endif
    }
    LT(tb)
    LD_SPAN(2, ta, tb)
    __syncthreads();       // (1) every wave is out of conv2: the slice / W2 regions are free
    // (as in tail_body: the lane index behind an optimisation barrier, or hipcc keeps the conv3 phase's per-lane addresses alive through conv2)
    asm volatile("" : "+v"(lane));
    l31 = lane & 31;
    h = lane >> 5;

    // ======================================================================================================== conv3 (1x1)
    {   // conversion tables (gathered through the channel list) by the seven waves that are not the loader
        const int t7 = tid;                                               // threads 0 .. 447
        // (round 6) the gathers of five entries per thread in flight together: as plain loops hipcc emits LDS read -> global load -> s_waitcnt
        // vmcnt(0) -> ds_write per entry -- ten memory latencies in a row per block at width 256, ~12 k of the 16 k cycles this phase took
        constexpr int TB = 5;
        for (int i = t7; i < NP; i += 448) {
            const int ch = i < Kb ? s_kidx[i] : -1;
            const float a = p.sc2[max(ch, 0)], c = p.ps2[max(ch, 0)];
            s_tab[i] = ch >= 0 ? a : 0.f;
            s_tab[NP + i] = ch >= 0 ? c : 0.f;
        }
        for (int i0 = t7; i0 < 16 * NP; i0 += 448 * TB) {
            float v[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int i = min(i0 + 448 * u, 16 * NP - 1);
                const int k = i / NP, n = i - k * NP;
                const int ch = n < Kb ? s_kidx[n] : -1;
                const float x = p.sh2[k * W + max(ch, 0)];
                v[u] = ch >= 0 ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < TB; ++u)
                if (i0 + 448 * u < 16 * NP) s_tab[2 * NP + i0 + 448 * u] = v[u];
        }
    }
    __syncthreads();       // (2) the tables are in LDS

    if (!consumer) return;

    // In place: the 16 fp32 accumulators of n-subtile j become 16 dwords of bf16 pairs (tail_body's conversion, same values)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float v[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int n0 = 32 * j + 8 * q4 + 4 * h;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_tab + n0);
            const f32x4 ps = *reinterpret_cast<const f32x4*>(s_tab + NP + n0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(s_tab + 2 * NP + cls * NP + n0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[4 * q4 + e] = fmaxf(acc[j][4 * q4 + e] * sc[e] + sh[e], 0.f) - ps[e];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float x0 = v[8 * t + 2 * d], x1 = v[8 * t + 2 * d + 1];
                const bf16x2 hi = {(__bf16)x0, (__bf16)x1};
                const bf16x2 lo = {(__bf16)(x0 - (float)hi[0]), (__bf16)(x1 - (float)hi[1])};
                acc[j][8 * t + d] = __builtin_bit_cast(float, hi);
                acc[j][8 * t + 4 + d] = __builtin_bit_cast(float, lo);
            }
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

    LT(tc)
    LD_SPAN(3, tb, tc)
    const int trw = lane >> 3, tcq = lane & 7;                 // epilogue layout: lane = (row trw + 8 it, 4 channels at 4 tcq)
    const unsigned a3_lane = (unsigned)(2 * h * W3_ROW + l31 * 8);
    const unsigned a3q = (unsigned)(h * 1024 + l31 * 16);      // staged form: the lane's channel in the hi plane of lane half h's group
    float* const scr = reinterpret_cast<float*>(s_scr + wave * 4096);   // this wave's 32 x 32 transpose scratch
#if LDN_LD3_DEFER && LDN_LD3_STAGED
    // DEFERRED epilogue (round 6, an experiment kept as a switch: measured NEUTRAL, 221-225 us per block either way -- timers around its pieces
    // show ~1.0 k cycles per chunk of instruction issue (VALU + stores) against ~0.2 k of waiting for the residual tile: the epilogue is issue-bound,
    // there is no latency to hide).  A chunk's K loop ends with the four
    // ds_write_b128 of its accumulators into the wave's scratch and the request of its residual tile; everything that has to WAIT -- the transposed
    // read-back, the residual's arrival, the stores -- is placed between the n-subtile steps of the NEXT chunk's K loop (program order = issue order:
    // the steps are separate basic blocks), where the MFMAs of this wave and its SIMD partner cover it.  Same values, same order per element.
    int slot = 0;
    f32x4 res[4], xr[4], sh;
    int c0p = 0;                                            // first channel of the chunk whose epilogue is pending
    auto epi_read = [&]() {                                 // piece 1: the tile back from the scratch in the row layout (written one K loop ago)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = trw + 8 * it;
            xr[it] = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tcq ^ (row & 7)) << 2));
        }
    };
    auto epi_store = [&]() {                                // piece 2: + shift + residual, ReLU, store; piece 3: the GAP partials
        wait_vm<0>();                                       // the residual tile (requested behind the previous K loop) and this wave's earlier stores
        asm volatile("" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(sh));
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int prow = wave * 32 + trw + 8 * it;
            f32x4 x = xr[it] + sh + res[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
            if (prow < npix) {
                __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p.out + (size_t)(pix0 + prow) * p.ldo + c0p + tcq * 4));
                csum += x;
            }
        }
        xr[0] = csum;                                       // (handed to the third piece)
    };
    auto epi_gap = [&]() {
        if (p.colsum) {
            f32x4 csum = xr[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);
            if (trw == 0) *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * 8 + wave) * p.cout + c0p + tcq * 4) = csum;
        }
    };
    for (int cc = 0; cc < nchunk3; ++cc) {
        const int c0 = cc * LD_CW;
        const bool pend = cc > 0;                           // (wave-uniform) chunk cc - 1's epilogue is pending
        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
        LD_TIMED(7, ld_wait_landed(sy, base3 + (unsigned)cc + 1u, q.seen))
        const unsigned char* ws = s_w3 + slot * slot3;
        slot = slot + 1 == R3 ? 0 : slot + 1;
        // pieces of the pending epilogue: the read-back behind step 0, the stores behind step js (late: the residual tile then had most of
        // a K loop to arrive), the GAP partials one step later
        const int js = min(LDN_LD3_STORE_STEP, nsub - 1), jg = min(js + 1, nsub - 1);
        bool todo = pend;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (j < nsub) {
                bf16x8 ah[2], al[2];      // group 4 j + 2 t + h of the slot: [hi plane 32 channels x 16 B | lo plane]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = *reinterpret_cast<const bf16x8*>(ws + a3q + (4 * j + 2 * t) * 1024);
                    al[t] = *reinterpret_cast<const bf16x8*>(ws + a3q + (4 * j + 2 * t) * 1024 + 512);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 hb = frag_hi(j, t), lb = frag_lo(j, t);
                    LDN_K16(false, acc3, ah[t], al[t], hb, lb)
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                if (pend) {
                    if (j == 0) epi_read();
                    if (j == js) epi_store();
                    if (j == jg) { epi_gap(); todo = false; }
                }
            }
        }
        if (todo) { epi_read(); epi_store(); epi_gap(); }   // (an image without live conv2 channels: no K steps to hide behind)
        ld_post_done(sy, wave, base3 + (unsigned)cc + 1u);
        // this chunk's accumulators into the scratch (C layout -> rows of 32 channels per pixel, 16-byte slots XOR-swizzled with the pixel), its
        // residual tile and shift requested: both are consumed inside the next K loop
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = {acc3[4 * q4], acc3[4 * q4 + 1], acc3[4 * q4 + 2], acc3[4 * q4 + 3]};
            *reinterpret_cast<f32x4*>(scr + l31 * 32 + (((2 * q4 + h) ^ (l31 & 7)) << 2)) = v;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int prow = wave * 32 + trw + 8 * it;
            const float* src = (p.residual && prow < npix) ? p.residual + (size_t)(pix0 + prow) * p.ldr + c0 + tcq * 4 : g_tail_zero;
            res[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        }
        sh = ld_global_f4(p.sh3 + c0 + tcq * 4);            // (address space 1: a flat load here makes the compiler drain vmcnt in front of it)
        c0p = c0;
    }
    if (nchunk3 > 0) { epi_read(); epi_store(); epi_gap(); }      // the last chunk's epilogue
#else
    int slot = 0;
    for (int cc = 0; cc < nchunk3; ++cc) {
        const int c0 = cc * LD_CW;
        // residual tile in the layout the epilogue stores in, requested before the K loop that hides its latency
        f32x4 res[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int prow = wave * 32 + trw + 8 * it;
            const float* src = (p.residual && prow < npix) ? p.residual + (size_t)(pix0 + prow) * p.ldr + c0 + tcq * 4 : g_tail_zero;
#if LDN_LD3_ABLATE & 3
            res[it] = f32x4{(float)prow, 0.f, 1.f, 2.f};
#else
            res[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
#endif
        }
        f32x4 sh = ld_global_f4(p.sh3 + c0 + tcq * 4);
        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
        LD_TIMED(7, ld_wait_landed(sy, base3 + (unsigned)cc + 1u, q.seen))
        const unsigned char* ws = s_w3 + slot * slot3;
        slot = slot + 1 == R3 ? 0 : slot + 1;
#if LDN_LD3_STAGED
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (j < nsub) {
                bf16x8 ah[2], al[2];      // group 4 j + 2 t + h of the slot: [hi plane 32 channels x 16 B | lo plane]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = *reinterpret_cast<const bf16x8*>(ws + a3q + (4 * j + 2 * t) * 1024);
                    al[t] = *reinterpret_cast<const bf16x8*>(ws + a3q + (4 * j + 2 * t) * 1024 + 512);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 hb = frag_hi(j, t), lb = frag_lo(j, t);
#if LDN_LD3_ABLATE & 4
                    asm volatile("" :: "v"(ah[t]), "v"(al[t]), "v"(hb), "v"(lb));
#else
                    LDN_K16(false, acc3, ah[t], al[t], hb, lb)
#endif
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }
#else
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (j < nsub) {
                u32x2 e[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        e[t][qq] = *reinterpret_cast<const u32x2*>(ws + a3_lane + (16 * j + 8 * t + (qq & 1) + 4 * (qq >> 1)) * W3_ROW);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4 ahu = {e[t][0][0], e[t][1][0], e[t][2][0], e[t][3][0]};
                    const u32x4 alu = {e[t][0][1], e[t][1][1], e[t][2][1], e[t][3][1]};
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahu), al = __builtin_bit_cast(bf16x8, alu);
                    const bf16x8 hb = frag_hi(j, t), lb = frag_lo(j, t);
                    LDN_K16(false, acc3, ah, al, hb, lb)
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }
#endif
        ld_post_done(sy, wave, base3 + (unsigned)cc + 1u);
        // the residual (and this wave's earlier stores) before this chunk's stores, which then fly through the next chunk (tail_body: gfx9 counts
        // loads and stores in one vmcnt and completes them out of order with each other)
#if LDN_LD3_ABLATE & 1
        asm volatile("" :: "v"(acc3), "v"(res[0]), "v"(res[3]), "v"(sh));
        continue;
#endif
        wait_vm<0>();
        asm volatile("" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(sh));
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
            f32x4 x = *reinterpret_cast<const f32x4*>(scr + row * 32 + ((tcq ^ (row & 7)) << 2));
            x = x + sh + res[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
            if (prow < npix) {
#if !(LDN_LD3_ABLATE & 2)
                __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p.out + (size_t)(pix0 + prow) * p.ldo + c0 + tcq * 4));
#endif
                csum += x;
            }
        }
        if (p.colsum) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] = sum_lane_bits_345(csum[e]);
            if (trw == 0) *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * 8 + wave) * p.cout + c0 + tcq * 4) = csum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
#endif
    LT(td)
    LD_SPAN(4, tc, td)
}

// Whole-image maps of at most 224 pixels leave the workgroup's eighth wave without pixels: the loader / consumer form applies.
template <int NS>
__global__ __launch_bounds__(512, 2) void k_chain_ld(const ChainArgs p) {
    constexpr int W = NS * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int HW = p.H * p.Wd;
    {   // GAP partial slots of the waves without pixels (the loader, and consumers beyond the map): zero, once -- nobody else writes them
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (wave >= ceil_div(HW, 32))
            for (int c = lane * 4; c < p.C; c += 256)
                *reinterpret_cast<f32x4*>(p.colsum + ((size_t)b * 8 + wave) * p.C + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#if LDN_LD_PRIO
    // Static priorities (MI355X_MICROARCH.md, two waves per SIMD): the second-dispatched half of the workgroup (waves 4-6) loses every VALU / issue
    // arbitration against its older SIMD partner and finishes every phase last -- and a phase ends with its LAST wave.  The loader's few instructions
    // sit on every hand-off's critical path.
    {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (w == LD_LOADER) __builtin_amdgcn_s_setprio(3);
        else if (w >= 4) __builtin_amdgcn_s_setprio(LDN_LD_PRIO);
    }
#endif
    LdSeq q;
#ifdef LDN_TRACE
    for (int k = 0; k < 8; ++k) q.t[k] = 0;
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
                                  mask_i, nullptr, idx_i, cnt_i, s_f, s_w, nullptr, true);
        }
        CT(c1)
        phase_fence();
        CT(c2)
        q.base = 0u;
        q.seen = 0u;
        {   // ---- conv1 -> h1 (pre-split)
            HeadArgs ha;
            ha.x = xin; ha.ldx = p.ldx; ha.B = p.B; ha.HW = HW; ha.cin = p.C; ha.W = W;
            ha.w1s = uniform_ptr(cb->w1s); ha.n_idx = idx_i; ha.n_cnt = cnt_i;
            ha.sc1 = uniform_ptr(cb->sc1); ha.sh1 = uniform_ptr(cb->sh1); ha.ps1 = uniform_ptr(cb->ps1);
            ha.h1 = p.h1; ha.h1_row_bytes = p.h1_row_bytes; ha.pix_per_blk = HW; ha.mblocks = 1; ha.xs = nullptr;
            head_ld<NS>(ha, b, smem, p.lds_total, opaque_tid(), q);
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
            tail_ld<NS>(ta, b, smem, p.lds_total, opaque_tid(), q);
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
    if (g_ld_trace && (threadIdx.x & 63) == 0) {
        unsigned long long* r = g_ld_trace + ((size_t)b * 8 + (threadIdx.x >> 6)) * 8;
        for (int k = 0; k < 8; ++k) r[k] = q.t[k];
    }
#endif
}

}  // namespace ldn
