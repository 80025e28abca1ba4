// Tuning only: L2 -> CU fetch rate of one workgroup per CU (8 waves) as a function of the bytes in flight, for plain global loads
// (to registers) and for LDS-DMA (global_load_lds_dwordx4).  All workgroups read the same L2-resident 4 MB buffer, each wave its own
// 1 KB pieces; N pieces are issued back to back, then the wave waits for all of them.
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/fetch_depth tools/experiments/fetch_depth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N, int MODE>
__global__ __launch_bounds__(512) void k_fetch(const unsigned char* base, int passes, float* sink, unsigned mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc = {0, 0, 0, 0};
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem + wave * N * 1024);
    unsigned off = (blockIdx.x * 7919u + wave * 131u) * 1024u;
    for (int c = 0; c < passes; ++c) {
        if (MODE == 0) {
            f32x4 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(base + ((off + (unsigned)i * 8192u) & mask) + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) dma16(base + ((off + (unsigned)i * 8192u) & mask) + lane * 16, lds0 + i * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        off += N * 8192u;
    }
    if (MODE == 1) acc = *reinterpret_cast<const f32x4*>(smem + threadIdx.x * 16);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
template <int N, int MODE> void run(const unsigned char* buf, float* sink, int grid, unsigned mask, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 4096 / N;
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_fetch<N, MODE>), dim3(grid), dim3(512), MODE ? 8 * N * 1024 : 0, 0, buf, passes, sink, mask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double gb = (double)grid * passes * N * 8 * 1024 / 1e9;
    printf("%-10s grid %4d  buffer %5u KB  %2d KB in flight per wave (%3d KB per CU)  %8.3f ms  %8.1f GB/s total  %6.1f GB/s per workgroup\n",
           what, grid, (mask + 1) >> 10, N, 8 * N, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / grid);
}
int main() {
    const size_t bytes = 256 << 20;
    unsigned char* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4096);
    hipMemset(buf, 0, bytes);
    for (unsigned mask : {(4u << 20) - 1, (64u << 20) - 1})
    for (int grid : {256, 512}) {
        run<1, 0>(buf, sink, grid, mask, "loads"); run<2, 0>(buf, sink, grid, mask, "loads"); run<4, 0>(buf, sink, grid, mask, "loads");
        run<8, 0>(buf, sink, grid, mask, "loads"); run<16, 0>(buf, sink, grid, mask, "loads");
        run<1, 1>(buf, sink, grid, mask, "lds-dma"); run<2, 1>(buf, sink, grid, mask, "lds-dma"); run<4, 1>(buf, sink, grid, mask, "lds-dma");
        run<8, 1>(buf, sink, grid, mask, "lds-dma"); if (grid == 256) run<16, 1>(buf, sink, grid, mask, "lds-dma");
    }
    return 0;
}
