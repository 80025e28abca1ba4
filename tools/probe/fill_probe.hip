// probe: per-CU L2/HBM -> LDS fill rate with global_load_lds_dwordx4, as a function of
//   * waves per CU issuing (1 block per CU, W waves),
//   * row stride of the 128-byte segments a wave instruction gathers (8 rows x 128 B per instruction),
//   * footprint (L2-resident vs HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k_fill(const float* __restrict__ src, size_t row_stride_f, int rows_per_cu, int seg_per_row, int iters,
                       float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const float* base = src + (size_t)blockIdx.x * rows_per_cu * row_stride_f;
    // each instruction: 8 rows x 128 B; instruction index ii -> rows (ii*8 .. ii*8+7) of segment `seg`
    int ii = wave;
    const int ins_per_seg = rows_per_cu / 8;
    for (int it = 0; it < iters; ++it) {
        for (int seg = 0; seg < seg_per_row; ++seg) {
            for (int q = ii; q < ins_per_seg; q += nwaves) {
                const int row = q * 8 + (lane >> 3);
                const float* g = base + (size_t)row * row_stride_f + seg * 32 + (lane & 7) * 4;
                float* l = lds + ((q & 3) * 256) + wave * 1024;     // recycle a small LDS window
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}

int main() {
    const int cus = 256;
    const size_t bytes = (size_t)1 << 30;   // 1 GiB source
    float *src, *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, cus * 4);
    hipMemset(src, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("waves/CU  row_stride  rows/CU  segs  footprint/CU  ->  GB/s per CU   TB/s chip\n");
    for (int waves : {1, 4, 8}) {
        for (size_t stride_b : {(size_t)128, (size_t)512, (size_t)1024, (size_t)4096}) {
            for (int rows : {32, 256}) {
                int segs = (int)(stride_b / 128); if (segs > 32) segs = 32;   // walk the row chunk by chunk like the conv K loop
                const size_t foot = (size_t)rows * stride_b;
                if (foot * cus > bytes) continue;
                const int iters = (int)(((size_t)8 << 20) / ((size_t)rows * 128 * segs)) + 1;   // ~8 MiB per CU
                const size_t lds = 16 * 1024 * 4;
                hipLaunchKernelGGL(k_fill, dim3(cus), dim3(waves * 64), lds, 0, src, stride_b / 4, rows, segs, 1, sink);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_fill, dim3(cus), dim3(waves * 64), lds, 0, src, stride_b / 4, rows, segs, iters, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double moved = (double)rows * 128 * segs * iters;
                printf("%5d  %9zu  %7d  %4d  %9.0f KB  ->  %8.1f   %6.2f\n", waves, stride_b, rows, segs, foot / 1024.0,
                       moved / (ms * 1e-3) / 1e9, moved * cus / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
