// probe: fp32 MFMA 32x32x2 issue rate vs #independent accumulators and waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int s = 0; s < NACC; ++s) for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    float x = a + threadIdx.x * 1e-6f, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
            for (int s = 0; s < NACC; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[s], 0, 0, 0);
    }
    float sum = 0;
    for (int s = 0; s < NACC; ++s) for (int r = 0; r < 16; ++r) sum += acc[s][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int NACC>
void run(int blocks_per_cu, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)grid * 4 * iters * 16;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4GHz)\n", NACC, blocks_per_cu,
           mfma * 4096 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 16.0 * blocks_per_cu));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int bpc : {1, 2, 4}) { run<1>(bpc, d); run<2>(bpc, d); run<4>(bpc, d); }
    return 0;
}
