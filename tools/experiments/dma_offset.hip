// Tuning only: does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* f = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 2048; i += 64) f[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const float* s = src + threadIdx.x * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(s), "s"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = f[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, o);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 8; ++blk) printf("LDS floats [%4d..]: %g %g %g ... %g\n", blk * 256, r[blk * 256], r[blk * 256 + 1], r[blk * 256 + 2], r[blk * 256 + 255]);
    return 0;
}
