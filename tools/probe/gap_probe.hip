// probe: what makes the ~5.8 us idle gap BEHIND some kernels (rocprofv3 --kernel-trace: next.start - prev.end)?
// sequences X, tiny, X, tiny ... for writer kernels with plain / nontemporal / write-through stores, a reader, and an LDS user.
// build: hipcc --offload-arch=gfx950 -O3 gap_probe.hip -o gap_probe ; run under rocprofv3 --kernel-trace --stats
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1.f; }
__global__ void w_plain(f32x4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{1.f, 2.f, 3.f, 4.f}; }
__global__ void w_nt(f32x4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, p + i); }
__global__ void w_wt(f32x4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = {1.f, 2.f, 3.f, 4.f};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
    }
}
__global__ void r_only(const f32x4* p, size_t n, float* o) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a[0] == 123.f) o[0] = a[1];
}
__global__ void lds_user(float* o) {
    extern __shared__ float s[];
    s[threadIdx.x] = threadIdx.x; __syncthreads();
    if (s[(threadIdx.x + 1) % blockDim.x] == -1.f) o[0] = 1.f;
}
int main() {
    const size_t MB = 1 << 20;
    f32x4* buf; float* o;
    hipMalloc(&buf, 512 * MB); hipMalloc(&o, 64);
    hipFuncSetAttribute((const void*)lds_user, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        for (size_t mb : {1, 8, 32, 128, 512}) {
            const size_t n = mb * MB / 16;
            hipLaunchKernelGGL(w_plain, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
            hipLaunchKernelGGL(w_nt, dim3(2048), dim3(256), 0, 0, buf, n);    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
            hipLaunchKernelGGL(w_wt, dim3(2048), dim3(256), 0, 0, buf, n);    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
            hipLaunchKernelGGL(r_only, dim3(2048), dim3(256), 0, 0, buf, n, o); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
        }
        hipLaunchKernelGGL(lds_user, dim3(256), dim3(512), 140 * 1024, 0, o); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
        hipLaunchKernelGGL(lds_user, dim3(4096), dim3(512), 140 * 1024, 0, o); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, o);
    }
    hipDeviceSynchronize();
    printf("done\n");
}
