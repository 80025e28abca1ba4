// probe: semantics of __builtin_amdgcn_global_load_lds (16 B) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int perm_mul) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each wave fills 1 KiB at lds + wave*256 floats; lane reads global float4 index (lane*perm_mul)%64 of its wave's 1 KiB
    const float* g = src + wave * 256 + ((lane * perm_mul) & 63) * 4;
    float* l = lds + __builtin_amdgcn_readfirstlane(wave) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int pm : {1, 3}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, pm);
        hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
            float want = (float)(w * 256 + ((lane * pm) & 63) * 4 + e);
            if (o[w * 256 + lane * 4 + e] != want) ++bad;
        }
        printf("perm_mul=%d: %s (bad=%d) sample lds[4..7]=%g %g %g %g\n", pm, bad ? "MISMATCH" : "OK lane-linear dest, per-lane src", bad, o[4], o[5], o[6], o[7]);
    }
    return 0;
}
