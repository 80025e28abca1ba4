// probe: lane semantics of v_permlane32_swap / v_permlane16_swap / DPP row_ror:8 on gfx950 (hipcc --offload-arch=gfx950 permlane_probe.hip -o permlane_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int l = threadIdx.x;
    const unsigned a = l, b = 100 + l;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const int d = __builtin_amdgcn_update_dpp(0, l, 0x128, 0xf, 0xf, false);
    o[l * 5 + 0] = r[0]; o[l * 5 + 1] = r[1]; o[l * 5 + 2] = q[0]; o[l * 5 + 3] = q[1]; o[l * 5 + 4] = d;
}
int main() {
    int* d; hipMalloc(&d, 64 * 5 * 4);
    k<<<1, 64>>>(d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: swap32 {%3d,%3d}  swap16 {%3d,%3d}  ror8 %2d\n", l, h[l*5], h[l*5+1], h[l*5+2], h[l*5+3], h[l*5+4]);
}
