// Tuning only: L2 -> CU fetch rate of a gathered weight chunk as a function of the row stride.
// Every workgroup (512 threads, one per CU x 256) reads NR rows of 128 B (lane = (row l/8, 16 B at l%8)) from the SAME
// L2-resident buffer, row r of pass c at  base + r * stride + (c % per_row) * 128  (the n-major weight layout: stride = K * 4 B)
// or at  base + c * NR * 128 + r * 128  (the chunk-major layout: stride 128).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_stride tools/experiments/fetch_stride.hip && /tmp/fetch_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_fetch(const unsigned char* base, long stride, int NR, int per_row, int passes, int chunk_major, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    for (int c = 0; c < passes; ++c) {
        const int cc = c % per_row;
        for (int r0 = wave * 8; r0 < NR; r0 += 64) {
            const int r = r0 + (lane >> 3);
            const unsigned char* src = chunk_major ? base + ((long)cc * NR + r) * 128 + (lane & 7) * 16
                                                   : base + (long)r * stride + (long)cc * 128 + (lane & 7) * 16;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
int main() {
    const size_t bytes = 64 << 20;
    unsigned char* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4096);
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char* name; long stride; int NR, per_row, cm; };
    std::vector<Cfg> cfgs = {
        {"n-major K=1024 (4 KB stride), 176 rows", 4096, 176, 32, 0},
        {"n-major K=256  (1 KB stride), 704 rows", 1024, 704, 8, 0},
        {"n-major 4 KB + 128 pad, 176 rows", 4224, 176, 32, 0},
        {"n-major 4 KB + 256 pad, 176 rows", 4352, 176, 32, 0},
        {"n-major 2 KB stride, 176 rows", 2048, 176, 16, 0},
        {"chunk-major (contiguous), 176 rows", 128, 176, 32, 1},
        {"n-major K=1024, 256 rows", 4096, 256, 32, 0},
        {"chunk-major, 256 rows", 128, 256, 32, 1},
    };
    for (int grid : {256, 512, 1024})
    for (auto& c : cfgs) {
        const int passes = 2048;
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fetch, dim3(grid), dim3(512), 0, 0, buf, c.stride, c.NR, c.per_row, passes, c.cm, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double gb = (double)grid * passes * c.NR * 128 / 1e9;
        printf("grid %4d  %-44s %8.3f ms  %8.1f GB/s total  %6.1f GB/s per workgroup\n", grid, c.name, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / grid);
    }
    return 0;
}
