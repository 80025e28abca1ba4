// tuning only: what one workgroup per CU fetches through LDS-DMA from an L2-RESIDENT region that every workgroup reads (the per-image weight subsets
// of the channel path: FETCH_SIZE says they are L2 hits, yet the kernels see ~26 GB/s per CU) as a function of the GRANULARITY of the gather:
//   pattern 0: a wave instruction = 1 KB contiguous (lane i -> bytes 16 i ..)
//   pattern 1: 16-byte pieces, every other one (stride 32 B: half of every 64-byte sector is used)
//   pattern 2: 16-byte pieces at pseudo-random 16-byte slots of a 4 KB row (one row per instruction)
//   pattern 3: 32-byte runs (two lanes) at random 32-byte slots of an 8 KB row
//   pattern 4: 64-byte runs (four lanes) at random 64-byte slots of a 16 KB row
// D instructions in flight per wave, 8 waves per workgroup.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ablate/libgather_probe.so tools/probe/gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D>
__global__ __launch_bounds__(512) void k_gather(const unsigned char* base, long region, int reps, int pattern, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(smem) + (unsigned)wave * D * 1024u;
    // per-lane offset inside the instruction's row, row bytes per instruction
    long lane_off, row_bytes;
    const unsigned hsh = (unsigned)(lane * 2654435761u) >> 16;
    if (pattern == 0) { lane_off = lane * 16; row_bytes = 1024; }
    else if (pattern == 1) { lane_off = lane * 32; row_bytes = 2048; }
    else if (pattern == 2) { lane_off = (long)((lane * 4 + (hsh & 3)) & 255) * 16; row_bytes = 4096; }          // one random slot of each group of four
    else if (pattern == 3) { lane_off = (long)(((lane >> 1) * 4 + (hsh & 3)) & 127) * 64 + (lane & 1) * 16; row_bytes = 8192; }
    else { lane_off = (long)(((lane >> 2) * 4 + (hsh & 3)) & 63) * 256 + (lane & 3) * 16; row_bytes = 16384; }
    const long rows = region / row_bytes;
    for (int r = 0; r < reps; ++r) {
        int issued = 0;
#pragma unroll 1
        for (long i = wave; i < rows; i += 8) {
            dma16(base + i * row_bytes + lane_off, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(issued % D) * 1024u)));
            ++issued;
            wait_vm<D - 1>();
        }
        wait_vm<0>();
    }
    if (reinterpret_cast<float*>(smem)[tid] == 123.456f) sink[0] = 1.f;
}

extern "C" int gather_probe(const void* base, long region, int wgs, int reps, int pattern, float* sink, void* stream) {
    hipLaunchKernelGGL((k_gather<8>), dim3(wgs), dim3(512), 8 * 8 * 1024, (hipStream_t)stream, (const unsigned char*)base, region, reps, pattern, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
