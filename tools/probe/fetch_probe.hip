// tuning only: what ONE workgroup per CU can fetch through LDS-DMA as a function of the bytes it keeps in flight (the per-image kernels of the
// channel path -- k_head / k_tail / k_chain -- are bound by exactly this: DESIGN.md 4v).  Every workgroup streams its own region of `bytes_per_wg`
// bytes `reps` times with D 1-KB DMA instructions in flight per wave (8 waves: 8 D KB per CU); MODE 0 = global_load_lds_dwordx4, 1 = global_load_dwordx4
// into registers.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ablate/libfetch_probe.so tools/probe/fetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, int MODE>
__global__ __launch_bounds__(512) void k_fetch(const unsigned char* base, long bytes_per_wg, int reps, int stride_lines, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* src = base + (long)blockIdx.x * bytes_per_wg;
    const unsigned lds0 = (unsigned)(uintptr_t)(smem) + (unsigned)wave * D * 1024u;
    const long n_instr = bytes_per_wg / (8 * 1024);          // per wave
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        // lane's 16 bytes: consecutive lanes consecutive pieces (stride_lines == 0), or every lane on its own 128-byte line
        long off = (long)wave * 1024 + (stride_lines ? (long)(lane & 7) * 16 + (long)(lane >> 3) * 128 : (long)lane * 16);
        int issued = 0;
#pragma unroll 1
        for (long i = 0; i < n_instr; ++i) {
            if (MODE == 0) {
                dma16(src + off, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(issued % D) * 1024u)));
            } else {
                acc += *reinterpret_cast<const f32x4*>(src + off);
            }
            off += 8 * 1024;
            ++issued;
            if (MODE == 0) wait_vm<D - 1>();
        }
        if (MODE == 0) wait_vm<0>();
    }
    if (MODE == 0) acc[0] = reinterpret_cast<float*>(smem)[tid];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

#define LAUNCH(D, MODE) hipLaunchKernelGGL((k_fetch<D, MODE>), dim3(wgs), dim3(512), 8 * D * 1024, (hipStream_t)stream, (const unsigned char*)base, bytes_per_wg, reps, stride_lines, sink)
extern "C" int fetch_probe(const void* base, long bytes_per_wg, int wgs, int reps, int depth, int mode, int stride_lines, float* sink, void* stream) {
    if (mode == 0) {
        switch (depth) {
            case 1: LAUNCH(1, 0); break; case 2: LAUNCH(2, 0); break; case 4: LAUNCH(4, 0); break; case 8: LAUNCH(8, 0); break;
            case 12: LAUNCH(12, 0); break; case 16: LAUNCH(16, 0); break; default: return -1;
        }
    } else {
        LAUNCH(1, 1);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
