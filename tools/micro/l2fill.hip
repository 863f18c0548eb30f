// Microbenchmark: per-CU fill bandwidth when every CU streams the SAME small (L2-resident) buffer, the access
// pattern of the fused MLP's weight streaming.  Variants: LDS-DMA (global_load_lds_dwordx4) vs register loads,
// 4 or 8 issuing waves, burst depth.   hipcc --offload-arch=gfx950 -O3 l2fill.hip -o l2fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int WAVES, int MODE, int BURST>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ buf, size_t bytes, int iters, unsigned* sink) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave >= WAVES) return;
    const size_t piece = 1024;                       // bytes per wave-instruction
        u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)WAVES * BURST * piece;      // bytes per round of the workgroup; bytes % stride == 0
    for (int it = 0; it < iters; ++it) {
        const char* src = buf + (size_t)wave * piece + lane * 16;
        const char* end = buf + bytes;
        for (; src < end; src += stride) {
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                if (MODE == 0) {
                    char* dst = lds + (size_t)(u * WAVES + wave) * piece;   // <= 96 KB
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src + (size_t)u * WAVES * piece), LDS_PTR(dst), 16, 0, 0);
                } else {
                    acc += *reinterpret_cast<const u32x4*>(src + (size_t)u * WAVES * piece);
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (MODE == 1 && acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
    if (MODE == 0 && lds[tid] == 0x7f && iters < 0) sink[1] = 1;
}

template <int WAVES, int MODE, int BURST>
void run(const char* name, const char* buf, size_t bytes, unsigned* sink) {
    const int iters = 40, grid = 256;
    auto k = fill_kernel<WAVES, MODE, BURST>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<grid, 512, 150 * 1024>>>(buf, bytes, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<grid, 512, 150 * 1024>>>(buf, bytes, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double tot = (double)bytes * iters * grid;
    printf("%-34s waves=%d burst=%2d : %7.3f ms  %6.2f TB/s aggregate  %5.1f B/clk/CU @2.4GHz\n", name, WAVES, BURST, ms,
           tot / ms / 1e9, tot / grid / (ms * 1e-3 * 2.4e9));
}

int main() {
    const size_t bytes = 2359296;   // one layer's W1+W2 in bf16
    char* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    run<4, 0, 12>("LDS-DMA", buf, bytes, sink);
    run<8, 0, 6>("LDS-DMA", buf, bytes, sink);
    run<8, 0, 12>("LDS-DMA", buf, bytes, sink);
    run<4, 0, 4>("LDS-DMA", buf, bytes, sink);
    run<1, 0, 12>("LDS-DMA", buf, bytes, sink);
    run<2, 0, 12>("LDS-DMA", buf, bytes, sink);
    run<4, 1, 12>("register loads", buf, bytes, sink);
    run<8, 1, 12>("register loads", buf, bytes, sink);
    run<8, 1, 4>("register loads", buf, bytes, sink);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
