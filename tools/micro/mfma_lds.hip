// Micro-benchmark: what does one ds_read_b128 per v_mfma_f32_32x32x16_bf16 cost a wave that is alone on its SIMD?
// Variants (template V): 0 = MFMA only; 1 = + 1 read per MFMA into a ring of 6 register sets, counted wait; 2 = the same, waits
// removed; 3 = reads every second MFMA; 4 = 2 reads per MFMA; 5 = read + wait, MFMAs all on ONE accumulator; 6 = as 1 with the
// accumulators in VGPRs (asm MFMA, "+v"); 7 = as 1, 16x16x32 MFMAs (2 per read).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int OFF, typename V> __device__ __forceinline__ void rd(V& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void waitl() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters, const char* wsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 6 * 24576 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 4095);
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + lane * 16 + (V == 12 ? 5 * 24576 : V == 13 ? 3 * 24576 - 4096 : 0);
    f32x16 acc[12];
    f32x4 acc16[48];
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int t = 0; t < 48; ++t) acc16[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 b0, b1, w[6], w2[6];
    for (int e = 0; e < 8; ++e) { b0[e] = (__bf16)(0.01f * (lane + e)); b1[e] = (__bf16)(0.02f * (lane - e)); }
    for (int q = 0; q < 6; ++q) w[q] = b0, w2[q] = b1;
    constexpr int D = 4, R = 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 108 * 24576, 0x00020000);
    int dsrc = 0, dslot = 1;
    if (V != 0) sfor<0, D>([&](auto q) { rd<decltype(q)::value * 1024>(w[decltype(q)::value % R], base); });
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 8 || V == 9) asm volatile("s_barrier" ::: "memory");
        if constexpr (V == 10 || V == 11) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        const int my_src = dsrc, my_slot = dslot;
        dsrc = dsrc + 24576 == 108 * 24576 ? 0 : dsrc + 24576;
        dslot = dslot + 1 == 6 ? 1 : dslot + 1;           // slots 1..5 (slot 0 is the one being read)
        sfor<0, 24>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            if constexpr ((V == 9 || V == 10) && i % 4 == 3) {
                constexpr int u = i / 4, hi = u >= 4 ? 4096 : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + my_slot * 24576 + wave * 6144 + hi), 16, wave * 6144 + lane * 16, my_src + hi, u * 1024 - hi, 0);
            }
            if constexpr (V == 1 || V == 2 || V == 5 || V == 6 || V == 7 || V >= 8 || (V == 3 && i % 2 == 0) || V == 4)
                rd<((i + D) % 24) * 1024>(w[(i + D) % R], base);
            if constexpr (V == 4) rd<((i + D + 7) % 24) * 1024>(w2[(i + D) % R], base);
            if constexpr (V == 1 || V == 5 || V == 6 || V == 7 || V >= 8) waitl<D>();
            else if constexpr (V == 4) waitl<2 * D>();
            else __builtin_amdgcn_sched_barrier(0);
            if constexpr (V == 5) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % R], (i & 1) ? b1 : b0, acc[0], 0, 0, 0);
            else if constexpr (V == 6) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % 12]) : "v"(w[i % R]), "v"((i & 1) ? b1 : b0));
            else if constexpr (V == 7) {
                acc16[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i % R], b0, acc16[2 * i], 0, 0, 0);
                acc16[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i % R], b1, acc16[2 * i + 1], 0, 0, 0);
            } else acc[i % 12] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % R], (i & 1) ? b1 : b0, acc[i % 12], 0, 0, 0);
            if constexpr (V == 4) asm volatile("" ::"v"(w2[i % R]));
        });
    }
    unsigned long long t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int t = 0; t < 48; ++t) s += acc16[t][0] + acc16[t][3];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int V> void run(const char* name, unsigned long long* d, float* sink, const char* wsrc) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 24576);
    for (int rep = 0; rep < 3; ++rep) k<V><<<256, 256, 6 * 24576>>>(d, sink, iters, wsrc);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 1024; ++i) m += (double)h[i];
    m /= 1024.0;
    printf("%-46s %7.2f cycles per slot\n", name, m / (iters * 24.0));
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 1024 * 8); hipMalloc(&sink, 4);
    char* wsrc; hipMalloc(&wsrc, 108 * 24576); hipMemset(wsrc, 0, 108 * 24576);
    run<0>("MFMA 32x32x16 only", d, sink, wsrc);
    run<1>("+ 1 ds_read_b128 per MFMA, counted wait", d, sink, wsrc);
    run<2>("+ 1 read per MFMA, no waits", d, sink, wsrc);
    run<3>("+ 1 read per 2 MFMAs, no waits", d, sink, wsrc);
    run<4>("+ 2 reads per MFMA, counted wait", d, sink, wsrc);
    run<5>("+ 1 read per MFMA, ONE accumulator chain", d, sink, wsrc);
    run<6>("+ 1 read per MFMA, accumulators in VGPRs (asm)", d, sink, wsrc);
    run<7>("+ 1 read per 2 MFMA 16x16x32", d, sink, wsrc);
    run<8>("V1 + s_barrier per 24 slots", d, sink, wsrc);
    run<9>("V8 + 6 LDS-DMA pieces per 24 slots", d, sink, wsrc);
    run<10>("V9 with vmcnt(12) before the barrier", d, sink, wsrc);
    run<11>("V1 + vmcnt(12), barrier, no DMA", d, sink, wsrc);
    run<12>("V1 reading LDS bytes 120-144 KiB", d, sink, wsrc);
    run<13>("V1 reading LDS bytes 68-92 KiB", d, sink, wsrc);
    return 0;
}
