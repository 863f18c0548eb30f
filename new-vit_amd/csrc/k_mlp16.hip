// Fused MLP block of a ViT layer for E = 384, 16-bit MFMA operands:
//     x <- x + ls2 * ( fc2( gelu( fc1( LayerNorm2(x) ) ) ) )          block.py:93-94,113; mlp.py:34-40
//     xn <- normalise(x)   (optional: the NEXT block's norm1, its affine folded into that block's QKV weights)
// in ONE kernel.  The 4E-wide hidden activations never leave the CU, LayerNorm needs no pass of its own, and per
// token the kernel moves ~3x1536 B in + 1536 B (+768 B) out instead of ~12 KB for LN + fc1 + fc2 as three launches
// (the unfused layer is HBM-bound: DESIGN.md section 5).
//
// Structure (persistent, one 8-wave workgroup per CU, 128 token rows per tile):
//   * waves work in PAIRS on 32 rows: both keep the 32 normalised rows as GEMM1 B-operand fragments in registers
//     (LN affine folded into W1 / b1 on the host); wave `hf` of a pair computes hidden units [16hf, 16hf+16) of
//     each 32-wide chunk (GEMM1) and output columns [192hf, 192hf+192) (GEMM2), so every weight fragment read from
//     LDS feeds TWO MFMAs (one fragment per MFMA saturates the LDS read port: measured on the first version);
//   * the GELU'd GEMM1 accumulators, converted in place, ARE GEMM2's B operand; the two half-fragments of a pair
//     are swapped through a 16 KB LDS buffer (same lane index on both sides); GEMM2 runs one chunk behind GEMM1,
//     so one `s_barrier` per chunk orders both the weight ring and the swap, and the GELU hides under MFMAs;
//   * weights stream through two 3-deep LDS rings (W1 rows / W2 columns of a chunk, 24 KB each), pre-packed on the
//     host as the exact swizzled LDS image (every LDS-DMA piece is 1 KiB contiguous); one DMA group stays in
//     flight behind counted `s_waitcnt vmcnt(6)`; b1 travels in registers one chunk ahead of its use;
//   * W2's columns are permuted on the host to the accumulator's k order and its rows so that a lane ends up
//     with 8-column groups (128-B row segments across the 4 lane groups) for the residual read-modify-write;
//   * epilogue: x += ls2*(y + b2); next LayerNorm from the registers (pair-wise statistics through LDS).
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int E = 384, HID = 1536, CH = 32;            // CH hidden units per chunk
constexpr int NCHUNK = HID / CH;                        // 48
constexpr int W1_BYTES = CH * E * 2;                    // 24 KiB
constexpr int W2_BYTES = E * CH * 2;                    // 24 KiB
constexpr int CHUNK_BYTES = W1_BYTES + W2_BYTES;        // packed weights per chunk in global memory
constexpr int NSLOT = 3;
constexpr int W2_RING = NSLOT * W1_BYTES;               // byte offset of the W2 ring
constexpr int HBUF = 2 * NSLOT * W1_BYTES;              // byte offset of the pair-swap buffer (2 x 8 KiB)
constexpr int LDS_BYTES = HBUF + 2 * 8192;              // 163,840 = all of the LDS
constexpr int KS = E / 32;                              // 12 k-steps of GEMM1
constexpr int NTH = E / 32;                             // 12 output tiles (of 16 columns) per wave

// ---- hand-pipelined LDS fragment reads.  hipcc schedules `ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs` 24 times per
// chunk here (LDS latency fully exposed); the reads are therefore issued as asm, DEPTH ahead of their MFMAs, behind
// counted lgkmcnt waits, each wait followed by sched_barrier(0) so no MFMA is hoisted above it (cdna guide rule 18).
template <int OFF, typename V> __device__ __forceinline__ void lds_read_b128(V& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
constexpr int DEPTH = 4;

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
#ifdef ABL_NOBAR
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <typename T>
__global__ __launch_bounds__(512) void mlp16_kernel(float* __restrict__ x, T* __restrict__ xn_out,
                                                    const char* __restrict__ wpack,
                                                    const float* __restrict__ b1f, const float* __restrict__ b2,
                                                    const float* __restrict__ ls2, int M, int ntiles, float eps) {
    typedef typename V8<T>::type vec8;
    typedef typename V8<T>::half_type vec4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave >> 1, hf = wave & 1;
    const int frow = lane & 15, g = lane >> 4;
    const int fslot = (g ^ ((0 - (frow >> 2)) & 3)) * 16;
    const int w1_off = (16 * hf + frow) * 64 + fslot;              // + ks*2048 within a W1 slot
    const int w2_off = (16 * (NTH * hf) + frow) * 64 + fslot;      // + tt*1024 within a W2 slot
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    char* const hmine = smem + HBUF + ((pr * 2 + 0) * 2 + hf) * 512 + lane * 8;        // + mt*1024 + parity*8192
    char* const hother = smem + HBUF + ((pr * 2 + 0) * 2 + (hf ^ 1)) * 512 + lane * 8;

    // LDS-DMA: a half-chunk (W1 or W2 image, 24 pieces of 1 KiB) = 3 pieces per wave
    auto dma_half = [&](const char* src, char* dst) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + (wave * 3 + u) * 1024 + lane * 16),
                                             LDS_PTR(dst + (wave * 3 + u) * 1024), 16, 0, 0);
    };
    auto dma_w1 = [&](int c) { dma_half(wpack + (size_t)c * CHUNK_BYTES, smem + (c % NSLOT) * W1_BYTES); };
    auto dma_w2 = [&](int c) { dma_half(wpack + (size_t)c * CHUNK_BYTES + W1_BYTES, smem + W2_RING + (c % NSLOT) * W2_BYTES); };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // weight stream of this tile: W1(0), W1(1), W2(0) now; then one group {W1(c+2), W2(c+1)} per chunk
        dma_w1(0);
        dma_w1(1);
        dma_w2(0);

        const int mrow = tile * 128 + pr * 32 + frow;   // rows mrow and mrow + 16
        // ---- LayerNorm statistics and the GEMM1 B-operand fragments (normalised rows, 16-bit)
        vec8 xa[2][KS];
        float mean_old[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mrow + 16 * mt;
            const float* xr = x + (size_t)(m < M ? m : M - 1) * E + 8 * g;
            float v[KS * 8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 a = *reinterpret_cast<const float4*>(xr + 32 * ks);
                const float4 b = *reinterpret_cast<const float4*>(xr + 32 * ks + 4);
                v[8 * ks + 0] = a.x; v[8 * ks + 1] = a.y; v[8 * ks + 2] = a.z; v[8 * ks + 3] = a.w;
                v[8 * ks + 4] = b.x; v[8 * ks + 5] = b.y; v[8 * ks + 6] = b.z; v[8 * ks + 7] = b.w;
            }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < KS * 8; ++i) sum += v[i];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / E);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < KS * 8; ++i) { const float d = v[i] - mean; sq = fmaf(d, d, sq); }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / E) + eps);
            mean_old[mt] = mean;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) xa[mt][ks][j] = (T)((v[8 * ks + j] - mean) * rstd);
        }
        // ---- GEMM2 accumulators start at b2: tile tt, register r <-> column 192*hf + 32*(tt>>1) + 8*g + 4*(tt&1) + r
        const int col0 = 192 * hf + 8 * g;
        f32x4 acc[NTH][2];
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + col0 + 32 * (tt >> 1) + 4 * (tt & 1));
            acc[tt][0] = bv;
            acc[tt][1] = bv;
        }
        f32x4 bias_next = *reinterpret_cast<const f32x4*>(b1f + 16 * hf + 4 * g);   // b1 of chunk 0

        vec4 hown[2];                                    // this wave's GELU'd half-fragments of the previous chunk
#pragma unroll 1
        for (int c = 0; c <= NCHUNK; ++c) {
            // barrier(c): W1(c) and W2(c-1) have landed for every wave; every wave wrote its H(c-1) halves
            if (c == 0) wait_vm_barrier<0>();
            else if (c < NCHUNK - 1) wait_vm_barrier<6>();
            else if (c == NCHUNK - 1) wait_vm_barrier<3>();
            else wait_vm_barrier<0>();
            const f32x4 bias_cur = bias_next;
#ifndef ABL_NOBIAS
            if (c + 1 < NCHUNK) bias_next = *reinterpret_cast<const f32x4*>(b1f + (c + 1) * CH + 16 * hf + 4 * g);
#endif
#ifndef ABL_NOLOAD
            if (c + 2 < NCHUNK) dma_w1(c + 2);          // slot of W1(c-1): free since barrier(c)
            if (c + 1 < NCHUNK) dma_w2(c + 1);          // slot of W2(c-2): free since barrier(c)
#endif
            // ---- GEMM2 of chunk c-1 (operands complete since the barrier)
            if (c >= 1) {
                const int par = (c - 1) & 1;
                vec8 hfrag[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const vec4 oth = *reinterpret_cast<const vec4*>(hother + mt * 1024 + par * 8192);
                    const vec4 lo = hf ? oth : hown[mt], hi = hf ? hown[mt] : oth;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hfrag[mt][j] = lo[j];
                        hfrag[mt][4 + j] = hi[j];
                    }
                }
                const unsigned w2a = lds_base + W2_RING + ((c - 1) % NSLOT) * W2_BYTES + w2_off;
                vec8 w2[NTH];
                wait_lgkm<0>();
                static_for<0, DEPTH>([&](auto i) { lds_read_b128<decltype(i)::value * 1024>(w2[decltype(i)::value], w2a); });
                static_for<0, NTH>([&](auto i) {
                    constexpr int tt = decltype(i)::value;
                    if constexpr (tt + DEPTH < NTH) lds_read_b128<(tt + DEPTH) * 1024>(w2[tt + DEPTH], w2a);
                    wait_lgkm<(NTH - 1 - tt < DEPTH) ? (NTH - 1 - tt) : DEPTH>();
#ifdef ABL_NOMFMA
                    asm volatile("" ::"v"(w2[tt]));
#else
                    acc[tt][0] = mfma16(w2[tt], hfrag[0], acc[tt][0]);
                    acc[tt][1] = mfma16(w2[tt], hfrag[1], acc[tt][1]);
#endif
                });
            }
            // ---- GEMM1 of chunk c: this wave's 16 hidden units x 32 rows, then GELU and the pair swap
            if (c < NCHUNK) {
                f32x4 h0 = bias_cur, h1 = bias_cur;
                const unsigned w1a = lds_base + (c % NSLOT) * W1_BYTES + w1_off;
                vec8 w1[KS];
                wait_lgkm<0>();
                static_for<0, DEPTH>([&](auto i) { lds_read_b128<decltype(i)::value * 2048>(w1[decltype(i)::value], w1a); });
                static_for<0, KS>([&](auto i) {
                    constexpr int ks = decltype(i)::value;
                    if constexpr (ks + DEPTH < KS) lds_read_b128<(ks + DEPTH) * 2048>(w1[ks + DEPTH], w1a);
                    wait_lgkm<(KS - 1 - ks < DEPTH) ? (KS - 1 - ks) : DEPTH>();
#ifdef ABL_NOMFMA
                    asm volatile("" ::"v"(w1[ks]));
#else
                    h0 = mfma16(w1[ks], xa[0][ks], h0);
                    h1 = mfma16(w1[ks], xa[1][ks], h1);
#endif
                });
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hown[0][r] = (T)gelu_poly(h0[r]);
                    hown[1][r] = (T)gelu_poly(h1[r]);
                }
                *reinterpret_cast<vec4*>(hmine + (c & 1) * 8192) = hown[0];
                *reinterpret_cast<vec4*>(hmine + 1024 + (c & 1) * 8192) = hown[1];
            }
        }

        // ---- epilogue: residual (+ LayerScale) on this wave's 192 columns, then the next LayerNorm
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mrow + 16 * mt;
            float* xw = x + (size_t)(m < M ? m : M - 1) * E + col0;
            float4 xo[NTH];
#pragma unroll
            for (int tt = 0; tt < NTH; ++tt) xo[tt] = *reinterpret_cast<const float4*>(xw + 32 * (tt >> 1) + 4 * (tt & 1));
#pragma unroll
            for (int tt = 0; tt < NTH; ++tt) {
                float4 gs = make_float4(1.f, 1.f, 1.f, 1.f);
                if (ls2) gs = *reinterpret_cast<const float4*>(ls2 + col0 + 32 * (tt >> 1) + 4 * (tt & 1));
                f32x4& a = acc[tt][mt];
                a[0] = xo[tt].x + gs.x * a[0];
                a[1] = xo[tt].y + gs.y * a[1];
                a[2] = xo[tt].z + gs.z * a[2];
                a[3] = xo[tt].w + gs.w * a[3];
                if (m < M) *reinterpret_cast<float4*>(xw + 32 * (tt >> 1) + 4 * (tt & 1)) = make_float4(a[0], a[1], a[2], a[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {            // shifted one-pass statistics (shift = the row's old mean)
                    const float d = a[r] - mean_old[mt];
                    s1[mt] += d;
                    s2[mt] = fmaf(d, d, s2[mt]);
                }
            }
        }
        asm volatile("s_barrier" ::: "memory");          // every wave is done with the rings and the swap buffer
        if (xn_out) {
            // pair-wise row statistics through LDS: stat[pr][hf][mt][frow] (after the 4 lane groups are folded)
            float* stat = reinterpret_cast<float*>(smem + HBUF);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                s1[mt] += __shfl_xor(s1[mt], 16, 64); s1[mt] += __shfl_xor(s1[mt], 32, 64);
                s2[mt] += __shfl_xor(s2[mt], 16, 64); s2[mt] += __shfl_xor(s2[mt], 32, 64);
                if (g == 0) {
                    stat[(((pr * 2 + hf) * 2 + mt) * 16 + frow) * 2 + 0] = s1[mt];
                    stat[(((pr * 2 + hf) * 2 + mt) * 16 + frow) * 2 + 1] = s2[mt];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float o1 = stat[(((pr * 2 + (hf ^ 1)) * 2 + mt) * 16 + frow) * 2 + 0];
                const float o2 = stat[(((pr * 2 + (hf ^ 1)) * 2 + mt) * 16 + frow) * 2 + 1];
                const float d1 = (s1[mt] + o1) * (1.0f / E);
                const float var = (s2[mt] + o2) * (1.0f / E) - d1 * d1;
                const float mean = mean_old[mt] + d1;
                const float rstd = rsqrtf(fmaxf(var, 0.f) + eps);
                const int m = mrow + 16 * mt;
                if (m < M) {
                    T* xo = xn_out + (size_t)m * E + col0;
#pragma unroll
                    for (int k6 = 0; k6 < NTH / 2; ++k6) {
                        vec8 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = (T)((acc[2 * k6][mt][r] - mean) * rstd);
                            o[4 + r] = (T)((acc[2 * k6 + 1][mt][r] - mean) * rstd);
                        }
                        *reinterpret_cast<vec8*>(xo + 32 * k6) = o;
                    }
                }
            }
            asm volatile("s_barrier" ::: "memory");      // statistics consumed before the next tile's swaps
        }
    }
}

template <typename T>
int launch_t(float* x, void* xn_out, const void* wpack, const float* b1f, const float* b2, const float* ls2, int64_t M,
             float eps, hipStream_t s) {
    static bool attr_set = false;
    auto kern = mlp16_kernel<T>;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int ntiles = (int)((M + 127) / 128);
    const int nblk = ntiles < 256 ? ntiles : 256;
    kern<<<dim3(nblk), dim3(512), LDS_BYTES, s>>>(x, (T*)xn_out, (const char*)wpack, b1f, b2, ls2, (int)M, ntiles, eps);
    return mst_check_launch("mlp16");
}

}  // namespace

int launch_mlp16(float* x, void* xn_out, int dt, const void* wpack, const float* b1f, const float* b2, const float* ls2,
                 int64_t M, int E_, float eps, hipStream_t s) {
    MST_CHECK_ARG(E_ == E, "mlp_fused: embed_dim=%d unsupported (384)", E_);
    MST_CHECK_ARG(M > 0 && M < (1ll << 31) - 128, "mlp_fused: bad M");
    if (dt == MST_BF16) return launch_t<bf16_t>(x, xn_out, wpack, b1f, b2, ls2, M, eps, s);
    if (dt == MST_F16) return launch_t<f16_t>(x, xn_out, wpack, b1f, b2, ls2, M, eps, s);
    mst_set_error("mlp_fused: dtype %d unsupported (f16 / bf16)", dt);
    return MST_EINVAL;
}
