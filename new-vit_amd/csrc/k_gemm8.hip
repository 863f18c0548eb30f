// FP8 (OCP e4m3) linear layers of the encoder blocks: BASELINE.json configs[4] ("fp8 MFMA inference"), SURVEY.md 8d row c5:
// e4m3 operands, one scale per tensor taken from its absolute maximum, fp32 accumulate.  The reference has no fp8 code; the
// arithmetic restated here is  F.linear(x, W, b)  (attention.py:58,67; mlp.py:35,38) with both operands rounded to e4m3:
//     y = (sa * sw) * (q(x / sa) . q(W / sw)^T) + b,   sa = max|x| / 448,  sw = max|W| / 448,  q = round-to-nearest-even e4m3
// (the test-side CPU restatement of the same formula is named in tests/test_fp8_gpu.py).
//
//   absmax_kernel    max|x| of a 16-bit activation tensor -> one fp32 word (atomicMax on the bit pattern: non-negative
//                    floats order like unsigned integers)
//   quant8_kernel    x * (448 / max|x|) -> e4m3 bytes (v_cvt_pk_fp8_f32, OCP format on gfx950)
//   gemm8_kernel     C = epi(scale * A8[M,K] . W8[N,K]^T + bias): the 128x128 tile / LDS-DMA / XOR-swizzle structure of
//                    k_gemm16.hip with 128-byte rows holding 128 k instead of 64.  A lane's 16-byte fragment read feeds
//                    TWO v_mfma_f32_16x16x32_fp8_fp8 (8 bytes each): both operands use the same k permutation, so the
//                    contraction is complete without any shuffle.
//
// First correct version (round 1): the activation scale is dynamic, so every GEMM input costs an absmax pass and a quantise
// pass over HBM; folding them into the producing kernels with calibrated scales, and the MX-scaled 16x16x128 MFMA (2x rate),
// are next-round work.
#include <stdlib.h>

#include <type_traits>

#include "mst_common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) long i64x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int BM = 128, BN = 128, BKB = 128;   // BKB: k per stage = bytes per LDS row
constexpr int TILE_BYTES = BM * BKB;           // 16 KiB per operand per stage
constexpr float F8_MAX = 448.0f;               // largest finite e4m3 (OCP)

// ---- activation quantisation -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, int64_t n8, unsigned int* amax_bits) {
    typedef typename V8<T>::type vec8;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const vec8 v = *reinterpret_cast<const vec8*>(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)v[j]));
    }
    // one atomic per workgroup, and only when it can raise the running maximum (a stale read is merely too small): thousands of
    // same-address atomics serialise in one L2 channel -- the first version spent more time there than reading the tensor
    __shared__ float wmax[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > __uint_as_float(*(volatile unsigned int*)amax_bits)) atomicMax(amax_bits, __float_as_uint(m));
    }
}

__device__ __forceinline__ float f8_inv_scale(float amax) { return amax > 0.f ? F8_MAX / amax : 0.f; }

template <typename T>
__global__ __launch_bounds__(256) void quant8_kernel(const T* __restrict__ x, int64_t n8, const float* __restrict__ amax,
                                                     uint8_t* __restrict__ out) {
    typedef typename V8<T>::type vec8;
    const float inv = f8_inv_scale(*amax);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const vec8 v = *reinterpret_cast<const vec8*>(x + i * 8);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_fmed3f((float)v[j] * inv, -F8_MAX, F8_MAX);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *reinterpret_cast<u32x2*>(out + i * 8) = (u32x2){(unsigned)lo, (unsigned)hi};
    }
}

// ---- GEMM ---------------------------------------------------------------------------------------
// MX = true: one v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, all E8M0 block scales = 2^0) per 128-k stage instead of four
// 16x16x32 fp8 MFMAs: the block-scaled form runs at twice the bf16 rate (MI355X_MICROARCH.md, FP8 row), the plain one at 1x.
template <int EPI, typename OutT, bool MX>
__global__ __launch_bounds__(256) void gemm8_kernel(const uint8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ W,
                                                    int64_t ldw, const float* __restrict__ bias,
                                                    const float* __restrict__ a_amax, float w_scale, OutT* C, int64_t ldc,
                                                    int M, int N, int K, const float* __restrict__ gamma, float col_scale,
                                                    int scale_cols, int tiles_n, int nwg, float* out_amax,
                                                    const float* __restrict__ c_amax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                   // [2][128 rows][128 B]
    char* const Ws = smem + 2 * TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm = wave >> 1, wn = wave & 1;

    // LDS-DMA sources: instruction i of this wave fills tile rows (wave*4+i)*8 .. +7, lane -> (row, 16-byte chunk); the chunk
    // fetched is the one whose swizzled home is LDS slot lane&7
    const uint8_t* asrc[4];
    const uint8_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int am = m0 + r;
        am = am < M ? am : M - 1;
        asrc[i] = A + (int64_t)am * lda + c * 16;
        wsrc[i] = W + (int64_t)(n0 + r) * ldw + c * 16;
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = buf * TILE_BYTES + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i]), LDS_PTR(As + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i]), LDS_PTR(Ws + off), 16, 0, 0);
            asrc[i] += BKB;
            wsrc[i] += BKB;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wm * 64 + (lane & 15)) * 128;
    const int w_row_off = (wn * 64 + (lane & 15)) * 128;

    const int nk = K / BKB;
    stage(0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nk) stage((t + 1) & 1);
        const char* Ab = As + (t & 1) * TILE_BYTES;
        const char* Wb = Ws + (t & 1) * TILE_BYTES;
        if constexpr (MX) {
            i32x4 af[2][4], wf[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) af[kk][j] = *reinterpret_cast<const i32x4*>(Ab + a_row_off + j * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[kk][i] = *reinterpret_cast<const i32x4*>(Wb + w_row_off + i * 16 * 128 + coff);
            }
            constexpr int ONE = 0x7F7F7F7F;   // E8M0 127 = 2^0 in every byte
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x8 wv = __builtin_shufflevector(wf[0][i], wf[1][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x8 av = __builtin_shufflevector(af[0][j], af[1][j], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, acc[i][j], 0, 0, 0, ONE, 0, ONE);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
                i64x2 af[4], wf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const i64x2*>(Ab + a_row_off + j * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const i64x2*>(Wb + w_row_off + i * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[i][0], af[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[i][1], af[j][1], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }

    // epilogue: lane owns C[m][n..n+3], m = m0+wm*64+j*16+(lane&15), n = n0+wn*64+i*16+(lane>>4)*4
    const float dq = (*a_amax) * (1.0f / F8_MAX) * w_scale;
    float omax = 0.f;   // max |C as stored| of this lane (16-bit outputs feeding the next e4m3 GEMM)
    float cq = 0.f;     // e4m3 output: quanta per unit under the caller's calibrated scale
    if constexpr (sizeof(OutT) == 1) cq = f8_inv_scale(*c_amax);
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (EPI == MST_EPI_RESIDUAL && gamma) gv = *reinterpret_cast<const float4*>(gamma + n);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = (n + r < scale_cols) ? col_scale : 1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 64 + j * 16 + (lane & 15);
                if (!FULL && m >= M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaf(acc[i][j][r], dq, bb[r]) * sc[r];
                    if constexpr (EPI == MST_EPI_BIAS_GELU) {
                        // the epilogue is VALU-bound (64 values per lane against 48 MFMAs per wave at K = 384): the sigmoid-form
                        // GELU of the fused MLP kernel (36 / 44 issue cycles, |err| 2.7e-4 / 2.5e-5) instead of the erf form (60)
                        if constexpr (std::is_same<OutT, float>::value) v[r] = gelu_erf(v[r]);
                        else if constexpr (std::is_same<OutT, f16_t>::value) v[r] = gelu_sig<f16_t>(v[r]);
                        else v[r] = gelu_sig<bf16_t>(v[r]);
                    }
                    if (EPI == MST_EPI_BIAS_RELU) v[r] = fmaxf(v[r], 0.f);
                }
                OutT* cp = C + (int64_t)m * ldc + n;
                if constexpr (EPI == MST_EPI_RESIDUAL) {
                    const float4 xv = *reinterpret_cast<const float4*>(cp);
                    float4 o;
                    o.x = xv.x + gv.x * v[0];
                    o.y = xv.y + gv.y * v[1];
                    o.z = xv.z + gv.z * v[2];
                    o.w = xv.w + gv.w * v[3];
                    *reinterpret_cast<float4*>(cp) = o;
                } else if constexpr (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (sizeof(OutT) == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r] * cq, -F8_MAX, F8_MAX);
                    int pk = 0;
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                    *reinterpret_cast<int*>(cp) = pk;
                } else {
                    typedef __attribute__((ext_vector_type(4))) OutT o4;
                    o4 pk;
                    pk[0] = (OutT)v[0];
                    pk[1] = (OutT)v[1];
                    pk[2] = (OutT)v[2];
                    pk[3] = (OutT)v[3];
                    *reinterpret_cast<o4*>(cp) = pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) omax = fmaxf(omax, fabsf((float)pk[r]));
                }
            }
        }
    };
    if (m0 + BM <= M) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if constexpr (sizeof(OutT) == 2) {
        if (out_amax) {   // uniform.  One poll of the running maximum per workgroup: polls of one address serialise in an L2 channel
            float* red = reinterpret_cast<float*>(smem);
            omax = wave_max(omax);
            __syncthreads();   // every wave is done with the operand tiles
            if (lane == 0) red[wave] = omax;
            __syncthreads();
            if (tid == 0) {
                omax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                if (omax > *(volatile float*)out_amax) atomicMax((unsigned int*)out_amax, __float_as_uint(omax));
            }
        }
    }
}

template <int EPI, typename OutT>
int launch_t(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* a_amax, float w_scale,
             void* C, int64_t ldc, int64_t M, int N, int K, const float* gamma, float cs, int sc, float* out_amax,
             const float* c_amax, hipStream_t s) {
    static mst_lds_once lds_once[2];
    static const bool mx = !(getenv("MST_FP8_MX") && atoi(getenv("MST_FP8_MX")) == 0);
    auto kern = mx ? gemm8_kernel<EPI, OutT, true> : gemm8_kernel<EPI, OutT, false>;
    mst_allow_lds((const void*)kern, 4 * TILE_BYTES, &lds_once[mx]);
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = N / BN;
    const int nwg = tiles_m * tiles_n;
    kern<<<dim3(nwg), dim3(256), 4 * TILE_BYTES, s>>>((const uint8_t*)A, lda, (const uint8_t*)W, ldw, bias, a_amax, w_scale,
                                                       (OutT*)C, ldc, (int)M, N, K, gamma, cs, sc, tiles_n, nwg, out_amax, c_amax);
    return mst_check_launch("gemm8");
}

template <typename OutT>
int dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* a_amax, float w_scale,
             void* C, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma, float cs, int sc, float* out_amax,
             const float* c_amax, hipStream_t s) {
    switch (epi) {
        case MST_EPI_BIAS:
            return launch_t<MST_EPI_BIAS, OutT>(A, lda, W, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, gamma, cs, sc, out_amax, c_amax, s);
        case MST_EPI_BIAS_GELU:
            return launch_t<MST_EPI_BIAS_GELU, OutT>(A, lda, W, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, gamma, cs, sc, out_amax, c_amax, s);
        case MST_EPI_BIAS_RELU:
            return launch_t<MST_EPI_BIAS_RELU, OutT>(A, lda, W, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, gamma, cs, sc, out_amax, c_amax, s);
    }
    mst_set_error("gemm8: bad epilogue %d", epi);
    return MST_EINVAL;
}

}  // namespace

__global__ void amax_merge_kernel(float* out, const float* in, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmaxf(out[i], in[i]);
}

int launch_amax_merge(float* out, const float* in, int n, hipStream_t s) {
    if (n <= 0) return MST_OK;
    amax_merge_kernel<<<(n + 63) / 64, 64, 0, s>>>(out, in, n);
    return mst_check_launch("amax_merge");
}

// scan: *amax = max(*amax, max|x|) first (amax_rw != nullptr), then quantise under `amax`
static int quant8_impl(const void* x, int dt, int64_t n, float* amax_rw, const float* amax, void* out8, hipStream_t s) {
    MST_CHECK_ARG(dt == MST_BF16 || dt == MST_F16, "quantize_fp8: input dtype %d must be bf16 or fp16", dt);
    MST_CHECK_ARG(n >= 0 && n % 8 == 0, "quantize_fp8: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return MST_OK;
    const int64_t n8 = n / 8;
    const int64_t want = (n8 + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    if (dt == MST_BF16) {
        if (amax_rw) absmax_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, n8, (unsigned int*)amax_rw);
        quant8_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, n8, amax, (uint8_t*)out8);
    } else {
        if (amax_rw) absmax_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)x, n8, (unsigned int*)amax_rw);
        quant8_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)x, n8, amax, (uint8_t*)out8);
    }
    return mst_check_launch("quantize_fp8");
}

int launch_quant8(const void* x, int dt, int64_t n, float* amax, void* out8, int scan, hipStream_t s) {
    return quant8_impl(x, dt, n, scan ? amax : nullptr, amax, out8, s);
}

int launch_quant8_static(const void* x, int dt, int64_t n, const float* amax, void* out8, hipStream_t s) {
    return quant8_impl(x, dt, n, nullptr, amax, out8, s);
}

int launch_gemm8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* bias, const float* a_amax,
                 float w_scale, void* C, int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                 float col_scale, int scale_cols, float* out_amax, const float* c_amax, hipStream_t s) {
    MST_CHECK_ARG(A8 && W8 && C && a_amax, "gemm8: null pointer");
    MST_CHECK_ARG(cdt != MST_F8E4M3 || (c_amax && epi != MST_EPI_RESIDUAL && !out_amax),
                  "gemm8: an e4m3 C needs c_amax and a non-residual epilogue");
    MST_CHECK_ARG(!out_amax || (cdt != MST_F32 && epi != MST_EPI_RESIDUAL), "gemm8: out_amax needs a 16-bit, non-residual C");
    MST_CHECK_ARG(K > 0 && K % BKB == 0, "gemm8: K=%d must be a multiple of %d", K, BKB);
    MST_CHECK_ARG(N > 0 && N % BN == 0, "gemm8: N=%d must be a multiple of %d", N, BN);
    MST_CHECK_ARG(lda % 16 == 0 && ldw % 16 == 0 && ldc % 4 == 0, "gemm8: lda/ldw must be multiples of 16, ldc of 4");
    MST_CHECK_ARG(M < (1ll << 31) - BM, "gemm8: M too large");
    MST_CHECK_ARG(cdt == MST_F32 || cdt == MST_BF16 || cdt == MST_F16 || cdt == MST_F8E4M3, "gemm8: bad C dtype %d", cdt);
    MST_CHECK_ARG(epi != MST_EPI_RESIDUAL || cdt == MST_F32, "gemm8: residual epilogue needs f32 C");
    if (M <= 0) return MST_OK;
    if (epi == MST_EPI_RESIDUAL)
        return launch_t<MST_EPI_RESIDUAL, float>(A8, lda, W8, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, gamma, col_scale,
                                                 scale_cols, nullptr, nullptr, s);
    if (cdt == MST_F8E4M3) return dispatch<uint8_t>(A8, lda, W8, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, out_amax, c_amax, s);
    if (cdt == MST_F32) return dispatch<float>(A8, lda, W8, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, out_amax, c_amax, s);
    if (cdt == MST_BF16) return dispatch<bf16_t>(A8, lda, W8, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, out_amax, c_amax, s);
    return dispatch<f16_t>(A8, lda, W8, ldw, bias, a_amax, w_scale, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, out_amax, c_amax, s);
}
