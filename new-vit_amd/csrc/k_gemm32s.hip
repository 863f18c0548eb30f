// Small-M variant of the exact-fp32 GEMM (contract of k_gemm32.hip) for the across-slice stage: M = B x (1 + D) is a few dozen
// to a few hundred rows, where the 128 x 128 LDS-tiled kernel runs 9-27 workgroups of 24 serial k-steps (36-44 us, latency-bound).
// Here one WAVE owns one 32 x 32 output tile: (M/32) x (N/32) single-wave workgroups, operands straight from global memory into
// registers (16-byte loads, everything is L2-resident), v_mfma_f32_32x32x2_f32.  Lane (row l&31, half l>>5) loads k = 8t + 4 half
// .. +3 of its A row and of its W row; MFMA e of block t multiplies element e of both, i.e. pairs k = 8t + e with k = 8t + 4 + e:
// any pairing that is the same for A and W sums the same products (in a different order than k_gemm32.hip: still exact fp32 FMAs).
#include "mst_common.h"

namespace {

template <int EPI>
__global__ __launch_bounds__(64) void gemm32s_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                     int64_t ldw, const float* __restrict__ bias, float* C, int64_t ldc, int M,
                                                     int N, int K, const float* __restrict__ gamma, float col_scale,
                                                     int scale_cols) {
    const int lane = threadIdx.x, r32 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int m = m0 + r32, n = n0 + r32;
    const float* ap = A + (int64_t)(m < M ? m : M - 1) * lda + 4 * hi;
    const float* wp = W + (int64_t)(n < N ? n : N - 1) * ldw + 4 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // D = W_tile . A_tile^T: rows of D = output columns n, columns of D = rows m (a lane then owns 4 consecutive n per quad)
    for (int k0 = 0; k0 < K; k0 += 32) {              // K % 16 == 0; blocks of 8 k per operand load
        f32x4 av[4], wv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 8 * t;
            av[t] = k < K ? *reinterpret_cast<const f32x4*>(ap + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
            wv[t] = k < K ? *reinterpret_cast<const f32x4*>(wp + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][e], av[t][e], acc, 0, 0, 0);
    }
    if (m >= M) return;
    // D[row = n_local = (r&3) + 8 (r>>2) + 4 hi][col = m_local = lane & 31]
    const bool vec_ok = (ldc % 4 == 0);
    constexpr bool RES = (EPI == MST_EPI_RESIDUAL || EPI == MST_EPI_RESIDUAL_RELU);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nn = n0 + 8 * q + 4 * hi;
        if (nn >= N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = acc[q * 4 + e];
            if (nn + e < N) {
                if (bias) t += bias[nn + e];
                if (nn + e < scale_cols) t *= col_scale;
                if (EPI == MST_EPI_BIAS_GELU) t = gelu_erf(t);
                if (EPI == MST_EPI_BIAS_RELU) t = fmaxf(t, 0.f);
                if (RES && gamma) t *= gamma[nn + e];
            }
            v[e] = t;
        }
        float* cp = C + (int64_t)m * ldc + nn;
        if (vec_ok && nn + 3 < N) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (RES) {
                const float4 xv = *reinterpret_cast<const float4*>(cp);
                o.x += xv.x; o.y += xv.y; o.z += xv.z; o.w += xv.w;
                if (EPI == MST_EPI_RESIDUAL_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            }
            *reinterpret_cast<float4*>(cp) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nn + e < N) cp[e] = (EPI == MST_EPI_RESIDUAL_RELU) ? fmaxf(cp[e] + v[e], 0.f) : RES ? cp[e] + v[e] : v[e];
        }
    }
}

template <int EPI>
int launch_t(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, int64_t M,
             int N, int K, const float* gamma, float cs, int sc, hipStream_t s) {
    gemm32s_kernel<EPI><<<dim3((N + 31) / 32, (unsigned)((M + 31) / 32)), dim3(64), 0, s>>>(A, lda, W, ldw, bias, C, ldc, (int)M, N,
                                                                                          K, gamma, cs, sc);
    return mst_check_launch("gemm32s");
}

}  // namespace

bool gemm32_small_applicable(int64_t M, int N, int K) { return M <= 1024 && K % 16 == 0 && (int64_t)((M + 31) / 32) * ((N + 31) / 32) <= 65535 * 8; }

int launch_gemm32_small(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                        int64_t M, int N, int K, int epi, const float* gamma, float col_scale, int scale_cols, hipStream_t s) {
    switch (epi) {
        case MST_EPI_BIAS: return launch_t<MST_EPI_BIAS>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_BIAS_GELU: return launch_t<MST_EPI_BIAS_GELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_BIAS_RELU: return launch_t<MST_EPI_BIAS_RELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_RESIDUAL: return launch_t<MST_EPI_RESIDUAL>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_RESIDUAL_RELU: return launch_t<MST_EPI_RESIDUAL_RELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
    }
    mst_set_error("gemm32s: bad epilogue %d", epi);
    return MST_EINVAL;
}
