// C[M,N] = epi(A[M,K] . W[N,K]^T + bias) in exact fp32: v_mfma_f32_32x32x2_f32 (bit-for-bit a
// k-ordered fmaf chain, 1/16 of the bf16 MFMA rate; gfx950 has no TF32/xf32 path).
//
// Used for (a) the across-slice transformer and head, whose GEMMs are tiny and latency-bound, in
// every precision mode, and (b) the whole encoder in MST_F32 ("exact parity") mode.
// Tile 128x128x16, 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32; register-staged loads with
// full bounds checks (any M, N; K % 16 == 0); LDS rows padded to 17 dwords so the per-lane
// ds_read_b32 fragment reads (row = lane&31, k = lane>>5) are bank-conflict-free.
//
// Reference arithmetic: F.linear at transformer_blocks.py:166,283,586; dino.py:135,166 (and the
// encoder linears in fp32 mode).
#include <stdlib.h>

#include "mst_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDP = 17;

// CONV: implicit-GEMM convolution.  A is not a matrix but the NHWC activation x [n, H, W, Cin]: row m = output pixel (img, oy, ox),
// column k = (ky, kx, c) with c fastest -- what mst_im2col_nhwc would have written, gathered on the fly (a 16-wide k-step stays inside
// one filter tap because Cin % 16 == 0, so the tap is a scalar per k-step and a thread's float4 is four consecutive channels of
// one input pixel, or zero outside the image).  No [rows, kh*kw*Cin] matrix exists in memory (9x the activation for a 3x3 layer).
struct ConvGeom { int H, W, C, kh, kw, stride, pad, Ho, Wo, dshift; };    // dshift: log2 of the INPUT dilation (0; 1 = the stride-2 layers' d input)

template <int EPI, bool CONV = false>
__global__ __launch_bounds__(256) void gemm32_kernel(const float* __restrict__ A, int64_t lda,
                                                     const float* __restrict__ W, int64_t ldw,
                                                     const float* __restrict__ bias, float* C,
                                                     int64_t ldc, int M, int N, int K,
                                                     const float* __restrict__ gamma, float col_scale,
                                                     int scale_cols, int tiles_n, int nwg, ConvGeom cg = ConvGeom{}) {
    __shared__ float As[BM * LDP];
    __shared__ float Ws[BN * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: 2 float4 per operand per thread
    const int srow0 = tid >> 2, skq = (tid & 3) * 4;  // rows srow0 and srow0 + 64
    float4 ra[2], rw[2];
    int c_img[2] = {0, 0}, c_y[2] = {0, 0}, c_x[2] = {0, 0};       // CONV: this thread's two output pixels
    if constexpr (CONV) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = m0 + srow0 + u * 64;
            const int hw = cg.Ho * cg.Wo;
            const int mm = m < M ? m : M - 1;
            c_img[u] = mm / hw;
            const int rem = mm - c_img[u] * hw;
            c_y[u] = (rem / cg.Wo) * cg.stride - cg.pad;
            c_x[u] = (rem % cg.Wo) * cg.stride - cg.pad;
        }
    }
    auto gload = [&](int k0) {
        int ky = 0, kx = 0, c0 = 0;
        bool tap_ok = true;
        if constexpr (CONV) {
            const int tap = k0 / cg.C;                   // scalar: the whole k-step lies in one filter tap
            c0 = k0 - tap * cg.C + skq;
            ky = tap / cg.kw;
            kx = tap - ky * cg.kw;
            tap_ok = tap < cg.kh * cg.kw;                // K is padded to a multiple of 16 with zero weights
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = srow0 + u * 64;
            if constexpr (CONV) {
                // (d input of a strided layer: the gradient rows sit on every 2^dshift-th position of the window grid)
                const int ny = c_y[u] + ky, nx = c_x[u] + kx, dmask = (1 << cg.dshift) - 1;
                const int iy = ny >> cg.dshift, ix = nx >> cg.dshift;
                const bool ok = tap_ok && m0 + row < M && ny >= 0 && nx >= 0 && ((ny | nx) & dmask) == 0 && iy < cg.H && ix < cg.W;
                ra[u] = ok ? *reinterpret_cast<const float4*>(A + (((int64_t)c_img[u] * cg.H + iy) * cg.W + ix) * cg.C + c0)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            } else
            ra[u] = (m0 + row < M) ? *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * lda + k0 + skq)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            rw[u] = (n0 + row < N) ? *reinterpret_cast<const float4*>(W + (int64_t)(n0 + row) * ldw + k0 + skq)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* pa = As + (srow0 + u * 64) * LDP + skq;
            float* pw = Ws + (srow0 + u * 64) * LDP + skq;
            pa[0] = ra[u].x; pa[1] = ra[u].y; pa[2] = ra[u].z; pa[3] = ra[u].w;
            pw[0] = rw[u].x; pw[1] = rw[u].y; pw[2] = rw[u].z; pw[3] = rw[u].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    gload(0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();  // previous tile fully consumed
        lstore();
        __syncthreads();
        if (t + 1 < nk) gload((t + 1) * BK);
        const float* ap = As + (wm * 64 + (lane & 31)) * LDP + (lane >> 5);
        const float* wp = Ws + (wn * 64 + (lane & 31)) * LDP + (lane >> 5);
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const float a0 = ap[kp * 2], a1 = ap[32 * LDP + kp * 2];
            const float w0 = wp[kp * 2], w1 = wp[32 * LDP + kp * 2];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, a1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, a1, acc[1][1], 0, 0, 0);
        }
    }

    // epilogue: D[row = n_local = (r&3)+8*(r>>2)+4*(lane>>5)][col = m_local = lane&31]
    const bool vec_ok = (ldc % 4 == 0);
    constexpr bool RES = (EPI == MST_EPI_RESIDUAL || EPI == MST_EPI_RESIDUAL_RELU);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][g * 4 + e];
                    if (n + e < N) {
                        if (bias) t += bias[n + e];
                        if (n + e < scale_cols) t *= col_scale;
                        if (EPI == MST_EPI_BIAS_GELU) t = gelu_erf(t);
                        if (EPI == MST_EPI_BIAS_RELU) t = fmaxf(t, 0.f);
                        if (RES && gamma) t *= gamma[n + e];
                    }
                    v[e] = t;
                }
                float* cp = C + (int64_t)m * ldc + n;
                if (vec_ok && n + 3 < N) {
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
                        if (n + e < N) cp[e] = (EPI == MST_EPI_RESIDUAL_RELU) ? fmaxf(cp[e] + v[e], 0.f) : RES ? cp[e] + v[e] : v[e];
                }
            }
        }
}

template <int EPI>
int launch_t(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
             int64_t ldc, int64_t M, int N, int K, const float* gamma, float cs, int sc, hipStream_t s) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    gemm32_kernel<EPI><<<dim3(nwg), dim3(256), 0, s>>>(A, lda, W, ldw, bias, C, ldc, (int)M, N, K, gamma, cs, sc,
                                                        tiles_n, nwg);
    return mst_check_launch("gemm32");
}

template <int EPI>
int launch_conv_t(const float* x, const ConvGeom& cg, int n, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, int N,
                  int K, const float* gamma, hipStream_t s) {
    const int64_t M = (int64_t)n * cg.Ho * cg.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    gemm32_kernel<EPI, true><<<dim3(nwg), dim3(256), 0, s>>>(x, 0, W, ldw, bias, C, ldc, (int)M, N, K, gamma, 1.f, 0, tiles_n, nwg, cg);
    return mst_check_launch("conv_gemm32");
}

}  // namespace

// out[(img, oy, ox)][co] = epi(sum_{ky,kx,c} x[img][oy*stride-pad+ky][ox*stride-pad+kx][c] * Wg[co][(ky,kx,c)] + bias[co]): the
// convolution as an implicit GEMM (no im2col matrix).  Wg [Cout, Kpad] as for mst_im2col_nhwc + mst_gemm; Cin % 16 == 0.
int launch_conv_gemm32(const float* x, int n, int H, int W_, int Cin, int kh, int kw, int stride, int pad, const float* Wg, int64_t ldw,
                       const float* bias, float* out, int64_t ldc, int Cout, int Kpad, int epi, const float* gamma, hipStream_t s) {
    MST_CHECK_ARG(x && Wg && out && n > 0 && H > 0 && W_ > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "conv_gemm: bad arguments");
    MST_CHECK_ARG(Cin % 16 == 0, "conv_gemm: Cin=%d must be a multiple of 16 (use mst_im2col_nhwc + mst_gemm otherwise)", Cin);
    MST_CHECK_ARG(Kpad >= kh * kw * Cin && Kpad % BK == 0 && ldw % 4 == 0 && ldw >= Kpad, "conv_gemm: Kpad=%d / ldw", Kpad);
    ConvGeom cg{H, W_, Cin, kh, kw, stride, pad, (H + 2 * pad - kh) / stride + 1, (W_ + 2 * pad - kw) / stride + 1, 0};
    MST_CHECK_ARG(cg.Ho > 0 && cg.Wo > 0 && (int64_t)n * cg.Ho * cg.Wo < (1ll << 31) - BM, "conv_gemm: output %d x %d x %d", n, cg.Ho, cg.Wo);
    switch (epi) {
        case MST_EPI_BIAS: return launch_conv_t<MST_EPI_BIAS>(x, cg, n, Wg, ldw, bias, out, ldc, Cout, Kpad, gamma, s);
        case MST_EPI_BIAS_RELU: return launch_conv_t<MST_EPI_BIAS_RELU>(x, cg, n, Wg, ldw, bias, out, ldc, Cout, Kpad, gamma, s);
        case MST_EPI_RESIDUAL: return launch_conv_t<MST_EPI_RESIDUAL>(x, cg, n, Wg, ldw, bias, out, ldc, Cout, Kpad, gamma, s);
        case MST_EPI_RESIDUAL_RELU: return launch_conv_t<MST_EPI_RESIDUAL_RELU>(x, cg, n, Wg, ldw, bias, out, ldc, Cout, Kpad, gamma, s);
    }
    mst_set_error("conv_gemm: epilogue %d unsupported (bias, bias + ReLU, residual, residual + ReLU)", epi);
    return MST_EINVAL;
}

// d input of a convolution as a convolution: dx[img][y][x][c] = sum_{ky',kx',co} dz[img][(y - pt + ky') / s][(x - pt + kx') / s][co] * Wt[c][(ky',kx',co)]
// over the positions the division is exact for, pt = k - 1 - pad, Wt[c][(ky',kx',co)] = W[co][c][k-1-ky'][k-1-kx'] (the caller flips and
// transposes the weight).  The same implicit GEMM with stride 1, padding pt and the gradient rows dilated by the forward stride: no
// [rows, kh*kw*Cin] gradient matrix, no scatter with atomics.  dz [n,Ho,Wo,Cout] (Cout % 16 == 0), dx [n*H*W, Cin] overwritten.
int launch_conv_dgrad32(const float* dz, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const float* Wt, int H, int W_,
                        int Cin, float* dx, hipStream_t s) {
    MST_CHECK_ARG(dz && Wt && dx && n > 0 && Ho > 0 && Wo > 0 && H > 0 && W_ > 0 && kh == kw && kh > 0 && pad >= 0 && pad < kh, "conv_dgrad: bad arguments");
    MST_CHECK_ARG(stride == 1 || stride == 2, "conv_dgrad: stride %d (1 or 2)", stride);
    MST_CHECK_ARG(Cout % 16 == 0, "conv_dgrad: Cout=%d must be a multiple of 16", Cout);
    MST_CHECK_ARG((H + 2 * pad - kh) / stride + 1 == Ho && (W_ + 2 * pad - kw) / stride + 1 == Wo, "conv_dgrad: %d x %d is not the output of a %d x %d input", Ho, Wo, H, W_);
    MST_CHECK_ARG((int64_t)n * H * W_ < (1ll << 31) - BM, "conv_dgrad: %d x %d x %d", n, H, W_);
    ConvGeom cg{Ho, Wo, Cout, kh, kw, 1, kh - 1 - pad, H, W_, stride - 1};
    const int K = kh * kw * Cout;
    return launch_conv_t<MST_EPI_BIAS>(dz, cg, n, Wt, K, nullptr, dx, Cin, Cin, K, nullptr, s);
}

int launch_gemm32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                  float* C, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                  float col_scale, int scale_cols, hipStream_t s) {
    MST_CHECK_ARG(K > 0 && K % BK == 0, "gemm32: K=%d must be a multiple of %d", K, BK);
    MST_CHECK_ARG(N > 0, "gemm32: N=%d", N);
    MST_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0, "gemm32: lda/ldw must be multiples of 4");
    MST_CHECK_ARG(M < (1ll << 31) - BM, "gemm32: M too large");
    if (M <= 0) return MST_OK;
    static const bool small_ok = !(getenv("MST_GEMM32_SMALL") && atoi(getenv("MST_GEMM32_SMALL")) == 0);
    if (small_ok && gemm32_small_applicable(M, N, K))   // across-slice stage: a few hundred rows (k_gemm32s.hip)
        return launch_gemm32_small(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    switch (epi) {
        case MST_EPI_BIAS: return launch_t<MST_EPI_BIAS>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_BIAS_GELU: return launch_t<MST_EPI_BIAS_GELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_BIAS_RELU: return launch_t<MST_EPI_BIAS_RELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_RESIDUAL: return launch_t<MST_EPI_RESIDUAL>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
        case MST_EPI_RESIDUAL_RELU: return launch_t<MST_EPI_RESIDUAL_RELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, col_scale, scale_cols, s);
    }
    mst_set_error("gemm32: bad epilogue %d", epi);
    return MST_EINVAL;
}
