// Convolutional backbone pieces for the ResNet models (reference mst/models/resnet.py:44-50,127-243 build on
// torchvision's resnet34, whose source is not part of the reference tree: the published architecture is restated).
// Activations are NHWC fp32 ([n*H*W, C] row-major), so that a convolution is  im2col -> GEMM(A . W^T + folded BatchNorm) with the
// existing exact-fp32 MFMA GEMM and its bias / ReLU / residual epilogues, and its output is the next layer's input as it stands.
//   im2col_nhwc   col[(n, oy, ox)][(ky, kx, c)] = x[n][oy*s - p + ky][ox*s - p + kx][c] (0 outside), K padded with zeros to Kpad
//   maxpool_nhwc  3 x 3, stride 2, padding 1 (torchvision resnet stem)
//   avgpool_nhwc  adaptive average pooling to 1 x 1
#include "mst_common.h"

namespace {

__global__ void im2col_nhwc_kernel(const float* __restrict__ x, int H, int W, int C, int kh, int kw, int stride, int pad, int Ho,
                                   int Wo, int K, int Kpad, int64_t rows, float* __restrict__ col) {
    const int64_t total = rows * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Kpad;
        const int k = (int)(i - r * Kpad);
        float v = 0.f;
        if (k < K) {
            const int c = k % C, kx = (k / C) % kw, ky = k / (C * kw);
            const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho);
            const int64_t n = r / ((int64_t)Wo * Ho);
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((n * H + iy) * W + ix) * C + c];
        }
        col[i] = v;
    }
}

__global__ void maxpool_nhwc_kernel(const float* __restrict__ x, int H, int W, int C, int Ho, int Wo, int64_t total,
                                    float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ox = (int)((i / C) % Wo), oy = (int)((i / ((int64_t)C * Wo)) % Ho);
        const int64_t n = i / ((int64_t)C * Wo * Ho);
        float m = -INFINITY;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                m = fmaxf(m, x[((n * H + iy) * W + ix) * C + c]);
            }
        }
        y[i] = m;
    }
}

// y[n][c] = mean over the HW positions: one workgroup per image, threads over channels
__global__ void avgpool_nhwc_kernel(const float* __restrict__ x, int HW, int C, float* __restrict__ y) {
    const int64_t n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += x[(n * HW + p) * C + c];
        y[n * C + c] = s / (float)HW;
    }
}

inline unsigned cgrid(int64_t n) {
    const int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

int launch_im2col_nhwc(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* col,
                       hipStream_t s) {
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1, K = kh * kw * C;
    MST_CHECK_ARG(Ho > 0 && Wo > 0 && Kpad >= K, "im2col: bad geometry (Ho=%d Wo=%d K=%d Kpad=%d)", Ho, Wo, K, Kpad);
    const int64_t rows = (int64_t)n * Ho * Wo;
    im2col_nhwc_kernel<<<dim3(cgrid(rows * Kpad)), dim3(256), 0, s>>>(x, H, W, C, kh, kw, stride, pad, Ho, Wo, K, Kpad, rows, col);
    return mst_check_launch("im2col_nhwc");
}

int launch_maxpool_nhwc(const float* x, int n, int H, int W, int C, float* y, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)n * Ho * Wo * C;
    maxpool_nhwc_kernel<<<dim3(cgrid(total)), dim3(256), 0, s>>>(x, H, W, C, Ho, Wo, total, y);
    return mst_check_launch("maxpool_nhwc");
}

int launch_avgpool_nhwc(const float* x, int n, int HW, int C, float* y, hipStream_t s) {
    avgpool_nhwc_kernel<<<dim3(n), dim3(256), 0, s>>>(x, HW, C, y);
    return mst_check_launch("avgpool_nhwc");
}
