// Convolutional backbone pieces for the ResNet models (reference mst/models/resnet.py:44-50,127-243 build on
// torchvision's resnet34, whose source is not part of the reference tree: the published architecture is restated).
// Activations are NHWC fp32 ([n*H*W, C] row-major), so that a convolution is  im2col -> GEMM(A . W^T + folded BatchNorm) with the
// existing exact-fp32 MFMA GEMM and its bias / ReLU / residual epilogues, and its output is the next layer's input as it stands.
//   im2col_nhwc   col[(n, oy, ox)][(ky, kx, c)] = x[n][oy*s - p + ky][ox*s - p + kx][c] (0 outside), K padded with zeros to Kpad
//   maxpool_nhwc  3 x 3, stride 2, padding 1 (torchvision resnet stem)
//   avgpool_nhwc  adaptive average pooling to 1 x 1
#include "mst_common.h"

namespace {

template <typename OutT>
__global__ void im2col_nhwc_kernel(const float* __restrict__ x, int H, int W, int C, int kh, int kw, int stride, int pad, int Ho,
                                   int Wo, int K, int Kpad, int64_t rows, OutT* __restrict__ col) {
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
        col[i] = (OutT)v;
    }
}

template <typename T>
__global__ void maxpool_nhwc_kernel(const T* __restrict__ x, int H, int W, int C, int Ho, int Wo, int64_t total,
                                    T* __restrict__ y) {
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
                m = fmaxf(m, (float)x[((n * H + iy) * W + ix) * C + c]);
            }
        }
        y[i] = (T)m;
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

// ---- training step of the backbone: BatchNorm with batch statistics (nn.BatchNorm2d in train mode) and the backward of the
// convolution / pooling layers; activations stay [rows = n*H*W, C] fp32.
// out[c] += sum_r (z[r][c] - mean[c])^2
__global__ void colsqdev_kernel(const float* __restrict__ z, const float* __restrict__ mean, int64_t rows, int C, int rpb,
                                float* __restrict__ out) {
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const float m = mean[c];
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) { const float d = z[r * C + c] - m; s = fmaf(d, d, s); }
        atomicAdd(out + c, s);
    }
}
// stage 0: mean = sum / rows.  stage 1: rstd = rsqrt(sqdev / rows + eps); running statistics as nn.BatchNorm2d updates them
// (momentum, unbiased variance)
__global__ void bn_finalize_kernel(int stage, const float* __restrict__ acc, int64_t rows, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (stage == 0) { mean[c] = acc[c] / (float)rows; return; }
    const float var = acc[c] / (float)rows;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)rows / (float)(rows > 1 ? rows - 1 : 1));
}
// y = gamma (z - mean) rstd + beta (+ res) (then ReLU)
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, const float* res, int relu,
                                int64_t total, int C, float* y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = gamma[c] * (z[i] - mean[c]) * rstd[c] + beta[c];
        if (res) v += res[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// dbeta[c] += sum_r dy, dgamma[c] += sum_r dy xhat
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ dy, int64_t rows, int C, int rpb, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const float m = mean[c], rs = rstd[c];
        float sg = 0.f, sb = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            const float d = dy[r * C + c];
            sb += d;
            sg = fmaf(d, (z[r * C + c] - m) * rs, sg);
        }
        atomicAdd(dgamma + c, sg);
        atomicAdd(dbeta + c, sb);
    }
}
// dz = gamma rstd (dy - dbeta / rows - xhat dgamma / rows)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dy, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, int64_t rows, int C, float* __restrict__ dz) {
    const int64_t total = rows * C;
    const float inv = 1.0f / (float)rows;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xh = (z[i] - mean[c]) * rstd[c];
        dz[i] = gamma[c] * rstd[c] * (dy[i] - dbeta[c] * inv - xh * dgamma[c] * inv);
    }
}
// adjoint of im2col_nhwc: dx[n][iy][ix][c] += dcol[(n,oy,ox)][(ky,kx,c)]
__global__ void col2im_nhwc_kernel(const float* __restrict__ dcol, int H, int W, int C, int kh, int kw, int stride, int pad, int Ho,
                                   int Wo, int K, int Kpad, int64_t rows, float* __restrict__ dx) {
    const int64_t total = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const int c = k % C, kx = (k / C) % kw, ky = k / (C * kw);
        const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho);
        const int64_t n = r / ((int64_t)Wo * Ho);
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) atomicAdd(dx + ((n * H + iy) * W + ix) * C + c, dcol[r * Kpad + k]);
    }
}
// max-pool backward: the gradient goes to the first maximum of the window (scan order ky, kx: torch's choice)
__global__ void maxpool_bwd_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ dy, int H, int W, int C, int Ho, int Wo,
                                        int64_t total, float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ox = (int)((i / C) % Wo), oy = (int)((i / ((int64_t)C * Wo)) % Ho);
        const int64_t n = i / ((int64_t)C * Wo * Ho);
        float m = -INFINITY;
        int64_t arg = -1;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const int64_t j = ((n * H + iy) * W + ix) * C + c;
                const float v = x[j];
                if (v > m) { m = v; arg = j; }
            }
        }
        if (arg >= 0) atomicAdd(dx + arg, dy[i]);
    }
}
__global__ void avgpool_bwd_nhwc_kernel(const float* __restrict__ dy, int HW, int C, int64_t total, float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t n = i / ((int64_t)C * HW);
        dx[i] = dy[n * C + c] / (float)HW;
    }
}

// Grad-CAM++ of the LAST ReLU output (reference resnet.py:66-68 + 93-118; the only map its getter exposes).  The loss is the sum of
// every image's largest output, so d loss / d pooled feature c = W[argmax][c] (or [c == argmax] without an fc); behind the global
// average pool the gradient of every position of channel c is that / HW, so the weights of equation 19 have a closed form per (n, c).
// state[0] = min, state[1] = max of the rectified maps (as uint bit patterns: the values are >= 0).
__global__ void gradcampp_kernel(const float* __restrict__ act, const float* __restrict__ out, int O, const float* __restrict__ W,
                                 int HW, int C, float* __restrict__ cam, unsigned* __restrict__ state) {
    extern __shared__ float wsh[];                                      // [C] weights of this image
    __shared__ int arg_sh;
    const int n = blockIdx.x;
    const float* a = act + (int64_t)n * HW * C;
    if (threadIdx.x == 0) {                                             // first maximum, as torch.argmax
        int arg = 0;
        float m = out[(int64_t)n * O];
        for (int o = 1; o < O; ++o) { const float v = out[(int64_t)n * O + o]; if (v > m) { m = v; arg = o; } }
        arg_sh = arg;
    }
    __syncthreads();
    const int arg = arg_sh;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sa = 0.f;
        for (int p = 0; p < HW; ++p) sa += a[(int64_t)p * C + c];
        const float gr = (W ? W[(int64_t)arg * C + c] : (c == arg ? 1.f : 0.f)) / (float)HW;
        const float g2 = gr * gr, g3 = g2 * gr;
        float den = 2.f * g2 + sa * g3 + 1e-6f;
        if (den == 0.f) den = 1.f;
        wsh[c] = (float)HW * fmaxf(gr, 0.f) * (g2 / den);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(wsh[c], a[(int64_t)p * C + c], s);
        s = fmaxf(s, 0.f);
        cam[(int64_t)n * HW + p] = s;
        atomicMin(state + 0, __float_as_uint(s));
        atomicMax(state + 1, __float_as_uint(s));
    }
}
__global__ void gradcam_norm_kernel(float* cam, int64_t total, const float* __restrict__ state) {
    const float mn = state[0], mx = state[1] - state[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        cam[i] = (cam[i] - mn) / mx;
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
    im2col_nhwc_kernel<float><<<dim3(cgrid(rows * Kpad)), dim3(256), 0, s>>>(x, H, W, C, kh, kw, stride, pad, Ho, Wo, K, Kpad, rows, col);
    return mst_check_launch("im2col_nhwc");
}

// the same rows rounded to a 16-bit type on the way out (the stem of the 16-bit inference backbone: its im2col rows are the "pixels" of a
// 1 x 1 mst_conv_gemm16) and the 3 x 3 / 2 max pool on 16-bit activations
int launch_im2col_nhwc16(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, void* col, int dt, hipStream_t s) {
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1, K = kh * kw * C;
    MST_CHECK_ARG(Ho > 0 && Wo > 0 && Kpad >= K, "im2col16: bad geometry (Ho=%d Wo=%d K=%d Kpad=%d)", Ho, Wo, K, Kpad);
    const int64_t rows = (int64_t)n * Ho * Wo;
    if (dt == MST_BF16) im2col_nhwc_kernel<bf16_t><<<dim3(cgrid(rows * Kpad)), dim3(256), 0, s>>>(x, H, W, C, kh, kw, stride, pad, Ho, Wo, K, Kpad, rows, (bf16_t*)col);
    else if (dt == MST_F16) im2col_nhwc_kernel<f16_t><<<dim3(cgrid(rows * Kpad)), dim3(256), 0, s>>>(x, H, W, C, kh, kw, stride, pad, Ho, Wo, K, Kpad, rows, (f16_t*)col);
    else { mst_set_error("im2col16: output dtype %d (bf16 / f16)", dt); return MST_EINVAL; }
    return mst_check_launch("im2col_nhwc16");
}

int launch_maxpool_nhwc16(const void* x, int dt, int n, int H, int W, int C, void* y, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)n * Ho * Wo * C;
    if (dt == MST_BF16) maxpool_nhwc_kernel<bf16_t><<<dim3(cgrid(total)), dim3(256), 0, s>>>((const bf16_t*)x, H, W, C, Ho, Wo, total, (bf16_t*)y);
    else if (dt == MST_F16) maxpool_nhwc_kernel<f16_t><<<dim3(cgrid(total)), dim3(256), 0, s>>>((const f16_t*)x, H, W, C, Ho, Wo, total, (f16_t*)y);
    else { mst_set_error("maxpool16: dtype %d (bf16 / f16)", dt); return MST_EINVAL; }
    return mst_check_launch("maxpool_nhwc16");
}

int launch_maxpool_nhwc(const float* x, int n, int H, int W, int C, float* y, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)n * Ho * Wo * C;
    maxpool_nhwc_kernel<float><<<dim3(cgrid(total)), dim3(256), 0, s>>>(x, H, W, C, Ho, Wo, total, y);
    return mst_check_launch("maxpool_nhwc");
}

int launch_avgpool_nhwc(const float* x, int n, int HW, int C, float* y, hipStream_t s) {
    avgpool_nhwc_kernel<<<dim3(n), dim3(256), 0, s>>>(x, HW, C, y);
    return mst_check_launch("avgpool_nhwc");
}

int launch_colsqdev(const float* z, const float* mean, int64_t rows, int C, float* out, hipStream_t s) {
    const int rpb = 256;
    colsqdev_kernel<<<dim3((C + 255) / 256, (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s>>>(z, mean, rows, C, rpb, out);
    return mst_check_launch("colsqdev");
}
int launch_bn_finalize(int stage, const float* acc, int64_t rows, int C, float eps, float momentum, float* mean, float* rstd,
                       float* running_mean, float* running_var, hipStream_t s) {
    bn_finalize_kernel<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(stage, acc, rows, C, eps, momentum, mean, rstd, running_mean, running_var);
    return mst_check_launch("bn_finalize");
}
int launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* res,
                    int relu, int64_t rows, int C, float* y, hipStream_t s) {
    bn_apply_kernel<<<dim3(cgrid(rows * C)), dim3(256), 0, s>>>(z, mean, rstd, gamma, beta, res, relu, rows * C, C, y);
    return mst_check_launch("bn_apply");
}
int launch_bn_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* dy, int64_t rows, int C,
                  float* dgamma, float* dbeta, float* dz, hipStream_t s) {
    const int rpb = 256;
    bn_bwd_reduce_kernel<<<dim3((C + 255) / 256, (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s>>>(z, mean, rstd, dy, rows, C, rpb, dgamma, dbeta);
    int rc = mst_check_launch("bn_bwd_reduce");
    if (rc) return rc;
    bn_bwd_apply_kernel<<<dim3(cgrid(rows * C)), dim3(256), 0, s>>>(z, mean, rstd, gamma, dy, dgamma, dbeta, rows, C, dz);
    return mst_check_launch("bn_bwd_apply");
}
int launch_col2im_nhwc(const float* dcol, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* dx,
                       hipStream_t s) {
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1, K = kh * kw * C;
    const int64_t rows = (int64_t)n * Ho * Wo;
    col2im_nhwc_kernel<<<dim3(cgrid(rows * K)), dim3(256), 0, s>>>(dcol, H, W, C, kh, kw, stride, pad, Ho, Wo, K, Kpad, rows, dx);
    return mst_check_launch("col2im_nhwc");
}
int launch_maxpool_bwd_nhwc(const float* x, const float* dy, int n, int H, int W, int C, float* dx, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)n * Ho * Wo * C;
    maxpool_bwd_nhwc_kernel<<<dim3(cgrid(total)), dim3(256), 0, s>>>(x, dy, H, W, C, Ho, Wo, total, dx);
    return mst_check_launch("maxpool_bwd_nhwc");
}
int launch_avgpool_bwd_nhwc(const float* dy, int n, int HW, int C, float* dx, hipStream_t s) {
    const int64_t total = (int64_t)n * HW * C;
    avgpool_bwd_nhwc_kernel<<<dim3(cgrid(total)), dim3(256), 0, s>>>(dy, HW, C, total, dx);
    return mst_check_launch("avgpool_bwd_nhwc");
}

int launch_gradcampp(const float* act, const float* out, int O, const float* W, int n, int HW, int C, float* cam, float* state,
                     hipStream_t s) {
    const unsigned init[2] = {0x7f800000u, 0u};                          // +inf, 0
    if (hipMemcpyAsync(state, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) { mst_set_error("gradcampp: state init failed"); return MST_ELAUNCH; }
    gradcampp_kernel<<<dim3(n), dim3(256), sizeof(float) * C, s>>>(act, out, O, W, HW, C, cam, (unsigned*)state);
    int rc = mst_check_launch("gradcampp");
    if (rc) return rc;
    gradcam_norm_kernel<<<dim3(cgrid((int64_t)n * HW)), dim3(256), 0, s>>>(cam, (int64_t)n * HW, state);
    return mst_check_launch("gradcam_norm");
}
