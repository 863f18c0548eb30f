// fp32 -> 16-bit operand images for the mixed-precision training step (the reference trains under Lightning's
// precision='16-mixed', main_train.py:110-123 -- matrix products on 16-bit operands with fp32 accumulation, everything else fp32).
// Activations, weights and gradients stay fp32 in memory; a GEMM's operands are rounded into scratch images right before it:
//   cvt16        out[r][c]  = T(scale * x[r][c])                                  (row-major copy; A operand, or W as nn.Linear stores it)
//   cvt16 (T)    out[c][r]  = T(scale * x[r][c]),  r < rows_pad zero-filled       (the transposed image: d weight = dY^T . X needs both
//                                                                                   operands K-contiguous along the ROW index)
// The transposed form moves 64 x 64 tiles through LDS: reads are 256-byte row segments, writes 128-byte segments of the output rows.
#include "mst_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void cvt16_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, float scale, T* __restrict__ out,
                                                    int64_t ldo) {
    const int c4 = cols >> 2;                                       // cols % 4 == 0 (checked by the launcher)
    const int64_t n = rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        typedef __attribute__((ext_vector_type(4))) T o4;
        o4 o;
        o[0] = (T)(v.x * scale); o[1] = (T)(v.y * scale); o[2] = (T)(v.z * scale); o[3] = (T)(v.w * scale);
        *reinterpret_cast<o4*>(out + r * ldo + c) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cvt16_t_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, float scale, T* __restrict__ out,
                                                      int64_t ldo, int64_t rows_pad) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty + 4 * i;
        const int64_t r = r0 + rr;
        tile[rr][tx] = (r < rows && c0 + tx < cols) ? x[r * ldx + c0 + tx] * scale : 0.f;   // rows .. rows_pad: zeros
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty + 4 * i;
        const int64_t r = r0 + tx;
        if (c0 + cc < cols && r < rows_pad) out[(int64_t)(c0 + cc) * ldo + r] = (T)tile[tx][cc];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cvt32_kernel(const T* __restrict__ x, int64_t n4, float* __restrict__ out) {
    typedef __attribute__((ext_vector_type(4))) T i4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const i4 v = reinterpret_cast<const i4*>(x)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

}  // namespace

// out[i] = float(x[i]) for n 16-bit values (n % 4 == 0): where an fp32 kernel reads what a 16-bit one wrote (the last activation of the
// 16-bit ResNet backbone before the average pool and Grad-CAM++)
int launch_cvt32(const void* x, int dt, int64_t n, float* out, hipStream_t s) {
    MST_CHECK_ARG(x && out && n > 0 && n % 4 == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)out & 15) == 0, "cvt32: n=%lld must be a multiple of 4, bases aligned", (long long)n);
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (dt == MST_BF16) cvt32_kernel<bf16_t><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)x, n4, out);
    else if (dt == MST_F16) cvt32_kernel<f16_t><<<dim3(grid), dim3(256), 0, s>>>((const f16_t*)x, n4, out);
    else { mst_set_error("cvt32: input dtype %d (bf16 / f16)", dt); return MST_EINVAL; }
    return mst_check_launch("cvt32");
}

int launch_cvt16(const float* x, int64_t ldx, int64_t rows, int cols, float scale, void* out, int dt, int64_t ldo, int transpose,
                 int64_t rows_pad, hipStream_t s) {
    MST_CHECK_ARG(x && out && rows > 0 && cols > 0, "cvt16: bad arguments");
    MST_CHECK_ARG(dt == MST_BF16 || dt == MST_F16, "cvt16: output dtype %d (bf16 / f16)", dt);
    if (transpose) {
        MST_CHECK_ARG(rows_pad >= rows && ldo >= rows_pad, "cvt16: rows_pad=%lld ldo=%lld", (long long)rows_pad, (long long)ldo);
        const dim3 grid((unsigned)((rows_pad + 63) / 64), (cols + 63) / 64);
        MST_CHECK_ARG(grid.y <= 65535, "cvt16: cols=%d", cols);
        if (dt == MST_BF16) cvt16_t_kernel<bf16_t><<<grid, dim3(256), 0, s>>>(x, ldx, rows, cols, scale, (bf16_t*)out, ldo, rows_pad);
        else cvt16_t_kernel<f16_t><<<grid, dim3(256), 0, s>>>(x, ldx, rows, cols, scale, (f16_t*)out, ldo, rows_pad);
        return mst_check_launch("cvt16_t");
    }
    MST_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0,
                  "cvt16: cols=%d and the row pitches must be multiples of 4, the bases aligned", cols);
    const int64_t n = rows * (cols / 4);
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dt == MST_BF16) cvt16_kernel<bf16_t><<<dim3(grid), dim3(256), 0, s>>>(x, ldx, rows, cols, scale, (bf16_t*)out, ldo);
    else cvt16_kernel<f16_t><<<dim3(grid), dim3(256), 0, s>>>(x, ldx, rows, cols, scale, (f16_t*)out, ldo);
    return mst_check_launch("cvt16");
}
