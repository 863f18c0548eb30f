// LayerNorm over the last dim, one 64-lane wave per row, statistics by wave shuffles (two-pass in
// registers: mean, then variance of the centred values).  HBM-bound: reads 4*cols bytes, writes
// 2*cols (16-bit out) or 4*cols per row.
// Reference arithmetic: nn.LayerNorm at block.py:63,75 / vision_transformer.py:165 (eps 1e-6) and
// transformer_blocks.py:499-500 / dino.py:95 (eps 1e-5).
#include "mst_common.h"

template <int NJ, typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t xs,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b,
                                                        OutT* __restrict__ out, int64_t os,
                                                        int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * xs;
    float2 v[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = j * 128 + lane * 2;
        if (c < cols) {
            v[j] = *reinterpret_cast<const float2*>(xr + c);
            s += v[j].x + v[j].y;
        } else {
            v[j] = make_float2(0.f, 0.f);
        }
    }
    const float inv_n = 1.0f / (float)cols;
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = j * 128 + lane * 2;
        if (c < cols) {
            const float dx = v[j].x - mean, dy = v[j].y - mean;
            q += dx * dx + dy * dy;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
    OutT* orow = out + row * os;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = j * 128 + lane * 2;
        if (c < cols) {
            float2 gg = make_float2(1.f, 1.f), bb = make_float2(0.f, 0.f);
            if (g) {   // g == b == nullptr: normalise only
                gg = *reinterpret_cast<const float2*>(g + c);
                bb = *reinterpret_cast<const float2*>(b + c);
            }
            const float y0 = (v[j].x - mean) * rstd * gg.x + bb.x;
            const float y1 = (v[j].y - mean) * rstd * gg.y + bb.y;
            if constexpr (sizeof(OutT) == 4) {
                *reinterpret_cast<float2*>(orow + c) = make_float2(y0, y1);
            } else {
                typedef __attribute__((ext_vector_type(2))) OutT o2;
                o2 pk;
                pk[0] = (OutT)y0;
                pk[1] = (OutT)y1;
                *reinterpret_cast<o2*>(orow + c) = pk;
            }
        }
    }
}

template <typename OutT>
static int launch_ln_t(const float* x, int64_t xs, const float* g, const float* b, void* out, int64_t os,
                       int64_t rows, int cols, float eps, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nj = (cols + 127) / 128;
    OutT* o = (OutT*)out;
    if (nj <= 1) layernorm_kernel<1, OutT><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps);
    else if (nj <= 3) layernorm_kernel<3, OutT><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps);
    else if (nj <= 6) layernorm_kernel<6, OutT><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps);
    else if (nj <= 8) layernorm_kernel<8, OutT><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps);
    else layernorm_kernel<16, OutT><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps);     // 2048: the slice transformer behind a bottleneck ResNet
    return mst_check_launch("layernorm");
}

int launch_layernorm(const float* x, int64_t xs, const float* g, const float* b, void* out, int odt,
                     int64_t os, int64_t rows, int cols, float eps, hipStream_t s) {
    MST_CHECK_ARG(cols > 0 && cols <= 2048 && (cols % 2) == 0, "layernorm: cols=%d must be even and <= 2048", cols);
    MST_CHECK_ARG((xs % 2) == 0 && (os % 2) == 0, "layernorm: row strides must be even");
    if (rows <= 0) return MST_OK;
    switch (odt) {
        case MST_F32: return launch_ln_t<float>(x, xs, g, b, out, os, rows, cols, eps, s);
        case MST_F16: return launch_ln_t<f16_t>(x, xs, g, b, out, os, rows, cols, eps, s);
        case MST_BF16: return launch_ln_t<bf16_t>(x, xs, g, b, out, os, rows, cols, eps, s);
    }
    mst_set_error("layernorm: bad out dtype %d", odt);
    return MST_EINVAL;
}

// LayerNorm writing OCP e4m3 bytes under a calibrated per-tensor scale (fp8 mode with static scales: the quantisation of the
// next GEMM's input is free here -- 1 byte per element written instead of 2, no scan, no second pass).
// Two rows per wave: both rows' loads are issued before the first reduction, which doubles the bytes in flight per wave (one row
// is only 1.5 KB at E = 384; with 32 waves per CU that is less than the chip's bandwidth-latency product).
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_f8_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ g,
                                                           const float* __restrict__ b, uint8_t* __restrict__ out, int64_t os,
                                                           int64_t rows, int cols, float eps, const float* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;                       // wave-uniform
    const float* xr[2] = {x + row0 * xs, x + (two ? row0 + 1 : row0) * xs};
    float2 v[2][NJ];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * 128 + lane * 2;
            if (c < cols) {
                v[r][j] = *reinterpret_cast<const float2*>(xr[r] + c);
                s[r] += v[r][j].x + v[r][j].y;
            } else {
                v[r][j] = make_float2(0.f, 0.f);
            }
        }
    const float inv_n = 1.0f / (float)cols;
    const float am = *amax;
    const float inv = am > 0.f ? 448.0f / am : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && !two) break;
        const float mean = wave_sum(s[r]) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * 128 + lane * 2;
            if (c < cols) {
                const float dx = v[r][j].x - mean, dy = v[r][j].y - mean;
                q += dx * dx + dy * dy;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        uint8_t* orow = out + (row0 + r) * os;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * 128 + lane * 2;
            if (c < cols) {
                float2 gg = make_float2(1.f, 1.f), bb = make_float2(0.f, 0.f);
                if (g) {
                    gg = *reinterpret_cast<const float2*>(g + c);
                    bb = *reinterpret_cast<const float2*>(b + c);
                }
                const float y0 = __builtin_amdgcn_fmed3f(((v[r][j].x - mean) * rstd * gg.x + bb.x) * inv, -448.0f, 448.0f);
                const float y1 = __builtin_amdgcn_fmed3f(((v[r][j].y - mean) * rstd * gg.y + bb.y) * inv, -448.0f, 448.0f);
                const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(y0, y1, 0, false);
                *reinterpret_cast<unsigned short*>(orow + c) = (unsigned short)(pk & 0xffff);
            }
        }
    }
}

int launch_layernorm_f8(const float* x, int64_t xs, const float* g, const float* b, void* out8, int64_t os, int64_t rows,
                        int cols, float eps, const float* amax, hipStream_t s) {
    MST_CHECK_ARG(x && out8 && amax, "layernorm_fp8: null pointer");
    MST_CHECK_ARG(cols > 0 && cols <= 1024 && (cols % 2) == 0, "layernorm_fp8: cols=%d must be even and <= 1024", cols);
    MST_CHECK_ARG((xs % 2) == 0 && (os % 2) == 0, "layernorm_fp8: row strides must be even");
    if (rows <= 0) return MST_OK;
    const dim3 grid((unsigned)((rows + 7) / 8)), block(256);   // 4 waves x 2 rows
    const int nj = (cols + 127) / 128;
    uint8_t* o = (uint8_t*)out8;
    if (nj <= 1) layernorm_f8_kernel<1><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps, amax);
    else if (nj <= 3) layernorm_f8_kernel<3><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps, amax);
    else if (nj <= 6) layernorm_f8_kernel<6><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps, amax);
    else layernorm_f8_kernel<8><<<grid, block, 0, s>>>(x, xs, g, b, o, os, rows, cols, eps, amax);
    return mst_check_launch("layernorm_fp8");
}
