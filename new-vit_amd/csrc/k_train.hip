// Kernels of the training step (the backward pass of DinoV2ClassifierSlice), all fp32 on the exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32): the first correct version -- gradients are checked against autograd of the CPU oracle at the
// fp32 bar.  What the reference gets from torch.autograd (base_model.py:148-181 `_step`; main_train.py:110-126) is spelled out
// here op by op:
//   gemm_ex        (k_gemm_ex.hip) C[b] = alpha * A[b] . B[b] (+ beta * C[b]), every operand with explicit element strides and a two-level
//                  batch: one entry covers dX = dY.W, dW = dY^T.X, and the four products of the attention backward on the packed
//                  q|k|v layout of the forward (attention.py:56-66)
//   softmax_rows / softmax_rows_bwd     P = softmax(S + key-padding mask),  dS = P o (dP - rowsum(dP o P))
//   layernorm_bwd  dx (+ residual gradient), d gamma, d beta   (nn.LayerNorm; block.py:63,75; transformer_blocks.py:499)
//   act_bwd        GELU (erf form, mlp.py:22) / ReLU (transformer_blocks.py:484) derivative times the upstream gradient
//   colsum / colsum_prod / mul_cols     bias and LayerScale gradients (layer_scale.py:25-27)
//   im2col14       the 14x14 patches of a gray volume as rows (patch_embed.py:68-81): d W_patch = dX^T . im2col
//   pos_interp_bwd adjoint of the bicubic position-grid resampling (vision_transformer.py:179-211)
#include "mst_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// P = softmax over the last dim of S [B, H, Lq, L] (in place); mask uint8 [B, L], 1 = key ignored (-inf before the softmax).
__global__ void softmax_rows_kernel(float* __restrict__ S, const uint8_t* __restrict__ mask, int64_t rows, int L, int rows_per_b) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* s = S + row * L;
    const uint8_t* mk = mask ? mask + (row / rows_per_b) * L : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, (mk && mk[j]) ? -INFINITY : s[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float e = (mk && mk[j]) ? 0.f : expf(s[j] - mx);
        s[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < L; j += 64) s[j] *= inv;
}

// dS = scale * P o (dP - rowsum(dP o P)), written over dP.
__global__ void softmax_rows_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int64_t rows, int L, float scale) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + row * L;
    float* d = dP + row * L;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot = fmaf(d[j], p[j], dot);
    dot = wave_sum(dot);
    for (int j = lane; j < L; j += 64) d[j] = scale * p[j] * (d[j] - dot);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  One wave per row at a time, waves stride over the rows; d gamma / d beta partial sums live in registers
// (column c = lane + 64 i, i < CI), meet in LDS at the end and are added to the fp32 outputs with one atomic per column per workgroup.
//   dx[row] = (dres ? dres[row] : 0) + rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),   dyg = dy * gamma
template <int CI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, int64_t dys, const float* dres, int64_t drs,
                                                            float* dx, int64_t dxs, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    float gsum[CI], bsum[CI], gm[CI];
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        gsum[i] = bsum[i] = 0.f;
        const int c = lane + 64 * i;
        gm[i] = (c < cols) ? (gamma ? gamma[c] : 1.f) : 0.f;
    }
    for (int64_t row = w; row < rows; row += nw) {
        const float* xr = x + row * xs;
        const float* dr = dy + row * dys;
        float xv[CI], dv[CI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < cols ? xr[c] : 0.f;
            dv[i] = c < cols ? dr[c] : 0.f;
            s += xv[i];
        }
        const float mean = wave_sum(s) / (float)cols;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const float d = (lane + 64 * i < cols) ? xv[i] - mean : 0.f;
            sq = fmaf(d, d, sq);
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)cols + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const bool in = lane + 64 * i < cols;
            const float xh = in ? (xv[i] - mean) * rstd : 0.f;
            const float dg = dv[i] * gm[i];
            xv[i] = xh;
            a += dg;
            b = fmaf(dg, xh, b);
            gsum[i] = fmaf(dv[i], xh, gsum[i]);
            bsum[i] += dv[i];
        }
        a = wave_sum(a) / (float)cols;
        b = wave_sum(b) / (float)cols;
        if (dx) {
#pragma unroll
            for (int i = 0; i < CI; ++i) {
                const int c = lane + 64 * i;
                if (c < cols) {
                    const float v = rstd * (dv[i] * gm[i] - a - xv[i] * b);
                    dx[row * dxs + c] = (dres ? dres[row * drs + c] : 0.f) + v;
                }
            }
        }
    }
    // the four waves' partial sums meet in LDS; one atomic per column per workgroup
    __shared__ float red[4][2][CI * 64];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        red[wave][0][lane + 64 * i] = gsum[i];
        red[wave][1][lane + 64 * i] = bsum[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        if (dgamma) atomicAdd(dgamma + c, (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]));
        if (dbeta) atomicAdd(dbeta + c, (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE of the across-slice attention (rotary_embedding_torch.py:38-62,159-173; transformer_blocks.py:262-264) on packed q | k | v rows, in
// place: the pairs (2p, 2p+1) of every head's q and k are rotated by sign * position * freqs[p] (position = row % L, the class token
// is position 0).  sign +1: the forward; sign -1: its adjoint on (dq', dk') -- rotations are orthogonal, and the frequencies are buffers
// without a gradient (learned_freq = False).
__global__ void rope_rows_kernel(float* __restrict__ qkv, int64_t rows, int L, int heads, int hd, const float* __restrict__ freqs, float sign) {
    const int half = hd / 2, e = heads * hd;
    const int64_t total = rows * 2 * heads * half;                     // (row, q|k, head, pair)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % half);
        const int h = (int)((i / half) % heads);
        const int which = (int)((i / ((int64_t)half * heads)) % 2);
        const int64_t r = i / ((int64_t)half * heads * 2);
        float sn, cs;
        sincosf(sign * (float)(r % L) * freqs[p], &sn, &cs);
        float* v = qkv + r * 3 * e + which * e + h * hd + 2 * p;
        const float a = v[0], b = v[1];
        v[0] = a * cs - b * sn;
        v[1] = b * cs + a * sn;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dh = dy * act'(h), in place over dy.  kind 0: GELU (erf form), 1: ReLU.
__global__ void act_bwd_kernel(const float* __restrict__ h, float* __restrict__ dy, int64_t n, int kind) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = h[i];
        float d;
        if (kind == 0) d = 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
        else d = v > 0.f ? 1.f : 0.f;
        dy[i] *= d;
    }
}

// y = act(h): the forward twin (the training forward keeps the pre-activation h for the backward).
__global__ void act_fwd_kernel(const float* __restrict__ h, float* __restrict__ y, int64_t n, int kind) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = h[i];
        y[i] = kind == 0 ? gelu_erf(v) : fmaxf(v, 0.f);
    }
}

// out[j] += sum_i a[i][j] * (b ? b[i][j] : 1).  A 256-thread block owns 64 columns x a chunk of rows: 16 threads x float4 across the
// columns (VEC) or 64 threads x 1, the other 16 (4) thread rows stride down the chunk; the partial sums meet in LDS and ONE atomic per
// column per block goes out (the round-2 kernel ran one thread per column down 256 rows: 34 workgroups for a [4112, 384] bias gradient).
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int64_t as, const float* __restrict__ b, int64_t bs, int64_t rows,
                                                     int cols, int rows_per_block, float* __restrict__ out) {
    // row blocks on gridDim.x (2^31 - 1 blocks), column blocks on gridDim.y: the stem's BatchNorm sums of a 2 x 128 x 512^2 ResNet
    // step have 16.8 M rows, more row blocks than gridDim.y takes
    constexpr int CW = VEC ? 4 : 1, TX = 64 / CW, TY = 256 / TX;
    __shared__ float red[TY][64 + 4];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int c = blockIdx.y * 64 + tx * CW;
    float acc[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) acc[e] = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + ty; r < r1; r += TY) {
            if (VEC) {
                const float4 av = *reinterpret_cast<const float4*>(a + r * as + c);
                const float4 bv = b ? *reinterpret_cast<const float4*>(b + r * bs + c) : make_float4(1.f, 1.f, 1.f, 1.f);
                acc[0] = fmaf(av.x, bv.x, acc[0]); acc[CW > 1 ? 1 : 0] = fmaf(av.y, bv.y, acc[CW > 1 ? 1 : 0]);
                acc[CW > 2 ? 2 : 0] = fmaf(av.z, bv.z, acc[CW > 2 ? 2 : 0]); acc[CW > 3 ? 3 : 0] = fmaf(av.w, bv.w, acc[CW > 3 ? 3 : 0]);
            } else {
                acc[0] = fmaf(a[r * as + c], b ? b[r * bs + c] : 1.f, acc[0]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < CW; ++e) red[ty][tx * CW + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.y * 64 + threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < TY; ++y) t += red[y][threadIdx.x];
        if (cc < cols) atomicAdd(out + cc, t);
    }
}

// y[i][j] = alpha * x[i][j] * (g ? g[j] : 1) + beta * y[i][j]   (LayerScale on a gradient; residual sums; plain scaling)
__global__ void axpby_cols_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ g, float alpha, float beta,
                                  float* __restrict__ y, int64_t ys, int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = alpha * x[r * xs + c] * (g ? g[c] : 1.f);
        float* o = y + r * ys + c;
        *o = v + (beta != 0.f ? beta * *o : 0.f);
    }
}

// rows of 14 x 14 pixels: col[(n*Np + p)][ky*14 + kx] = vol[n][py*14 + ky][px*14 + kx]
template <typename T>
__global__ void im2col14_kernel(const T* __restrict__ vol, int H, int W, int gw, int Np, int64_t total, float* __restrict__ col) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / 196;
        const int k = (int)(i - row * 196), ky = k / 14, kx = k - ky * 14;
        const int64_t n = row / Np;
        const int p = (int)(row - n * Np), py = p / gw, px = p - py * gw;
        col[i] = to_f32(vol[(n * H + py * 14 + ky) * W + px * 14 + kx]);
    }
}

__device__ __forceinline__ void cubic_w(float x, float w[4]) {   // F.interpolate bicubic, A = -0.75 (as k_patch.hip)
    const float A = -0.75f;
    float t = x + 1.0f;
    w[0] = ((A * t - 5.0f * A) * t + 8.0f * A) * t - 4.0f * A;
    w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    t = 1.0f - x;
    w[2] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    t = 2.0f - x;
    w[3] = ((A * t - 5.0f * A) * t + 8.0f * A) * t - 4.0f * A;
}
// adjoint of pos_interp_kernel: dpos[yy][xx][e] += wy[a] wx[b] dout[oy][ox][e]
__global__ void pos_interp_bwd_kernel(const float* __restrict__ dout, int M, int E, int gh, int gw, float scale_y, float scale_x,
                                      float* __restrict__ dpos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)gh * gw * E) return;
    const int e = (int)(i % E);
    const int ox = (int)((i / E) % gw);
    const int oy = (int)(i / ((int64_t)E * gw));
    const float ry = scale_y * ((float)oy + 0.5f) - 0.5f;
    const float rx = scale_x * ((float)ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float wy[4], wx[4];
    cubic_w(ry - (float)iy, wy);
    cubic_w(rx - (float)ix, wx);
    const float d = dout[i];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), M - 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), M - 1);
            atomicAdd(dpos + ((int64_t)yy * M + xx) * E + e, wy[a] * wx[b] * d);
        }
    }
}

inline unsigned grid_for(int64_t n, int block = 256, int cap = 16384) {
    const int64_t gneed = (n + block - 1) / block;
    return (unsigned)(gneed < 1 ? 1 : (gneed > cap ? cap : gneed));
}

}  // namespace

int launch_softmax_rows(float* S, const uint8_t* mask, int64_t rows, int L, int rows_per_b, hipStream_t s) {
    softmax_rows_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(S, mask, rows, L, rows_per_b);
    return mst_check_launch("softmax_rows");
}

int launch_softmax_rows_bwd(const float* P, float* dP, int64_t rows, int L, float scale, hipStream_t s) {
    softmax_rows_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(P, dP, rows, L, scale);
    return mst_check_launch("softmax_rows_bwd");
}

int launch_layernorm_bwd(const float* x, int64_t xs, const float* gamma, const float* dy, int64_t dys, const float* dres, int64_t drs,
                         float* dx, int64_t dxs, float* dgamma, float* dbeta, int64_t rows, int cols, float eps, hipStream_t s) {
    MST_CHECK_ARG(cols > 0 && cols <= 2048, "layernorm_bwd: cols=%d unsupported (<= 2048)", cols);
    // four rows per wave at least, 1,024 workgroups at most: the d gamma / d beta atomics are per workgroup
    const unsigned grid = (unsigned)(rows < 16 ? 1 : (rows / 16 < 1024 ? rows / 16 : 1024));
#define LNB(CI) layernorm_bwd_kernel<CI><<<dim3(grid), dim3(256), 0, s>>>(x, xs, gamma, dy, dys, dres, drs, dx, dxs, dgamma, dbeta, rows, cols, eps)
    if (cols <= 128) LNB(2);
    else if (cols <= 384) LNB(6);
    else if (cols <= 768) LNB(12);
    else if (cols <= 1024) LNB(16);
    else LNB(32);
#undef LNB
    return mst_check_launch("layernorm_bwd");
}

int launch_rope_rows(float* qkv, int64_t rows, int L, int heads, int hd, const float* freqs, float sign, hipStream_t s) {
    MST_CHECK_ARG(qkv && freqs && rows > 0 && L > 0 && heads > 0 && hd > 0 && hd % 2 == 0, "rope_rows: bad arguments");
    rope_rows_kernel<<<dim3(grid_for(rows * heads * hd)), dim3(256), 0, s>>>(qkv, rows, L, heads, hd, freqs, sign);
    return mst_check_launch("rope_rows");
}

int launch_act_bwd(const float* h, float* dy, int64_t n, int kind, hipStream_t s) {
    act_bwd_kernel<<<dim3(grid_for(n)), dim3(256), 0, s>>>(h, dy, n, kind);
    return mst_check_launch("act_bwd");
}

int launch_act_fwd(const float* h, float* y, int64_t n, int kind, hipStream_t s) {
    act_fwd_kernel<<<dim3(grid_for(n)), dim3(256), 0, s>>>(h, y, n, kind);
    return mst_check_launch("act_fwd");
}

int launch_colsum(const float* a, int64_t as, const float* b, int64_t bs, int64_t rows, int cols, float* out, hipStream_t s) {
    MST_CHECK_ARG(rows > 0 && cols > 0, "colsum: rows=%lld cols=%d out of range", (long long)rows, cols);
    const int cblocks = (cols + 63) / 64;
    MST_CHECK_ARG(cblocks <= 65535, "colsum: cols=%d out of range", cols);
    // enough blocks to fill the chip (about 2,048), chunks of at least 64 rows so that the atomics stay few
    int64_t rpb = (rows * cblocks + 2047) / 2048;
    rpb = rpb < 64 ? 64 : (rpb + 15) / 16 * 16;
    const int64_t rblocks = (rows + rpb - 1) / rpb;
    MST_CHECK_ARG(rblocks < (1ll << 31) && rpb < (1ll << 31), "colsum: rows=%lld out of range", (long long)rows);
    const bool vec = cols % 4 == 0 && as % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
                     (!b || (bs % 4 == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0));
    const dim3 grid((unsigned)rblocks, cblocks);
    if (vec) colsum_kernel<true><<<grid, dim3(256), 0, s>>>(a, as, b, bs, rows, cols, (int)rpb, out);
    else colsum_kernel<false><<<grid, dim3(256), 0, s>>>(a, as, b, bs, rows, cols, (int)rpb, out);
    return mst_check_launch("colsum");
}

int launch_axpby_cols(const float* x, int64_t xs, const float* g, float alpha, float beta, float* y, int64_t ys, int64_t rows,
                      int cols, hipStream_t s) {
    axpby_cols_kernel<<<dim3(grid_for(rows * cols)), dim3(256), 0, s>>>(x, xs, g, alpha, beta, y, ys, rows, cols);
    return mst_check_launch("axpby_cols");
}

int launch_im2col14(const void* vol, int dt, int n, int H, int W, float* col, hipStream_t s) {
    const int gw = W / 14, Np = (H / 14) * gw;
    const int64_t total = (int64_t)n * Np * 196;
    if (dt == MST_F32) im2col14_kernel<float><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const float*)vol, H, W, gw, Np, total, col);
    else if (dt == MST_BF16) im2col14_kernel<bf16_t><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const bf16_t*)vol, H, W, gw, Np, total, col);
    else if (dt == MST_F16) im2col14_kernel<f16_t><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const f16_t*)vol, H, W, gw, Np, total, col);
    else { mst_set_error("im2col14: bad dtype %d", dt); return MST_EINVAL; }
    return mst_check_launch("im2col14");
}

int launch_pos_interp_bwd(const float* dout, int M, int E, int gh, int gw, double offset, float* dpos, hipStream_t s) {
    const float sy = (float)(1.0 / (((double)gh + offset) / (double)M));
    const float sx = (float)(1.0 / (((double)gw + offset) / (double)M));
    const int64_t tot = (int64_t)gh * gw * E;
    pos_interp_bwd_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(dout, M, E, gh, gw, sy, sx, dpos);
    return mst_check_launch("pos_interp_bwd");
}
