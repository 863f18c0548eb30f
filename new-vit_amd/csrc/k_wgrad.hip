// d weight of a convolution as an IMPLICIT GEMM (BASELINE configs[3]: what torch.autograd's conv backward gives the
// reference's ResNet training step, mst/models/resnet.py:172-193 + base_model.py:148-181):
//   dW[co][(ky,kx,c)] = sum over output pixels r of dz[r][co] * x[pixel(r) + (ky,kx)][c]
// The round-2 form wrote the im2col matrix [rows, kh*kw*Cin] (9x the activation for a 3 x 3 layer; 37 % of the configs[3]-shape step by
// itself) and multiplied it with the strided GEMM.  Here the B operand is gathered from the NHWC activation: a 64-column tile of the
// output lies inside ONE filter tap (Cin % 64 == 0), so the tap is a scalar per workgroup and a thread's float4 is four channels of the
// input pixel that tap pairs with its output pixel (zero outside the image).  Both operands are contiguous along the GEMM's M / N index
// and strided along K (the pixel index): they enter LDS as [k][m] / [k][n] rows, the layout the fp32 MFMA (32x32x2) reads directly.
// The pixels are split over blockIdx.z into partial products (the caller sums them with mst_colsum), as the explicit form did.
// Structure and pipeline: k_gemm_ex.hip (64 x 64 tile, K-step 16, register prefetch, LDS double buffer, one barrier per step).
#include "mst_common.h"

namespace {

constexpr int LDT = 68;

struct WgradArgs {
    const float* dz; const float* x; float* part;
    int Cout, Kc, H, W, Cin, kw, stride, pad, Ho, Wo;
    int64_t rows, rows_per_split;
};

__global__ __launch_bounds__(256) void wgrad32_kernel(WgradArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][16][LDT], Bs[2][16][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;                 // output channels, (tap, c) columns
    const int tap = n0 / g.Cin, c0 = n0 - tap * g.Cin;                    // the whole tile lies in one filter tap
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int64_t r_begin = (int64_t)blockIdx.z * g.rows_per_split;
    const int64_t r_end = r_begin + g.rows_per_split < g.rows ? r_begin + g.rows_per_split : g.rows;
    const int kq = tid >> 4, q4 = (tid & 15) * 4;                         // this thread's pixel within a K-step, its four columns
    const int hw = g.Ho * g.Wo;
    float4 ra, rb;
    auto gload = [&](int64_t r0) {
        const int64_t r = r0 + kq;
        const bool in = r < r_end;
        ra = (in && m0 + q4 < g.Cout) ? *reinterpret_cast<const float4*>(g.dz + r * g.Cout + m0 + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t rr = in ? r : 0;
        const int img = (int)(rr / hw), rem = (int)(rr - (int64_t)img * hw);
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
        const bool ok = in && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        rb = ok ? *reinterpret_cast<const float4*>(g.x + (((int64_t)img * g.H + iy) * g.W + ix) * g.Cin + c0 + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][kq][q4]) = ra;
        *reinterpret_cast<float4*>(&Bs[buf][kq][q4]) = rb;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    gload(r_begin);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 16) {
        const bool more = r0 + 16 < r_end;
        if (more) gload(r0 + 16);                                         // in flight across the MFMAs below
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][k][wr * 32 + (lane & 31)], Bs[cur][k][wc * 32 + (lane & 31)], acc, 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    float* C = g.part + (int64_t)blockIdx.z * g.Cout * g.Kc;
    const int col = n0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.Cout) C[(int64_t)row * g.Kc + col] = acc[r];
    }
}

}  // namespace

// part[z][co][(ky,kx,c)] = sum over the output pixels [z * rows_per_split, (z + 1) * rows_per_split) of dz[r][co] * x[...]: nsplit partial
// products of the weight gradient, fp32 [nsplit, Cout, kh*kw*Cin].  dz [n*Ho*Wo, Cout], x [n,H,W,Cin] fp32; Cin % 64 == 0, Cout % 4 == 0.
int launch_conv_wgrad32(const float* dz, const float* x, int n, int H, int W_, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                        int nsplit, int64_t rows_per_split, hipStream_t s) {
    MST_CHECK_ARG(dz && x && part && n > 0 && H > 0 && W_ > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "conv_wgrad: bad arguments");
    MST_CHECK_ARG(Cin % 64 == 0, "conv_wgrad: Cin=%d must be a multiple of 64 (use mst_im2col_nhwc + mst_gemm_ex otherwise)", Cin);
    MST_CHECK_ARG(Cout > 0 && Cout % 4 == 0, "conv_wgrad: Cout=%d must be a multiple of 4", Cout);
    MST_CHECK_ARG(((uintptr_t)dz & 15) == 0 && ((uintptr_t)x & 15) == 0, "conv_wgrad: bases must be 16-byte aligned");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W_ + 2 * pad - kw) / stride + 1;
    const int64_t rows = (int64_t)n * Ho * Wo;
    MST_CHECK_ARG(Ho > 0 && Wo > 0 && nsplit > 0 && nsplit <= 65535 && rows_per_split > 0 && (int64_t)nsplit * rows_per_split >= rows,
                  "conv_wgrad: %d splits of %lld rows do not cover %lld", nsplit, (long long)rows_per_split, (long long)rows);
    WgradArgs g{dz, x, part, Cout, kh * kw * Cin, H, W_, Cin, kw, stride, pad, Ho, Wo, rows, rows_per_split};
    wgrad32_kernel<<<dim3(g.Kc / 64, (Cout + 63) / 64, nsplit), dim3(256), 0, s>>>(g);
    return mst_check_launch("conv_wgrad");
}
