// Implicit-GEMM convolution on 16-bit MFMA operands (bf16 / fp16 activations NHWC, fp32 accumulate): the 16-bit inference path of the
// ResNet backbone (reference mst/models/resnet.py:44-50,172-193 over torchvision's resnet: F.conv2d + folded BatchNorm +
// ReLU / residual).  The structure is k_gemm16.hip's (128 x BN x 64 tile, 4 waves 2 x 2, LDS-DMA double buffer, swizzle on the per-lane
// SOURCE address), with the A operand GATHERED: row m = output pixel (img, oy, ox), a 64-wide K-step = 64 consecutive input channels of
// ONE filter tap (Cin % 64 == 0), so every 1 KiB LDS-DMA piece is eight 128-byte runs of eight input pixels -- or of a zero page for
// taps outside the image and rows beyond M (LDS-DMA has no predicated zero fill; the page is 128 bytes of zeros in device memory).
// No [rows, kh*kw*Cin] matrix exists.  BN = 64 serves the 64-channel stage (and the stem's im2col product) without wasted MFMAs.
//   epilogues: bias | bias + ReLU (16-bit or fp32 out) | relu(C + acc + bias) in place on a 16-bit C (the unit's exit).
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int BM = 128, BK = 64;
__device__ __attribute__((aligned(128))) char conv16_zero_page[128];      // zero-initialised device memory

struct Conv16Geom { int H, W, C, kh, kw, stride, pad, Ho, Wo, dshift; };   // dshift: log2 of the input dilation (d input of a stride-2 layer: 1)

template <typename T, int BN, int EPI, typename OutT>
__global__ __launch_bounds__(256) void conv16_kernel(const T* __restrict__ x, Conv16Geom g, const T* __restrict__ Wg, int64_t ldw,
                                                     const float* __restrict__ bias, OutT* C, int64_t ldc, int M, int N, int tiles_n, int nwg) {
    typedef typename V8<T>::type vec8;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;       // per stage
    constexpr int NI = BN / 32;                                       // 16-column accumulator blocks per wave (wave tile 64 x BN/2)
    constexpr int PW = BN / 32;                                       // W pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                                            // [2][128][64] T
    char* const Ws = smem + 2 * A_BYTES;                              // [2][BN][64] T

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- A pieces: instruction i of this wave fills tile rows (wave*4+i)*8 .. +7; lane -> row r, 16-byte chunk c of the 128-byte run
    int py[4], px[4];                                                 // top-left input pixel of the row's window (may be negative)
    int64_t pbase[4];                                                 // element offset of image `img` + this lane's chunk
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);                    // chunk whose swizzled home is LDS slot lane&7
        const int m = m0 + r;
        const int mm = m < M ? m : 0;
        const int img = mm / hw, rem = mm - img * hw;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        py[i] = m < M ? oy * g.stride - g.pad : -(1 << 20);           // rows beyond M: every tap out of range -> zero page
        px[i] = ox * g.stride - g.pad;
        pbase[i] = (int64_t)img * g.H * g.W * g.C + c * 8;
    }
    const T* wsrc[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int r = (wave * PW + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int n = n0 + r < N ? n0 + r : N - 1;
        wsrc[i] = Wg + (int64_t)n * ldw + c * 8;
    }
    const char* const zsrc = conv16_zero_page + ((lane & 7) << 4);
    const int cpk = g.C / BK;                                         // K-steps per filter tap
    auto stage = [&](int t, int buf) {
        const int tap = t / cpk, cc = (t - tap * cpk) * BK;           // scalar
        const int ky = tap / g.kw, kx = tap - ky * g.kw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ny = py[i] + ky, nx = px[i] + kx, dmask = (1 << g.dshift) - 1;
            const int iy = ny >> g.dshift, ix = nx >> g.dshift;
            const bool ok = ny >= 0 && nx >= 0 && ((ny | nx) & dmask) == 0 && iy < g.H && ix < g.W;
            const char* src = ok ? reinterpret_cast<const char*>(x + pbase[i] + ((int64_t)iy * g.W + ix) * g.C + cc) : zsrc;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(As + buf * A_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i]), LDS_PTR(Ws + buf * W_BYTES + (wave * PW + i) * 1024), 16, 0, 0);
            wsrc[i] += BK;
        }
    };

    // accumulators start at the bias
    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
        if (bias && n < N) b0 = *reinterpret_cast<const f32x4*>(bias + n);                 // N % 4 == 0
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = b0;
    }

    const int sw = (lane >> 1) & 7;                                   // ((row >> 1) & 7) for row = 16*k + (lane & 15)
    const int a_row_off = (wm * 64 + (lane & 15)) * 128;
    const int w_row_off = (wn * (BN / 2) + (lane & 15)) * 128;

    const int nk = g.kh * g.kw * cpk;
    stage(0, 0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // tile t has landed; every wave is done reading the other buffer
        if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
        const char* Ab = As + (t & 1) * A_BYTES;
        const char* Wb = Ws + (t & 1) * W_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
            vec8 af[4], wf[NI];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const vec8*>(Ab + a_row_off + j * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const vec8*>(Wb + w_row_off + i * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], af[j], acc[i][j]);
        }
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = m0+wm*64+j*16+(lane&15), n = n0+wn*(BN/2)+i*16+(lane>>4)*4
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + (lane & 15);
            if (m >= M) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            OutT* cp = C + (int64_t)m * ldc + n;
            if constexpr (EPI == MST_EPI_RESIDUAL_RELU) {
                typedef __attribute__((ext_vector_type(4))) OutT o4;
                const o4 idv = *reinterpret_cast<const o4*>(cp);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + (float)idv[r], 0.f);
            } else if constexpr (EPI == MST_EPI_BIAS_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (sizeof(OutT) == 4) {
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                typedef __attribute__((ext_vector_type(4))) OutT o4;
                o4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (OutT)v[r];
                *reinterpret_cast<o4*>(cp) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// d weight on 16-bit operands: dW[co][(ky,kx,c)] = sum over output pixels r of dz[r][co] * x[pixel(r) + (ky,kx)][c] (see k_wgrad.hip for the fp32
// form).  Both operands are stored pixel-major ([r][co], [pixel][c]) but the MFMA wants 8 consecutive K values (pixels) per lane: the tiles
// land in LDS as they are (64 pixels x 64 columns images, LDS-DMA, the x rows gathered per tap with the zero page for padding) and the
// fragments are read with the hardware transpose (ds_read_b64_tr_b16), the way the attention kernel reads V^T (k_attn16.hip): the same
// lane map on both operands permutes K identically, so the sum is unchanged.  128 x 128 (or 64 x 128) output tile per workgroup, 4 waves x
// 64 x 64 accumulators of v_mfma_f32_32x32x16, K-step 64 pixels, three-stage LDS ring; pixels split over blockIdx.z into fp32 partial products.
struct Wgrad16Args {
    int Cout, Kc, H, W, Cin, kw, stride, pad, Ho, Wo;
    int64_t rows, rows_per_split;
};

template <typename T>
__device__ __forceinline__ typename V8<T>::type tr_frag(const char* p) {       // p, p + 8 rows: elements j = 0..3 | 4..7 of the fragment
    typedef typename V8<T>::type vec8;
    union { struct { s16x4 lo, hi; } s; vec8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * 128));
    return u.v;
}

// MI = 2: 128 x 128 output tile (two dz images: wave (wm, wn) multiplies dz image wm with x image wn over the whole stage).
// MI = 1: 64 x 128 tile for Cout <= 64 (the 64-channel stage, where the second dz image would be padding): one dz image, the waves wm = 0 / 1
//         take the first / second 32 pixels of each stage and their accumulators are added through LDS at the end.
// NS = 2: double buffer (64 / 48 KiB: two or three workgroups per CU cover each other's LDS-DMA latency).  NS = 3 behind a counted vmcnt was
// measured SLOWER on the 128-row tile (0.094 -> 0.117 ms on the DINOv2 fc1 shape: one workgroup per CU at 96 KiB; the kernel is bound by
// its LDS reads and address arithmetic, not by the latency), so 2 is what runs.
template <typename T, int MI, int NS = 2>
__global__ __launch_bounds__(256) void wgrad16_kernel(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ part, Wgrad16Args g) {
    typedef typename V8<T>::type vec8;
    constexpr int IMG = 64 * 128;                                     // one image: 64 pixels x 64 columns x 2 B
    constexpr int NIMG = MI + 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [NS][dz images 0 .. MI-1, x image 0, x image 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * (64 * MI), n0 = blockIdx.x * 128;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t r_begin = (int64_t)blockIdx.z * g.rows_per_split;
    const int64_t r_end = r_begin + g.rows_per_split < g.rows ? r_begin + g.rows_per_split : g.rows;

    // ---- staging: one image per loader wave and stage (MI = 2: waves 0, 1 the dz images, 2, 3 the x images; MI = 1: wave 0 dz, 1, 2 x, wave
    // 3 loads nothing), eight 1 KiB pieces of 8 pixels; lane -> pixel row (lane >> 3) of the piece, LDS slot lane & 7, source chunk
    // permuted like the V image of k_attn16.hip (the permutation depends on bit 1 of the row only: constant per lane)
    const int my_img = wave;                                          // image index inside a stage
    const bool loader = my_img < NIMG;
    const bool is_b = my_img >= MI;
    const int sub = is_b ? my_img - MI : my_img;
    const int col0 = is_b ? n0 + 64 * sub : m0 + 64 * sub;            // first column of this image
    const bool img_ok = loader && (is_b ? col0 < g.Kc : col0 < g.Cout);
    const int tap = is_b ? col0 / g.Cin : 0, c0 = is_b ? col0 - tap * g.Cin : col0;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ (((prow >> 1) & 1) << 2);
    int64_t r_lane = r_begin + prow;
    const T* a_ptr = dz + r_lane * g.Cout + c0 + chunk * 8;           // dz loader: + 8 rows per piece
    int p_img, p_oy, p_ox;                                            // x loader: (image, oy, ox) of pixel r_lane, advanced by 8 pixels per piece
    {
        const int hw = g.Ho * g.Wo;
        p_img = (int)(r_lane / hw);
        const int rem = (int)(r_lane - (int64_t)p_img * hw);
        p_oy = rem / g.Wo;
        p_ox = rem - p_oy * g.Wo;
    }
    const int64_t img_elems = (int64_t)g.H * g.W * g.Cin;             // < 2^31 (launcher): per-image offsets fit 32 bits
    const T* x_img = x + p_img * img_elems + c0 + chunk * 8;
    const bool plain = g.Ho * g.Wo == 1;                              // nn.Linear's d weight: pixel = row, no window
    const char* const zsrc = conv16_zero_page + ((lane & 7) << 4);
    auto stage = [&](int st) {
        if (!loader) return;
        char* dst = smem + (st * NIMG + my_img) * IMG;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const char* src = zsrc;
            if (img_ok && r_lane < r_end) {
                if (!is_b) {
                    src = reinterpret_cast<const char*>(a_ptr);
                } else {
                    const int iy = p_oy * g.stride - g.pad + ky, ix = p_ox * g.stride - g.pad + kx;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) src = reinterpret_cast<const char*>(x_img + (iy * g.W + ix) * g.Cin);
                }
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst + p8 * 1024), 16, 0, 0);
            r_lane += 8;
            a_ptr += 8 * (int64_t)g.Cout;
            if (plain) {
                x_img += 8 * img_elems;
            } else {
                p_ox += 8;
                while (p_ox >= g.Wo) { p_ox -= g.Wo; ++p_oy; }
                while (p_oy >= g.Ho) { p_oy -= g.Ho; x_img += img_elems; }
            }
        }
    };

    // ---- fragment addresses (k_attn16.hip's V map): 16-lane group reads 4 pixels x 16 columns, transposed by the hardware
    const int h2 = lane >> 5, vi = lane & 15;
    const int vrow = 4 * h2 + (vi >> 2);
    const int vsw = ((vrow >> 1) & 1) << 2;
    const int vchunk = ((lane >> 4) & 1) * 2 + ((vi & 3) >> 1);
    const int v_lane_off = vrow * 128 + (vi & 1) * 8;
    int foff[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) foff[db] = v_lane_off + (((db * 4 + vchunk) ^ vsw) * 16);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int64_t nst = (r_end - r_begin + 63) / 64;
    stage(0);
    if (NS == 3 && nst > 1) stage(1);
    for (int64_t t = 0; t < nst; ++t) {
        // NS = 3: stage t has landed once only the stage behind it is outstanding (8 pieces of this wave; LDS-DMA completes in issue order)
        if (NS == 3 && t + 1 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // stage t is visible; every wave is done reading the buffer the next issue overwrites
        if (t + NS - 1 < nst) stage((int)((t + NS - 1) % NS));
        const int st = (int)(t % NS);
        const char* Ai = smem + (st * NIMG + (MI == 2 ? wm : 0)) * IMG;
        const char* Bi = smem + (st * NIMG + MI + wn) * IMG;
#pragma unroll
        for (int kq = 0; kq < (MI == 2 ? 4 : 2); ++kq) {
            const int ks = MI == 2 ? kq : 2 * wm + kq;                // MI = 1: this wave's half of the stage's pixels
            vec8 af[2], bf[2];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                af[db] = tr_frag<T>(Ai + ks * 16 * 128 + foff[db]);
                bf[db] = tr_frag<T>(Bi + ks * 16 * 128 + foff[db]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (an empty split issued its first stage and never waited for it)
    if (MI == 1) {                                                    // the two pixel halves meet: waves 2, 3 hand their sums to waves 0, 1
        __syncthreads();                                              // the ring is free
        float* ex = reinterpret_cast<float*>(smem) + (wn * 64 + lane) * 65;      // 64 floats per lane (+1: bank spread), per x image
        if (wm == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[(i * 2 + j) * 16 + r] = acc[i][j][r];
        }
        __syncthreads();
        if (wm == 1) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[(i * 2 + j) * 16 + r];
    }
    float* C = part + (int64_t)blockIdx.z * g.Cout * g.Kc;
    const int mrow0 = m0 + (MI == 2 ? wm * 64 : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.Kc) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.Cout) C[(int64_t)row * g.Kc + col] = acc[i][j][r];
            }
        }
}

template <typename T, int BN, int EPI, typename OutT>
int launch_k(const void* x, const Conv16Geom& g, int n, const void* Wg, int64_t ldw, const float* bias, void* C, int64_t ldc, int N,
             hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = conv16_kernel<T, BN, EPI, OutT>;
    constexpr int lds = 2 * BM * BK * 2 + 2 * BN * BK * 2;
    mst_allow_lds((const void*)kern, lds, &lds_once);
    const int64_t M = (int64_t)n * g.Ho * g.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    kern<<<dim3(nwg), dim3(256), lds, s>>>((const T*)x, g, (const T*)Wg, ldw, bias, (OutT*)C, ldc, (int)M, N, tiles_n, nwg);
    return mst_check_launch("conv16");
}

template <typename T, int BN>
int dispatch_epi(const void* x, const Conv16Geom& g, int n, const void* Wg, int64_t ldw, const float* bias, void* C, int cdt, int64_t ldc,
                 int N, int epi, hipStream_t s) {
    const bool f32out = cdt == MST_F32;
    switch (epi) {
        case MST_EPI_BIAS:
            return f32out ? launch_k<T, BN, MST_EPI_BIAS, float>(x, g, n, Wg, ldw, bias, C, ldc, N, s)
                          : launch_k<T, BN, MST_EPI_BIAS, T>(x, g, n, Wg, ldw, bias, C, ldc, N, s);
        case MST_EPI_BIAS_RELU:
            return f32out ? launch_k<T, BN, MST_EPI_BIAS_RELU, float>(x, g, n, Wg, ldw, bias, C, ldc, N, s)
                          : launch_k<T, BN, MST_EPI_BIAS_RELU, T>(x, g, n, Wg, ldw, bias, C, ldc, N, s);
        case MST_EPI_RESIDUAL_RELU:
            if (!f32out) return launch_k<T, BN, MST_EPI_RESIDUAL_RELU, T>(x, g, n, Wg, ldw, bias, C, ldc, N, s);
    }
    mst_set_error("conv_gemm16: epilogue %d with output dtype %d unsupported (bias, bias + ReLU: 16-bit or f32 out; residual + ReLU: 16-bit in place)", epi, cdt);
    return MST_EINVAL;
}

}  // namespace

// out[(img, oy, ox)][co] = epi(sum_{ky,kx,c} x[img][oy*stride-pad+ky][ox*stride-pad+kx][c] * Wg[co][(ky,kx,c)] + bias[co]) on 16-bit MFMA
// operands: x [n,H,W,Cin] and Wg [Cout, kh*kw*Cin] of type dt (Cin % 64 == 0, Cout % 4 == 0), out [n*Ho*Wo, Cout] of type cdt.
int launch_conv_gemm16(const void* x, int dt, int n, int H, int W_, int Cin, int kh, int kw, int stride, int pad, const void* Wg, const float* bias,
                       void* out, int cdt, int Cout, int epi, hipStream_t s) {
    MST_CHECK_ARG(x && Wg && out && n > 0 && H > 0 && W_ > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "conv_gemm16: bad arguments");
    MST_CHECK_ARG(dt == MST_BF16 || dt == MST_F16, "conv_gemm16: operand dtype %d (bf16 / f16)", dt);
    MST_CHECK_ARG(cdt == MST_F32 || cdt == dt, "conv_gemm16: output dtype must be f32 or the operand dtype");
    MST_CHECK_ARG(Cin % BK == 0, "conv_gemm16: Cin=%d must be a multiple of %d", Cin, BK);
    MST_CHECK_ARG(Cout > 0 && Cout % 4 == 0, "conv_gemm16: Cout=%d must be a multiple of 4", Cout);
    MST_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)Wg & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                  "conv_gemm16: bases (x, Wg, out, bias) must be 16-byte aligned");
    Conv16Geom g{H, W_, Cin, kh, kw, stride, pad, (H + 2 * pad - kh) / stride + 1, (W_ + 2 * pad - kw) / stride + 1, 0};
    MST_CHECK_ARG(g.Ho > 0 && g.Wo > 0 && (int64_t)n * g.Ho * g.Wo < (1ll << 31) - BM, "conv_gemm16: output %d x %d x %d", n, g.Ho, g.Wo);
    const int64_t ldw = (int64_t)kh * kw * Cin;
    const bool narrow = Cout <= 64;
    if (dt == MST_BF16)
        return narrow ? dispatch_epi<bf16_t, 64>(x, g, n, Wg, ldw, bias, out, cdt, Cout, Cout, epi, s)
                      : dispatch_epi<bf16_t, 128>(x, g, n, Wg, ldw, bias, out, cdt, Cout, Cout, epi, s);
    return narrow ? dispatch_epi<f16_t, 64>(x, g, n, Wg, ldw, bias, out, cdt, Cout, Cout, epi, s)
                  : dispatch_epi<f16_t, 128>(x, g, n, Wg, ldw, bias, out, cdt, Cout, Cout, epi, s);
}

// d input of a convolution on 16-bit operands (see launch_conv_dgrad32): dz [n,Ho,Wo,Cout] and Wt [Cin, kh*kw*Cout] of type dt (Cout % 64 == 0),
// dx [n*H*W, Cin] fp32, overwritten.
int launch_conv_dgrad16(const void* dz, int dt, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const void* Wt, int H, int W_,
                        int Cin, float* dx, hipStream_t s) {
    MST_CHECK_ARG(dz && Wt && dx && n > 0 && Ho > 0 && Wo > 0 && H > 0 && W_ > 0 && kh == kw && kh > 0 && pad >= 0 && pad < kh, "conv_dgrad16: bad arguments");
    MST_CHECK_ARG(dt == MST_BF16 || dt == MST_F16, "conv_dgrad16: operand dtype %d (bf16 / f16)", dt);
    MST_CHECK_ARG(stride == 1 || stride == 2, "conv_dgrad16: stride %d (1 or 2)", stride);
    MST_CHECK_ARG(Cout % BK == 0 && Cin % 4 == 0, "conv_dgrad16: Cout=%d must be a multiple of %d, Cin=%d of 4", Cout, BK, Cin);
    MST_CHECK_ARG((H + 2 * pad - kh) / stride + 1 == Ho && (W_ + 2 * pad - kw) / stride + 1 == Wo, "conv_dgrad16: %d x %d is not the output of a %d x %d input", Ho, Wo, H, W_);
    MST_CHECK_ARG((int64_t)n * H * W_ < (1ll << 31) - BM, "conv_dgrad16: %d x %d x %d", n, H, W_);
    MST_CHECK_ARG(((uintptr_t)dz & 15) == 0 && ((uintptr_t)Wt & 15) == 0 && ((uintptr_t)dx & 15) == 0, "conv_dgrad16: bases must be 16-byte aligned");
    Conv16Geom g{Ho, Wo, Cout, kh, kw, 1, kh - 1 - pad, H, W_, stride - 1};
    const int64_t ldw = (int64_t)kh * kw * Cout;
    const bool narrow = Cin <= 64;
    if (dt == MST_BF16)
        return narrow ? launch_k<bf16_t, 64, MST_EPI_BIAS, float>(dz, g, n, Wt, ldw, nullptr, dx, Cin, Cin, s)
                      : launch_k<bf16_t, 128, MST_EPI_BIAS, float>(dz, g, n, Wt, ldw, nullptr, dx, Cin, Cin, s);
    return narrow ? launch_k<f16_t, 64, MST_EPI_BIAS, float>(dz, g, n, Wt, ldw, nullptr, dx, Cin, Cin, s)
                  : launch_k<f16_t, 128, MST_EPI_BIAS, float>(dz, g, n, Wt, ldw, nullptr, dx, Cin, Cin, s);
}

// part[z][co][(ky,kx,c)] on 16-bit operands: dz [n*Ho*Wo, Cout] and x [n,H,W,Cin] of type dt (Cin % 64 == 0, Cout % 64 == 0); see launch_conv_wgrad32.
int launch_conv_wgrad16(const void* dz, const void* x, int dt, int n, int H, int W_, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                        int nsplit, int64_t rows_per_split, hipStream_t s) {
    MST_CHECK_ARG(dz && x && part && n > 0 && H > 0 && W_ > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "conv_wgrad16: bad arguments");
    MST_CHECK_ARG(dt == MST_BF16 || dt == MST_F16, "conv_wgrad16: operand dtype %d (bf16 / f16)", dt);
    MST_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0, "conv_wgrad16: Cin=%d and Cout=%d must be multiples of 64", Cin, Cout);
    MST_CHECK_ARG(((uintptr_t)dz & 15) == 0 && ((uintptr_t)x & 15) == 0, "conv_wgrad16: bases must be 16-byte aligned");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W_ + 2 * pad - kw) / stride + 1;
    const int64_t rows = (int64_t)n * Ho * Wo;
    MST_CHECK_ARG(Ho > 0 && Wo > 0 && nsplit > 0 && nsplit <= 65535 && rows_per_split > 0 && rows_per_split % 64 == 0 && (int64_t)nsplit * rows_per_split >= rows,
                  "conv_wgrad16: %d splits of %lld rows (a multiple of 64) do not cover %lld", nsplit, (long long)rows_per_split, (long long)rows);
    MST_CHECK_ARG((int64_t)H * W_ * Cin < (1ll << 31), "conv_wgrad16: one image of %d x %d x %d elements exceeds 32-bit offsets", H, W_, Cin);
    Wgrad16Args g{Cout, kh * kw * Cin, H, W_, Cin, kw, stride, pad, Ho, Wo, rows, rows_per_split};
    constexpr int IMG = 64 * 128;
#define WG16(T, MI)                                                                                                        \
    {                                                                                                                      \
        static mst_lds_once once;                                                                                          \
        auto kern = wgrad16_kernel<T, MI>;                                                                                 \
        constexpr int lds = 2 * (MI + 2) * IMG;                                                                            \
        mst_allow_lds((const void*)kern, lds, &once);                                                                      \
        kern<<<dim3((g.Kc + 127) / 128, (Cout + 64 * MI - 1) / (64 * MI), nsplit), dim3(256), lds, s>>>((const T*)dz, (const T*)x, part, g); \
    }
    if (Cout <= 64) {
        if (dt == MST_BF16) WG16(bf16_t, 1) else WG16(f16_t, 1)
    } else {
        if (dt == MST_BF16) WG16(bf16_t, 2) else WG16(f16_t, 2)
    }
#undef WG16
    return mst_check_launch("conv_wgrad16");
}
