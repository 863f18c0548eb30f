// Patch embedding fused with tokenisation, straight from the [slices, H, W] volume:
//   x[s, 1+R+p, :] = conv14x14(gray->RGB slice s)[p] + bias + pos_patch[p]   (patch rows)
//   x[s, r, :]     = prefix[r]   (r = 0: cls_token + pos[0]; r = 1..R: register tokens)
// The reference materialises a 3x RGB copy (dino.py:125-127) and runs Conv2d(3,E,14,14)
// (patch_embed.py:65,75-77); the three channels are identical, so the kernel is pre-summed over
// C_in on the host (K 588 -> 196) and the volume is read exactly once, coalesced along W.
// Each workgroup builds a 128-patch x (14 rows x 16 cols, cols 14,15 zero) im2col tile in LDS
// (converted to the MFMA operand type), a 128-channel weight tile next to it, and runs
// 7 k-steps of v_mfma_f32_16x16x32 (fp32 mode: a k-loop of v_mfma_f32_16x16x4_f32).
// Also here: bicubic resampling of the position grid (vision_transformer.py:179-211).
#include "mst_common.h"

namespace {

constexpr int PATCH = 14, KP = 224;
// tile = (2*TW*16)^2: 128x128 for 16-bit operands, 64x64 in fp32 mode (LDS: 116 KB / 113 KB)
#ifndef PATCH_TILEW
#define PATCH_TILEW 2      // 64-patch tiles: 59 KB of LDS, two workgroups per CU (0.42 ms vs 0.455 ms with 128-patch tiles)
#endif
template <typename T> struct TileW { static constexpr int v = PATCH_TILEW; };
template <> struct TileW<float> { static constexpr int v = 2; };

template <typename T> struct RowBytes { static constexpr int v = 464; };     // 232 x 2 B (29 slots: odd)
template <> struct RowBytes<float> { static constexpr int v = 225 * 4; };    // 225 dwords (odd)

template <typename InT> struct In2;
template <> struct In2<float> { typedef float2 type; };
template <> struct In2<f16_t> { typedef __attribute__((ext_vector_type(2))) f16_t type; };
template <> struct In2<bf16_t> { typedef __attribute__((ext_vector_type(2))) bf16_t type; };

// T = MFMA operand type (bf16/f16/float), InT = volume dtype
template <typename T, typename InT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const InT* __restrict__ vol, int H, int W,
                                                          int gw, int Np, int64_t total,
                                                          const T* __restrict__ wp,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ pos_patch,
                                                          int n_prefix, int E, float* __restrict__ x) {
    constexpr int RB = RowBytes<T>::v;
    constexpr int TW = TileW<T>::v, BMP = 32 * TW, BNP = 32 * TW, WT = 16 * TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;             // [128][RB] patches
    char* const Ws = smem + BMP * RB;  // [128][RB] channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * TileW<T>::v);
    const int n0 = blockIdx.y * (32 * TileW<T>::v);
    const int N = n_prefix + Np;

    // ---- im2col: (patch, ky) -> 14 pixels + 2 zeros = one 16-element k group
    typedef typename In2<InT>::type in2;
    for (int idx = tid; idx < BMP * PATCH; idx += 256) {
        const int pl = idx % BMP, ky = idx / BMP;
        int64_t m = m0 + pl;
        if (m >= total) m = total - 1;
        const int s = (int)(m / Np), p = (int)(m % Np);
        const int py = p / gw, px = p % gw;
        const InT* src = vol + ((int64_t)s * H + py * PATCH + ky) * W + px * PATCH;
        T v[16];
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            const in2 t = *reinterpret_cast<const in2*>(src + 2 * e);
            if constexpr (sizeof(InT) == 4) {
                v[2 * e] = (T)t.x;
                v[2 * e + 1] = (T)t.y;
            } else {
                v[2 * e] = (T)(float)t[0];
                v[2 * e + 1] = (T)(float)t[1];
            }
        }
        v[14] = (T)0.f;
        v[15] = (T)0.f;
        T* dst = reinterpret_cast<T*>(As + pl * RB) + ky * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[e] = v[e];
    }
    // ---- weight tile: rows n0..n0+127 of wp[E][224]
    {
        constexpr int CH = KP * (int)sizeof(T) / 16;  // 16-byte chunks per row
        for (int idx = tid; idx < BNP * CH; idx += 256) {
            const int r = idx / CH, c = idx % CH;
            const u32x4 t = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wp + (int64_t)(n0 + r) * KP) + c * 16);
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<u32x4*>(Ws + r * RB + c * 16) = t;
            } else {
                unsigned* d = reinterpret_cast<unsigned*>(Ws + r * RB + c * 16);
                d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
            }
        }
    }
    __syncthreads();

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (sizeof(T) == 2) {
        typedef typename V8<T>::type vec8;
#pragma unroll
        for (int kk = 0; kk < KP / 32; ++kk) {
            const int coff = (kk * 4 + (lane >> 4)) * 16;
            vec8 af[TW], wf[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j)
                af[j] = *reinterpret_cast<const vec8*>(As + (wm * WT + j * 16 + (lane & 15)) * RB + coff);
#pragma unroll
            for (int i = 0; i < TW; ++i)
                wf[i] = *reinterpret_cast<const vec8*>(Ws + (wn * WT + i * 16 + (lane & 15)) * RB + coff);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = mfma16(wf[i], af[j], acc[i][j]);
        }
    } else {
        // fp32: v_mfma_f32_16x16x4_f32, lane holds A[lane&15][k = lane>>4]
        for (int k4 = 0; k4 < KP / 4; ++k4) {
            const int kof = (k4 * 4 + (lane >> 4)) * 4;
            float af[TW], wf[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j)
                af[j] = *reinterpret_cast<const float*>(As + (wm * WT + j * 16 + (lane & 15)) * RB + kof);
#pragma unroll
            for (int i = 0; i < TW; ++i)
                wf[i] = *reinterpret_cast<const float*>(Ws + (wn * WT + i * 16 + (lane & 15)) * RB + kof);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns channels n..n+3 of patch m
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int64_t m = m0 + wm * WT + j * 16 + (lane & 15);
        if (m >= total) continue;
        const int s = (int)(m / Np), p = (int)(m % Np);
        float* xr = x + ((int64_t)s * N + n_prefix + p) * E;
        const float* pr = pos_patch + (int64_t)p * E;
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const int n = n0 + wn * WT + i * 16 + (lane >> 4) * 4;
            const float4 bv = *reinterpret_cast<const float4*>(bias + n);
            const float4 pv = *reinterpret_cast<const float4*>(pr + n);
            float4 o;
            o.x = acc[i][j][0] + bv.x + pv.x;
            o.y = acc[i][j][1] + bv.y + pv.y;
            o.z = acc[i][j][2] + bv.z + pv.z;
            o.w = acc[i][j][3] + bv.w + pv.w;
            *reinterpret_cast<float4*>(xr + n) = o;
        }
    }
}

__global__ void prefix_rows_kernel(const float* __restrict__ prefix, int n_prefix, int E, int N, int n,
                                   float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = (int64_t)n * n_prefix * E;
    if (i >= tot) return;
    const int e = (int)(i % E);
    const int r = (int)((i / E) % n_prefix);
    const int64_t s = i / ((int64_t)E * n_prefix);
    x[(s * N + r) * E + e] = prefix[r * E + e];
}

// cubic convolution coefficients, A = -0.75 (what F.interpolate(mode='bicubic') uses)
__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
    const float A = -0.75f;
    float x = t + 1.0f;
    w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

// pos [M*M, E] -> out [gh*gw, E]; source index = scale*(dst+0.5)-0.5 with scale = 1/scale_factor,
// border taps clamped (align_corners=False, antialias off).
__global__ void pos_interp_kernel(const float* __restrict__ pos, int M, int E, int gh, int gw, float scale_y,
                                  float scale_x, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)gh * gw * E) return;
    const int e = (int)(i % E);
    const int ox = (int)((i / E) % gw);
    const int oy = (int)(i / ((int64_t)E * gw));
    const float ry = scale_y * ((float)oy + 0.5f) - 0.5f;
    const float rx = scale_x * ((float)ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float wy[4], wx[4];
    cubic_coeffs(ry - (float)iy, wy);
    cubic_coeffs(rx - (float)ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), M - 1);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), M - 1);
            row += wx[b] * pos[((int64_t)yy * M + xx) * E + e];
        }
        acc += wy[a] * row;
    }
    out[i] = acc;
}

// Anti-aliased bicubic (interpolate_antialias=True: the hub's register models), torch's separable "aa" resampling
// (aten UpSampleKernel _compute_indices_weights_aa): support = 2 * max(scale, 1), centre = scale * (dst + 0.5), taps
// [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the grid, weights = Keys cubic with A = -0.5
// (not the -0.75 of the plain bicubic) of (tap - centre + 0.5) / max(scale, 1), normalised to sum 1.
__device__ __forceinline__ float cubic_aa(float x) {
    const float A = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * A;
    return 0.0f;
}
__global__ void pos_interp_aa_kernel(const float* __restrict__ pos, int M, int E, int gh, int gw, float scale_y,
                                     float scale_x, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)gh * gw * E) return;
    const int e = (int)(i % E);
    const int ox = (int)((i / E) % gw);
    const int oy = (int)(i / ((int64_t)E * gw));
    const float supy = 2.0f * fmaxf(scale_y, 1.0f), supx = 2.0f * fmaxf(scale_x, 1.0f);
    const float invy = 1.0f / fmaxf(scale_y, 1.0f), invx = 1.0f / fmaxf(scale_x, 1.0f);
    const float cy = scale_y * ((float)oy + 0.5f), cx = scale_x * ((float)ox + 0.5f);
    const int y0 = max((int)(cy - supy + 0.5f), 0), y1 = min((int)(cy + supy + 0.5f), M);
    const int x0 = max((int)(cx - supx + 0.5f), 0), x1 = min((int)(cx + supx + 0.5f), M);
    float sy = 0.f, sx = 0.f;
    for (int y = y0; y < y1; ++y) sy += cubic_aa(((float)y - cy + 0.5f) * invy);
    for (int x = x0; x < x1; ++x) sx += cubic_aa(((float)x - cx + 0.5f) * invx);
    float acc = 0.f;
    for (int y = y0; y < y1; ++y) {                      // horizontal pass per row, then the vertical weight (torch's order)
        float row = 0.f;
        for (int x = x0; x < x1; ++x) row += (cubic_aa(((float)x - cx + 0.5f) * invx) / sx) * pos[((int64_t)y * M + x) * E + e];
        acc += (cubic_aa(((float)y - cy + 0.5f) * invy) / sy) * row;
    }
    out[i] = acc;
}

template <typename T, typename InT>
int launch_pe(const void* vol, int n, int H, int W, const void* wp, const float* bias, const float* prefix,
              int n_prefix, const float* pos_patch, int E, float* x, hipStream_t s) {
    const int gh = H / PATCH, gw = W / PATCH, Np = gh * gw;
    const int64_t total = (int64_t)n * Np;
    constexpr int BMP = 32 * TileW<T>::v, BNP = BMP;
    constexpr int sh = 2 * BMP * RowBytes<T>::v;
    auto kern = patch_embed_kernel<T, InT>;
    static mst_lds_once lds_once;
    mst_allow_lds((const void*)kern, sh, &lds_once);
    const dim3 grid((unsigned)((total + BMP - 1) / BMP), E / BNP);
    kern<<<grid, dim3(256), sh, s>>>((const InT*)vol, H, W, gw, Np, total, (const T*)wp, bias, pos_patch, n_prefix, E, x);
    int rc = mst_check_launch("patch_embed");
    if (rc) return rc;
    const int64_t tot = (int64_t)n * n_prefix * E;
    prefix_rows_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(prefix, n_prefix, E, n_prefix + Np, n, x);
    return mst_check_launch("prefix_rows");
}

template <typename T>
int launch_pe_in(const void* vol, int idt, int n, int H, int W, const void* wp, const float* bias,
                 const float* prefix, int n_prefix, const float* pos_patch, int E, float* x, hipStream_t s) {
    switch (idt) {
        case MST_F32: return launch_pe<T, float>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
        case MST_F16: return launch_pe<T, f16_t>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
        case MST_BF16: return launch_pe<T, bf16_t>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
    }
    mst_set_error("patch_embed: bad volume dtype %d", idt);
    return MST_EINVAL;
}

}  // namespace

int launch_patch_embed(const void* vol, int idt, int n, int H, int W, const void* wp, int dt,
                       const float* bias, const float* prefix, int n_prefix, const float* pos_patch,
                       int E, float* x, hipStream_t s) {
    // the reference's own error convention for illegal sizes is a Python AssertionError
    // (patch_embed.py:72-73); the host module raises it before reaching this call.
    MST_CHECK_ARG(H > 0 && W > 0 && H % PATCH == 0 && W % PATCH == 0, "patch_embed: H=%d W=%d must be multiples of 14", H, W);
    MST_CHECK_ARG(E % 128 == 0, "patch_embed: E=%d must be a multiple of 128", E);
    MST_CHECK_ARG(n > 0 && n_prefix >= 1, "patch_embed: n=%d n_prefix=%d", n, n_prefix);
    switch (dt) {
        case MST_F32: return launch_pe_in<float>(vol, idt, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
        case MST_F16: return launch_pe_in<f16_t>(vol, idt, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
        case MST_BF16: return launch_pe_in<bf16_t>(vol, idt, n, H, W, wp, bias, prefix, n_prefix, pos_patch, E, x, s);
    }
    mst_set_error("patch_embed: bad compute dtype %d", dt);
    return MST_EINVAL;
}

int launch_pos_interp(const float* pos, int M, int E, int gh, int gw, double offset, int antialias, float* out,
                      hipStream_t s) {
    MST_CHECK_ARG(M > 0 && E > 0 && gh > 0 && gw > 0, "pos_interp: bad sizes");
    // vision_transformer.py:197-199: scale_factor = (g + offset) / M; torch maps dst->src with 1/scale_factor
    const float sy = (float)(1.0 / (((double)gh + offset) / (double)M));
    const float sx = (float)(1.0 / (((double)gw + offset) / (double)M));
    const int64_t tot = (int64_t)gh * gw * E;
    if (antialias) pos_interp_aa_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(pos, M, E, gh, gw, sy, sx, out);
    else pos_interp_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(pos, M, E, gh, gw, sy, sx, out);
    return mst_check_launch("pos_interp");
}
