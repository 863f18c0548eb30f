// Input packing in front of the encoder.
//   slices2rgb (reference mst/models/dino.py:10-27; its call at dino.py:129 is commented out): three consecutive gray slices
//   become the three channels of one image, [B,1,D,H,W] -> [B*ceil(D/3), 3, H, W]; when D is not a multiple of 3 the volume is
//   padded along D with its own first 3 - D%3 slices.  Pure index arithmetic, HBM-bound: 16-byte loads and stores.
#include "mst_common.h"

namespace {

// one thread per 16 bytes of output; `hw16` = 16-byte pieces per slice
__global__ void slices2rgb_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int D, int Dp, int64_t hw16, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / hw16, p = i - s * hw16;      // s = output slice index b*Dp + d
        const int64_t b = s / Dp;
        int d = (int)(s - b * Dp);
        if (d >= D) d -= D;                                // padding = the first slices again (dino.py:19-20)
        out[i] = in[(b * D + d) * hw16 + p];
    }
}

__global__ void slices2rgb_scalar_kernel(const char* __restrict__ in, char* __restrict__ out, int D, int Dp, int64_t hw,
                                         int esz, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / hw, p = i - s * hw;
        const int64_t b = s / Dp;
        int d = (int)(s - b * Dp);
        if (d >= D) d -= D;
        const char* src = in + ((b * D + d) * hw + p) * esz;
        char* dst = out + i * esz;
        for (int e = 0; e < esz; ++e) dst[e] = src[e];
    }
}

}  // namespace

int launch_slices2rgb(const void* vol, int dt, int B, int D, int H, int W, void* out, hipStream_t s) {
    const int esz = dt == MST_F32 ? 4 : 2;
    const int pad = (D % 3) ? 3 - D % 3 : 0;
    MST_CHECK_ARG(pad <= D, "slices2rgb: D=%d cannot pad itself to a multiple of 3", D);
    const int Dp = D + pad;
    const int64_t hw = (int64_t)H * W;
    if ((hw * esz) % 16 == 0 && ((uintptr_t)vol % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        const int64_t hw16 = hw * esz / 16, total = (int64_t)B * Dp * hw16;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        slices2rgb_kernel<<<dim3(grid), dim3(256), 0, s>>>((const uint4*)vol, (uint4*)out, D, Dp, hw16, total);
    } else {
        const int64_t total = (int64_t)B * Dp * hw;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        slices2rgb_scalar_kernel<<<dim3(grid), dim3(256), 0, s>>>((const char*)vol, (char*)out, D, Dp, hw, esz, total);
    }
    return mst_check_launch("slices2rgb");
}

// ---------------------------------------------------------------------------------------------------------------------------
// the two array transforms the datasets apply before the model, on the device, so that a volume can stay in HBM
// from the loader to the encoder (reference: mst/data/datasets/augmentations/augmentations_3d.py; torchio 0.19.9 underneath).
//   CropOrPad (l.144-195, deterministic centre: ini = ceil(n/2), fin = n - ini per axis): torchio.Pad = numpy.pad per axis in
//   turn, padding_mode 'minimum' (the datasets' choice: every 1-D line is padded with its own minimum, later axes see the earlier
//   pads) or a constant, then torchio.Crop.
//   ZNormalization (l.40-86), per channel: mask = (x > min) & (x < max) of the UNclamped data; clamp to the
//   torch.quantile(masked values, percentiles / 100) cut-offs (linear interpolation); mean / unbiased std of the masked clamped
//   values; (x - mean) / std everywhere.
// All HBM-bound streaming passes; the exact order statistics come from a radix select (4 rounds of 256-bin histograms per rank,
// all on the device: no host round trip).
namespace {

// fill the pads of ONE axis of a [n0, n1, n2] fp32 array (row-major) with the minimum (or a constant) of each 1-D line's valid
// part [lo, hi).  Lines run along `axis`; the other two coordinates range over [a0, a1) x [b0, b1) (the region np.pad's
// iterative scheme has defined so far).
__global__ void pad_axis_kernel(float* __restrict__ v, int n0, int n1, int n2, int axis, int lo, int hi, int a0, int a1, int b0,
                                int b1, int use_const, float cval) {
    const int64_t nl = (int64_t)(a1 - a0) * (b1 - b0);
    const int n_ax = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
    const int64_t s_ax = axis == 0 ? (int64_t)n1 * n2 : (axis == 1 ? n2 : 1);
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nl; l += (int64_t)gridDim.x * blockDim.x) {
        const int a = a0 + (int)(l / (b1 - b0)), b = b0 + (int)(l % (b1 - b0));
        // the two fixed coordinates, in axis order
        int64_t base;
        if (axis == 0) base = (int64_t)a * n2 + b;                       // (a, b) = (i1, i2)
        else if (axis == 1) base = (int64_t)a * n1 * n2 + b;             // (a, b) = (i0, i2)
        else base = ((int64_t)a * n1 + b) * n2;                          // (a, b) = (i0, i1)
        float m = cval;
        if (!use_const) {
            m = INFINITY;
            for (int i = lo; i < hi; ++i) m = fminf(m, v[base + i * s_ax]);
        }
        for (int i = 0; i < lo; ++i) v[base + i * s_ax] = m;
        for (int i = hi; i < n_ax; ++i) v[base + i * s_ax] = m;
    }
}

// dst[d0+i][d1+j][d2+k] = src[s0+i][s1+j][s2+k] over a [c0, c1, c2] block
__global__ void copy_block_kernel(const float* __restrict__ src, int sn1, int sn2, int s0, int s1, int s2, float* __restrict__ dst,
                                  int dn1, int dn2, int d0, int d1, int d2, int c0, int c1, int c2) {
    const int64_t n = (int64_t)c0 * c1 * c2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % c2), j = (int)((i / c2) % c1), a = (int)(i / ((int64_t)c1 * c2));
        dst[((int64_t)(d0 + a) * dn1 + d1 + j) * dn2 + d2 + k] = src[((int64_t)(s0 + a) * sn1 + s1 + j) * sn2 + s2 + k];
    }
}

// order-preserving map float -> uint32
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// state words (device memory, per channel): see znorm_state
struct ZState {
    float mn, mx;                 // extrema of the raw data
    unsigned long long count;     // masked voxels
    unsigned prefix[4];           // radix-select state of the four order statistics (lo/hi neighbours of the two quantiles)
    unsigned long long rank[4];   // remaining rank inside the current prefix bucket
    unsigned hist[4][256];
    float cut_lo, cut_hi;
    double sum, sq;
    float mean, sd;
    int zero_std;
};

__global__ void z_init_kernel(ZState* st) {
    if (threadIdx.x == 0) {
        st->mn = INFINITY; st->mx = -INFINITY; st->count = 0; st->sum = 0.0; st->sq = 0.0; st->zero_std = 0;
        for (int t = 0; t < 4; ++t) { st->prefix[t] = 0; st->rank[t] = 0; }
    }
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&st->hist[0][0])[i] = 0;
}

__device__ __forceinline__ void atomic_minf(float* a, float v) {   // bit-pattern trick, both signs
    if (v >= 0.f) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned*)a, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_maxf(float* a, float v) {
    if (v >= 0.f) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned*)a, __float_as_uint(v));
}

__global__ void z_minmax_kernel(const float* __restrict__ x, int64_t n, ZState* st) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomic_minf(&st->mn, mn); atomic_maxf(&st->mx, mx); }
}

__global__ void z_count_kernel(const float* __restrict__ x, int64_t n, ZState* st) {
    const float mn = st->mn, mx = st->mx;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        c += (v > mn && v < mx) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&st->count, c);
}

// ranks of the order statistics torch.quantile interpolates between: pos = q (m - 1); floor and ceil
__global__ void z_ranks_kernel(ZState* st, float q_lo, float q_hi) {
    if (threadIdx.x) return;
    const unsigned long long m = st->count;
    if (m == 0) { st->zero_std = 1; return; }
    const double p0 = (double)q_lo * (double)(m - 1), p1 = (double)q_hi * (double)(m - 1);
    st->rank[0] = (unsigned long long)floor(p0);
    st->rank[1] = (unsigned long long)ceil(p0);
    st->rank[2] = (unsigned long long)floor(p1);
    st->rank[3] = (unsigned long long)ceil(p1);
}

// one radix round (8 bits at `shift`) for all four targets: histogram of the masked keys that match each target's prefix
__global__ void z_hist_kernel(const float* __restrict__ x, int64_t n, ZState* st, int shift) {
    __shared__ unsigned h[4][256];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const float mn = st->mn, mx = st->mx;
    const unsigned hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    unsigned pf[4];
    bool own[4];                                         // targets that share a prefix share the first one's histogram
    for (int t = 0; t < 4; ++t) {
        pf[t] = st->prefix[t];
        own[t] = true;
        for (int u = 0; u < t; ++u) own[t] = own[t] && pf[u] != pf[t];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!(v > mn && v < mx)) continue;
        const unsigned k = fkey(v), top = k & hi_mask, b = (k >> shift) & 255u;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (own[t] && top == pf[t]) atomicAdd(&h[t][b], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&(&st->hist[0][0])[i], c);
    }
}

// narrow every target by one digit: find the bucket holding its rank (targets sharing a prefix share the first one's histogram)
__global__ void z_scan_kernel(ZState* st, int shift) {
    if (threadIdx.x == 0) {
        unsigned pf[4];
        for (int t = 0; t < 4; ++t) pf[t] = st->prefix[t];
        for (int t = 0; t < 4; ++t) {
            int src = t;
            for (int u = 0; u < t; ++u)
                if (pf[u] == pf[t]) { src = u; break; }
            unsigned long long r = st->rank[t];
            int b = 0;
            for (; b < 255; ++b) {
                const unsigned c = st->hist[src][b];
                if (r < c) break;
                r -= c;
            }
            st->rank[t] = r;
            st->prefix[t] = pf[t] | ((unsigned)b << shift);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) (&st->hist[0][0])[i] = 0;
}

__global__ void z_cutoffs_kernel(ZState* st, float q_lo, float q_hi) {
    if (threadIdx.x || st->count == 0) return;
    const unsigned long long m = st->count;
    const double p0 = (double)q_lo * (double)(m - 1), p1 = (double)q_hi * (double)(m - 1);
    const float w0 = (float)(p0 - floor(p0)), w1 = (float)(p1 - floor(p1));
    const float a0 = fkey_inv(st->prefix[0]), a1 = fkey_inv(st->prefix[1]);
    const float b0 = fkey_inv(st->prefix[2]), b1 = fkey_inv(st->prefix[3]);
    st->cut_lo = a0 + w0 * (a1 - a0);                   // torch.lerp(lo, hi, weight)
    st->cut_hi = b0 + w1 * (b1 - b0);
}

template <int PASS>   // 0: sum of the masked clamped values; 1: sum of squared deviations
__global__ void z_moments_kernel(const float* __restrict__ x, int64_t n, ZState* st) {
    const float mn = st->mn, mx = st->mx, lo = st->cut_lo, hi = st->cut_hi;
    const float mean = st->mean;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!(v > mn && v < mx)) continue;
        const float c = fminf(fmaxf(v, lo), hi);
        if (PASS == 0) acc += (double)c;
        else { const double d = (double)c - (double)mean; acc += d * d; }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(PASS == 0 ? &st->sum : &st->sq, acc);
}

__global__ void z_finish_kernel(ZState* st, int pass) {
    if (threadIdx.x || st->count == 0) return;
    if (pass == 0) st->mean = (float)(st->sum / (double)st->count);
    else {
        const double var = st->count > 1 ? st->sq / (double)(st->count - 1) : 0.0;    // torch.std: unbiased
        const float sd = (float)sqrt(var);
        if (!(sd > 0.f)) st->zero_std = 1;
        st->sd = sd;
    }
}

__global__ void z_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, const ZState* st) {
    const float lo = st->cut_lo, hi = st->cut_hi, mean = st->mean;
    const float sd = st->sd > 0.f ? st->sd : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (fminf(fmaxf(x[i], lo), hi) - mean) / sd;        // tensor -= mean; tensor /= std
}

inline unsigned sgrid(int64_t n) {
    const int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

size_t znorm_state_bytes(void) { return (sizeof(ZState) + 255) / 256 * 256; }

int launch_pad_axis(float* v, int n0, int n1, int n2, int axis, int lo, int hi, int a0, int a1, int b0, int b1, int use_const,
                    float cval, hipStream_t s) {
    const int64_t nl = (int64_t)(a1 - a0) * (b1 - b0);
    if (nl <= 0) return MST_OK;
    pad_axis_kernel<<<dim3(sgrid(nl)), dim3(256), 0, s>>>(v, n0, n1, n2, axis, lo, hi, a0, a1, b0, b1, use_const, cval);
    return mst_check_launch("pad_axis");
}

int launch_copy_block(const float* src, int sn1, int sn2, int s0, int s1, int s2, float* dst, int dn1, int dn2, int d0, int d1,
                      int d2, int c0, int c1, int c2, hipStream_t s) {
    const int64_t n = (int64_t)c0 * c1 * c2;
    if (n <= 0) return MST_OK;
    copy_block_kernel<<<dim3(sgrid(n)), dim3(256), 0, s>>>(src, sn1, sn2, s0, s1, s2, dst, dn1, dn2, d0, d1, d2, c0, c1, c2);
    return mst_check_launch("copy_block");
}

int launch_znorm(const float* x, int64_t n, float q_lo, float q_hi, float* y, void* state, hipStream_t s) {
    ZState* st = (ZState*)state;
    const dim3 g(sgrid(n)), b(256);
    z_init_kernel<<<1, 256, 0, s>>>(st);
    z_minmax_kernel<<<g, b, 0, s>>>(x, n, st);
    z_count_kernel<<<g, b, 0, s>>>(x, n, st);
    z_ranks_kernel<<<1, 64, 0, s>>>(st, q_lo, q_hi);
    for (int shift = 24; shift >= 0; shift -= 8) {
        z_hist_kernel<<<g, b, 0, s>>>(x, n, st, shift);
        z_scan_kernel<<<1, 256, 0, s>>>(st, shift);
    }
    z_cutoffs_kernel<<<1, 64, 0, s>>>(st, q_lo, q_hi);
    z_moments_kernel<0><<<g, b, 0, s>>>(x, n, st);
    z_finish_kernel<<<1, 64, 0, s>>>(st, 0);
    z_moments_kernel<1><<<g, b, 0, s>>>(x, n, st);
    z_finish_kernel<<<1, 64, 0, s>>>(st, 1);
    z_apply_kernel<<<g, b, 0, s>>>(x, y, n, st);
    return mst_check_launch("znorm");
}
