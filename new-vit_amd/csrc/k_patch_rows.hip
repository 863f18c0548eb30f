// Patch embedding + tokenisation + the first block's (plain) LayerNorm in ONE pass, for the fused 16-bit pipeline (E = 384):
//   x[s, 1+R+p, :]  = conv14x14(gray->RGB slice s)[p] + bias + pos_patch[p]          (fp32 residual stream; patch_embed.py:68-81,
//                                                                                      vision_transformer.py:213-232)
//   xn[s, 1+R+p, :] = (x - mean) * rstd  in the MFMA operand type                     (block.py:90 norm1; its affine is folded into
//                                                                                      the QKV weights, k_gemm16*.hip)
//   prefix rows (cls + pos[0], registers) of every slice likewise.
// What it replaces: k_patch.hip's 64 x 64 tiles (every patch tile is im2col'ed by the six workgroups of its six channel tiles: 975 MB
// fetched per launch for a 137 MB volume, profiles/r02x_pmc_*) followed by a LayerNorm launch that re-reads the 539 MB it just wrote.
// Here a workgroup owns whole token rows, so the statistics are at hand:
//   * one persistent 8-wave workgroup per CU; wave w keeps the channel-folded kernel W[48 channels][224] as 21 MFMA A-operand
//     fragments in 84 registers (rows fetched in the order that gives a lane 12 consecutive channels, as k_gemm16_wreg.hip);
//   * chunks of 32 patches: thread (patch, ky) loads the 14 pixels of one kernel row (adjacent threads = adjacent 28-byte runs of
//     the volume row) one chunk ahead and writes them as one 16-element k group into a 464-byte-pitch LDS image (two buffers);
//   * 42 v_mfma_f32_16x16x32 per wave and chunk, raw results into an fp32 staging tile; behind a barrier wave w takes rows
//     4w .. 4w+3: + bias + position row (coalesced), two-pass statistics by wave shuffles exactly as layernorm_kernel, 512-byte
//     stores of x and 256-byte stores of xn.
#include "mst_common.h"

namespace {

constexpr int PATCH = 14, KP = 224, EE = 384, CHP = 32, NKT = KP / 32;
constexpr int A_PITCH = 464;                      // 29 sixteen-byte slots per row (odd: fragment reads spread over the banks)
constexpr int A_BYTES = CHP * A_PITCH;
constexpr int S_PITCH = EE * 4 + 16;              // staging row: 1536 bytes + 16
constexpr int S_BYTES = CHP * S_PITCH;
constexpr int LDS_BYTES = 2 * A_BYTES + S_BYTES;   // 79,360

template <typename InT> struct In2;
template <> struct In2<float> { typedef float2 type; };
template <> struct In2<f16_t> { typedef __attribute__((ext_vector_type(2))) f16_t type; };
template <> struct In2<bf16_t> { typedef __attribute__((ext_vector_type(2))) bf16_t type; };

template <typename T, typename InT>
__global__ __launch_bounds__(512) void patch_rows16_kernel(const InT* __restrict__ vol, int H, int W, int gw, int Np, int64_t total,
                                                           const T* __restrict__ wp, const float* __restrict__ bias,
                                                           const float* __restrict__ pos_patch, int n_prefix, float* __restrict__ x,
                                                           T* __restrict__ xn, int nchunks) {
    typedef typename V8<T>::type vec8;
    typedef typename In2<InT>::type in2;
    typedef __attribute__((ext_vector_type(2))) T out2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Stg = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, kq = lane >> 4;
    const int N = n_prefix + Np;

    // ---- the wave's 48 channels of the kernel: tile t row m = frow <-> channel 48 w + 12 (m >> 2) + 4 t + (m & 3)
    vec8 w[3][NKT];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const T* wr = wp + (int64_t)(wave * 48 + 12 * (frow >> 2) + 4 * t + (frow & 3)) * KP + kq * 8;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) w[t][kt] = *reinterpret_cast<const vec8*>(wr + kt * 32);
    }
    float2 bv[3];                                    // bias of the flush layout: channels 128 j + 2 lane, +1
#pragma unroll
    for (int j = 0; j < 3; ++j) bv[j] = *reinterpret_cast<const float2*>(bias + j * 128 + lane * 2);

    // ---- im2col duty of this thread: patch pl of the chunk, kernel row ky
    const bool loader = tid < CHP * PATCH;
    const int pl = tid % CHP, ky = tid / CHP;
    in2 px[7];
    auto gload = [&](int chunk) {
        if (!loader) return;
        int64_t m = (int64_t)chunk * CHP + pl;
        if (m >= total) m = total - 1;
        const int s = (int)(m / Np), p = (int)(m % Np);
        const int py = p / gw, pxx = p % gw;
        const InT* src = vol + ((int64_t)s * H + py * PATCH + ky) * W + pxx * PATCH;
#pragma unroll
        for (int e = 0; e < 7; ++e) px[e] = *reinterpret_cast<const in2*>(src + 2 * e);
    };
    auto lstore = [&](int buf) {
        if (!loader) return;
        T v[16];
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            if constexpr (sizeof(InT) == 4) {
                v[2 * e] = (T)px[e].x;
                v[2 * e + 1] = (T)px[e].y;
            } else {
                v[2 * e] = (T)(float)px[e][0];
                v[2 * e + 1] = (T)(float)px[e][1];
            }
        }
        v[14] = (T)0.f;
        v[15] = (T)0.f;
        vec8* dst = reinterpret_cast<vec8*>(smem + buf * A_BYTES + pl * A_PITCH + ky * 32);
        vec8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lo[e] = v[e];
            hi[e] = v[8 + e];
        }
        dst[0] = lo;
        dst[1] = hi;
    };

    const int a_off = frow * A_PITCH + kq * 16;                                       // + mt*16*A_PITCH + kt*64
    const int st_off = frow * S_PITCH + (wave * 48 + 12 * kq) * 4;                    // + mt*16*S_PITCH + t*16

    int chunk = blockIdx.x;
    if (chunk >= nchunks) return;
    gload(chunk);
    for (int it = 0; chunk < nchunks; chunk += gridDim.x, ++it) {
        const int buf = it & 1;
        lstore(buf);
        if (chunk + (int)gridDim.x < nchunks) gload(chunk + gridDim.x);
        __syncthreads();                                                              // image of this chunk complete; staging free
        f32x4 acc[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* As = smem + buf * A_BYTES + a_off;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const vec8 a0 = *reinterpret_cast<const vec8*>(As + kt * 64);
            const vec8 a1 = *reinterpret_cast<const vec8*>(As + 16 * A_PITCH + kt * 64);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[0][t] = mfma16(w[t][kt], a0, acc[0][t]);
                acc[1][t] = mfma16(w[t][kt], a1, acc[1][t]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(Stg + st_off + mt * 16 * S_PITCH + t * 16) = acc[mt][t];
        __syncthreads();                                                              // staging tile complete
        // ---- rows 4w .. 4w+3: + bias + position, statistics, x and xn
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r;
            const int64_t m = (int64_t)chunk * CHP + row;
            if (m >= total) break;                                                    // wave-uniform
            const int s = (int)(m / Np), p = (int)(m % Np);
            const float* pr = pos_patch + (int64_t)p * EE;
            const int64_t orow = ((int64_t)s * N + n_prefix + p) * EE;
            float2 v[3];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = j * 128 + lane * 2;
                const float2 a = *reinterpret_cast<const float2*>(Stg + row * S_PITCH + c * 4);
                const float2 pv = *reinterpret_cast<const float2*>(pr + c);
                v[j].x = a.x + bv[j].x + pv.x;
                v[j].y = a.y + bv[j].y + pv.y;
                sum += v[j].x + v[j].y;
            }
            const float mean = wave_sum(sum) * (1.0f / EE);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dx = v[j].x - mean, dy = v[j].y - mean;
                q += dx * dx + dy * dy;
            }
            const float rstd = rsqrtf(wave_sum(q) * (1.0f / EE) + 1e-6f);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = j * 128 + lane * 2;
                *reinterpret_cast<float2*>(x + orow + c) = v[j];
                out2 o;
                o[0] = (T)((v[j].x - mean) * rstd);
                o[1] = (T)((v[j].y - mean) * rstd);
                *reinterpret_cast<out2*>(xn + orow + c) = o;
            }
        }
    }
}

// prefix rows (cls + pos[0], register tokens): the same row for every slice, with its plain LayerNorm; one wave per (slice, row)
template <typename T>
__global__ __launch_bounds__(256) void prefix_rows_ln_kernel(const float* __restrict__ prefix, int n_prefix, int N, int n,
                                                             float* __restrict__ x, T* __restrict__ xn) {
    typedef __attribute__((ext_vector_type(2))) T out2;
    const int lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= (int64_t)n * n_prefix) return;
    const int r = (int)(id % n_prefix);
    const int64_t s = id / n_prefix;
    float2 v[3];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j] = *reinterpret_cast<const float2*>(prefix + (int64_t)r * EE + j * 128 + lane * 2);
        sum += v[j].x + v[j].y;
    }
    const float mean = wave_sum(sum) * (1.0f / EE);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float dx = v[j].x - mean, dy = v[j].y - mean;
        q += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / EE) + 1e-6f);
    const int64_t orow = (s * N + r) * EE;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = j * 128 + lane * 2;
        *reinterpret_cast<float2*>(x + orow + c) = v[j];
        out2 o;
        o[0] = (T)((v[j].x - mean) * rstd);
        o[1] = (T)((v[j].y - mean) * rstd);
        *reinterpret_cast<out2*>(xn + orow + c) = o;
    }
}

template <typename T, typename InT>
int launch_t(const void* vol, int n, int H, int W, const void* wp, const float* bias, const float* prefix, int n_prefix,
             const float* pos_patch, float* x, void* xn, hipStream_t s) {
    const int gh = H / PATCH, gw = W / PATCH, Np = gh * gw;
    const int64_t total = (int64_t)n * Np;
    const int nchunks = (int)((total + CHP - 1) / CHP);
    auto kern = patch_rows16_kernel<T, InT>;
    static mst_lds_once lds_once;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int cus = mst_persistent_grid();
    kern<<<dim3(nchunks < cus ? nchunks : cus), dim3(512), LDS_BYTES, s>>>((const InT*)vol, H, W, gw, Np, total, (const T*)wp, bias,
                                                                          pos_patch, n_prefix, x, (T*)xn, nchunks);
    int rc = mst_check_launch("patch_rows16");
    if (rc) return rc;
    const int64_t rows = (int64_t)n * n_prefix;
    prefix_rows_ln_kernel<T><<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(prefix, n_prefix, n_prefix + Np, n, x, (T*)xn);
    return mst_check_launch("prefix_rows_ln");
}

template <typename T>
int launch_in(const void* vol, int idt, int n, int H, int W, const void* wp, const float* bias, const float* prefix, int n_prefix,
              const float* pos_patch, float* x, void* xn, hipStream_t s) {
    switch (idt) {
        case MST_F32: return launch_t<T, float>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, x, xn, s);
        case MST_F16: return launch_t<T, f16_t>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, x, xn, s);
        case MST_BF16: return launch_t<T, bf16_t>(vol, n, H, W, wp, bias, prefix, n_prefix, pos_patch, x, xn, s);
    }
    mst_set_error("patch_rows16: bad volume dtype %d", idt);
    return MST_EINVAL;
}

}  // namespace

// E = 384 and a 16-bit operand type only (the fused pipeline of mst_vit_encode); x fp32 [n*(n_prefix+Np), 384], xn the same rows in dt
int launch_patch_rows16(const void* vol, int idt, int n, int H, int W, const void* wp, int dt, const float* bias, const float* prefix,
                        int n_prefix, const float* pos_patch, float* x, void* xn, hipStream_t s) {
    MST_CHECK_ARG(H > 0 && W > 0 && H % PATCH == 0 && W % PATCH == 0, "patch_rows16: H=%d W=%d must be multiples of 14", H, W);
    MST_CHECK_ARG(n > 0 && n_prefix >= 1 && (int64_t)n * (H / PATCH) * (W / PATCH) < (1ll << 31) - CHP, "patch_rows16: n=%d n_prefix=%d", n,
                  n_prefix);
    if (dt == MST_BF16) return launch_in<bf16_t>(vol, idt, n, H, W, wp, bias, prefix, n_prefix, pos_patch, x, xn, s);
    if (dt == MST_F16) return launch_in<f16_t>(vol, idt, n, H, W, wp, bias, prefix, n_prefix, pos_patch, x, xn, s);
    mst_set_error("patch_rows16: bad operand dtype %d", dt);
    return MST_EINVAL;
}
