// Small fp32 kernels of the across-slice stage (dino.py:138-153; transformer_blocks.py:210-295)
// and of the saliency read-outs (dino.py:173-202).  Sequence length is 1+D <= 257 and the whole
// stage is ~0.1 GFLOP per volume: these kernels are latency-bound, kept in exact fp32 on the VALU.
#include "mst_common.h"

namespace {

// xs[b][0] = cls; xs[b][1+d] = emb[b*D+d] (+ slice_pos_emb[d])        dino.py:140-145
__global__ void slice_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                    const float* __restrict__ pos, int B, int D, int E, float* __restrict__ xs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = (int64_t)B * (D + 1) * E;
    if (i >= tot) return;
    const int e = (int)(i % E);
    const int l = (int)((i / E) % (D + 1));
    const int64_t b = i / ((int64_t)E * (D + 1));
    float v;
    if (l == 0) v = cls[e];
    else {
        v = emb[(b * D + (l - 1)) * E + e];
        if (pos) v += pos[(int64_t)(l - 1) * E + e];
    }
    xs[i] = v;
}

// One workgroup per (batch, head): K and V of the head in LDS (RoPE applied to q and k on the fly:
// rotary_embedding_torch.py:38-62,159-173), one query row per thread, three passes over the keys
// (max, sum + PV, optional probability write).  Masked keys score -inf (transformer_blocks.py:244-252).
template <int HD>
__global__ __launch_bounds__(256) void slice_attn_kernel(const float* __restrict__ qkv, int L, int heads,
                                                         const uint8_t* __restrict__ mask,
                                                         const float* __restrict__ rope,
                                                         const float* __restrict__ liere,
                                                         float* __restrict__ out, float* __restrict__ probs) {
    extern __shared__ float sm[];
    float* Ks = sm;                 // [L][HD]
    float* Vs = sm + (size_t)L * HD;  // [L][HD]
    float* Mb = Vs + (size_t)L * HD;  // [L] additive mask
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int E = heads * HD, ld = 3 * E;
    const float* base = qkv + (int64_t)b * L * ld;
    if (liere) {
        // LieRE as the reference runs it (rotary_embedding_torch.py:347-396 + the view at transformer_blocks.py:
        // 263-264): every q/k vector is multiplied by one rotation R [HD,HD]; the rotated tensor stays in
        // [L, heads, HD] order and is then viewed as [heads, L, HD], so pseudo-head h, position j holds the
        // vector of (token, head) pair number r = h*L + j of that order.  V keeps its own head.
        for (int i = tid; i < L * HD; i += 256) {
            const int j = i / HD, d = i % HD;
            const int r = h * L + j, t = r / heads, hh = r % heads;
            const float* kp = base + (int64_t)t * ld + E + hh * HD;
            float a = 0.f;
#pragma unroll 8
            for (int e = 0; e < HD; ++e) a = fmaf(liere[d * HD + e], kp[e], a);
            Ks[i] = a;
            Vs[i] = base[(int64_t)j * ld + 2 * E + h * HD + d];
        }
    } else
    for (int i = tid; i < L * (HD / 2); i += 256) {
        const int j = i / (HD / 2), p = i % (HD / 2);
        const float* kp = base + (int64_t)j * ld + E + h * HD + 2 * p;
        float k0 = kp[0], k1 = kp[1];
        if (rope) {
            float sn, cs;
            sincosf((float)j * rope[p], &sn, &cs);
            const float t0 = k0 * cs - k1 * sn, t1 = k1 * cs + k0 * sn;
            k0 = t0;
            k1 = t1;
        }
        Ks[j * HD + 2 * p] = k0;
        Ks[j * HD + 2 * p + 1] = k1;
        Vs[j * HD + 2 * p] = kp[E];
        Vs[j * HD + 2 * p + 1] = kp[E + 1];
    }
    for (int j = tid; j < L; j += 256) {
        // column 0 is the CLS token, never padded (dino.py:147-150)
        Mb[j] = (mask && j > 0 && mask[(int64_t)b * (L - 1) + (j - 1)]) ? -INFINITY : 0.f;
    }
    __syncthreads();
    const float scale = sqrtf(1.0f / (float)HD);  // transformer_blocks.py:268
    for (int qi = tid; qi < L; qi += 256) {
        float q[HD];
        const float* qp = base + (int64_t)qi * ld + h * HD;
        if (liere) {
            const int r = h * L + qi;
            qp = base + (int64_t)(r / heads) * ld + (r % heads) * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                float a = 0.f;
                for (int e = 0; e < HD; ++e) a = fmaf(liere[d * HD + e], qp[e], a);
                q[d] = a * scale;
            }
        } else
#pragma unroll
        for (int p = 0; p < HD / 2; ++p) {
            float q0 = qp[2 * p], q1 = qp[2 * p + 1];
            if (rope) {
                float sn, cs;
                sincosf((float)qi * rope[p], &sn, &cs);
                const float t0 = q0 * cs - q1 * sn, t1 = q1 * cs + q0 * sn;
                q0 = t0;
                q1 = t1;
            }
            q[2 * p] = q0 * scale;
            q[2 * p + 1] = q1 * scale;
        }
        float mx = -INFINITY;
        for (int j = 0; j < L; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j * HD + d], a);
            mx = fmaxf(mx, a + Mb[j]);
        }
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        float sum = 0.f;
        for (int j = 0; j < L; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j * HD + d], a);
            const float p = expf(a + Mb[j] - mx);
            sum += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, Vs[j * HD + d], acc[d]);
        }
        const float inv = 1.0f / sum;
        float* op = out + ((int64_t)b * L + qi) * E + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
        if (probs) {
            float* pp = probs + (((int64_t)b * heads + h) * L + qi) * L;
            for (int j = 0; j < L; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j * HD + d], a);
                pp[j] = expf(a + Mb[j] - mx) * inv;
            }
        }
    }
}

// R_blk = exp(A), A[i][j] = -A[j][i] = sum_p p * v[(i(i-1)/2 + j) * P + p] for i > j (flat_to_skew + the
// position-index contraction of rotary_embedding_torch.py:319-326,364-366), written into the block diagonal
// of R [hd, hd].  One workgroup of n*n threads per block; fp64 scaling-and-squaring with a degree-18 Taylor
// polynomial at ||A/2^s||_1 <= 0.5 (truncation < 1e-21), so R is orthogonal to ~1e-13 before the fp32 store.
__global__ void liere_expm_kernel(const float* __restrict__ vars, int n, int P, int hd, float* __restrict__ R) {
    __shared__ double A[16 * 16], T[16 * 16], X[16 * 16], red[16];
    __shared__ int sq;
    const int blk = blockIdx.x, tid = threadIdx.x, i = tid / n, j = tid % n;
    const float* v = vars + (int64_t)blk * (n * (n - 1) / 2) * P;
    double a = 0.0;
    if (i != j) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        const float* row = v + (int64_t)(hi * (hi - 1) / 2 + lo) * P;
        for (int p = 0; p < P; ++p) a += (double)row[p] * (double)p;
        if (i < j) a = -a;
    }
    A[tid] = a;
    __syncthreads();
    if (tid < n) {  // 1-norm: max column sum
        double c = 0.0;
        for (int r = 0; r < n; ++r) c += fabs(A[r * n + tid]);
        red[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        double nm = 0.0;
        for (int c = 0; c < n; ++c) nm = fmax(nm, red[c]);
        int s = 0;
        while (nm > 0.5 && s < 60) { nm *= 0.5; ++s; }
        sq = s;
    }
    __syncthreads();
    const int s = sq;
    a = ldexp(a, -s);
    A[tid] = a;
    // Horner: T = I + A/1 (I + A/2 (I + ... (I + A/18)))
    T[tid] = (i == j ? 1.0 : 0.0) + a / 18.0;
    __syncthreads();
    for (int k = 17; k >= 1; --k) {
        double acc = 0.0;
        for (int e = 0; e < n; ++e) acc += A[i * n + e] * T[e * n + j];
        X[tid] = (i == j ? 1.0 : 0.0) + acc / (double)k;
        __syncthreads();
        T[tid] = X[tid];
        __syncthreads();
    }
    for (int q = 0; q < s; ++q) {
        double acc = 0.0;
        for (int e = 0; e < n; ++e) acc += T[i * n + e] * T[e * n + j];
        __syncthreads();
        T[tid] = acc;
        __syncthreads();
    }
    R[(int64_t)(blk * n + i) * hd + blk * n + j] = (float)T[tid];
}

__global__ void rows_copy_kernel(const float* __restrict__ src, int64_t ss, float* __restrict__ dst, int64_t ds,
                                 int rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int c = (int)(i % cols);
    const int64_t r = i / cols;
    dst[r * ds + c] = src[r * ss + c];
}

// 'average' fusion: x.mean(dim=1)   dino.py:156-157
__global__ void mean_slices_kernel(const float* __restrict__ x, int B, int D, int E, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * E) return;
    const int b = i / E, e = i % E;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += x[((int64_t)b * D + d) * E + e];
    out[i] = s / (float)D;
}

// get_slice_attention (dino.py:173-187): CLS row of the slice map, slices only, renormalised per head,
// mean over heads -> sa[n].   One block per batch element.
__global__ void slice_readout_kernel(const float* __restrict__ sp, int D, int sheads, float* __restrict__ sa) {
    extern __shared__ float sm[];  // [sheads] row sums
    const int b = blockIdx.x, L = D + 1;
    const float* row0 = sp + (int64_t)b * sheads * L * L;  // head h row 0 at h*L*L
    for (int h = threadIdx.x; h < sheads; h += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += row0[(int64_t)h * L * L + 1 + d];
        sm[h] = s;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int h = 0; h < sheads; ++h) a += row0[(int64_t)h * L * L + 1 + d] / sm[h];
        sa[(int64_t)b * D + d] = a / (float)sheads;
    }
}

// get_plane_attention / get_attention_maps (dino.py:189-202): per (slice, head): patch columns of the
// CLS row, first patch zeroed, renormalised; maps = slice_attn[n] * plane.
__global__ __launch_bounds__(256) void plane_readout_kernel(const float* __restrict__ cp, int heads, int N, int skip,
                                                            const float* __restrict__ sa,
                                                            float* __restrict__ plane, float* __restrict__ maps) {
    __shared__ float red[4];
    const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int Np = N - skip;
    const float* row = cp + ((int64_t)n * heads + h) * N + skip;
    float s = 0.f;
    for (int p = 1 + tid; p < Np; p += 256) s += row[p];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    const float w = sa ? sa[n] : 0.f;
    const int64_t o = ((int64_t)n * heads + h) * Np;
    for (int p = tid; p < Np; p += 256) {
        const float v = ((p == 0) ? 0.f : row[p]) * inv;   // a one-patch grid gives 0 * inf = NaN, like the reference's 0 / 0
        if (plane) plane[o + p] = v;
        if (maps) maps[o + p] = w * v;
    }
}

}  // namespace

int launch_slice_tokens(const float* emb, const float* cls, const float* pos, int B, int D, int E, float* xs, hipStream_t s) {
    const int64_t tot = (int64_t)B * (D + 1) * E;
    slice_tokens_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(emb, cls, pos, B, D, E, xs);
    return mst_check_launch("slice_tokens");
}

int launch_slice_attn(const float* qkv, int B, int L, int heads, int hd, const uint8_t* mask, const float* rope,
                      const float* liere, float* out, float* probs, hipStream_t s) {
    const size_t sh = ((size_t)2 * L * hd + L) * sizeof(float);
    MST_CHECK_ARG(sh <= 160 * 1024, "slice_attn: L=%d head_dim=%d does not fit LDS", L, hd);
    const dim3 grid(B, heads), block(256);
#define SA_CASE(HD)                                                                                              \
    case HD: {                                                                                                   \
        auto kern = slice_attn_kernel<HD>;                                                                       \
        if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        kern<<<grid, block, sh, s>>>(qkv, L, heads, mask, rope, liere, out, probs);                                     \
        break;                                                                                                   \
    }
    switch (hd) {
        SA_CASE(4) SA_CASE(8) SA_CASE(16) SA_CASE(32) SA_CASE(64) SA_CASE(128)     // 128: 16 heads over the 2048-wide embeddings of a bottleneck ResNet
        default: mst_set_error("slice_attn: head_dim=%d unsupported (4,8,16,32,64,128)", hd); return MST_EINVAL;
    }
#undef SA_CASE
    return mst_check_launch("slice_attn");
}

int launch_liere_rotation(const float* vars, int n_blocks, int n, int P, float* R, hipStream_t s) {
    MST_CHECK_ARG(n >= 2 && n <= 16 && n_blocks >= 1 && P >= 1, "liere_rotation: block size %d (2..16), %d blocks", n, n_blocks);
    const int hd = n * n_blocks;
    if (hipMemsetAsync(R, 0, sizeof(float) * hd * hd, s) != hipSuccess) {
        mst_set_error("liere_rotation: memset failed");
        return MST_ELAUNCH;
    }
    liere_expm_kernel<<<dim3(n_blocks), dim3(n * n), 0, s>>>(vars, n, P, hd, R);
    return mst_check_launch("liere_expm");
}

int launch_rows_copy(const float* src, int64_t ss, float* dst, int64_t ds, int rows, int cols, hipStream_t s) {
    const int64_t tot = (int64_t)rows * cols;
    rows_copy_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s>>>(src, ss, dst, ds, rows, cols);
    return mst_check_launch("rows_copy");
}

int launch_mean_slices(const float* x, int B, int D, int E, float* out, hipStream_t s) {
    mean_slices_kernel<<<dim3((B * E + 255) / 256), dim3(256), 0, s>>>(x, B, D, E, out);
    return mst_check_launch("mean_slices");
}

int launch_readout(const float* cls_probs, const float* slice_probs, int B, int D, int heads, int N, int R, int sheads,
                   float* plane, float* slice_attn, float* maps, hipStream_t s) {
    MST_CHECK_ARG(!maps || (slice_attn && slice_probs), "readout: maps needs slice_probs and a slice_attn buffer");
    if (slice_probs && slice_attn) {
        slice_readout_kernel<<<dim3(B), dim3(256), sheads * sizeof(float), s>>>(slice_probs, D, sheads, slice_attn);
        int rc = mst_check_launch("slice_readout");
        if (rc) return rc;
    }
    if (cls_probs && (plane || maps)) {
        // gridDim.y is limited to 65535: walk the slices in pieces (B*D can exceed it for small slices)
        const int64_t n = (int64_t)B * D, Np = N - 1 - R;
        for (int64_t n0 = 0; n0 < n; n0 += 65535) {
            const int c = (int)((n - n0 < 65535) ? n - n0 : 65535);
            plane_readout_kernel<<<dim3(heads, c), dim3(256), 0, s>>>(cls_probs + n0 * heads * N, heads, N, 1 + R,
                                                                      maps ? slice_attn + n0 : nullptr,
                                                                      plane ? plane + n0 * heads * Np : nullptr,
                                                                      maps ? maps + n0 * heads * Np : nullptr);
            int rc = mst_check_launch("plane_readout");
            if (rc) return rc;
        }
    }
    return MST_OK;
}
