// mst_gemm_ex: generic strided, batched fp32 GEMM of the training step on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   C[b] = alpha * A[b] . B[b] (+ beta * C[b]), every operand with explicit element strides and a two-level batch: one entry covers
//   dX = dY.W, dW = dY^T.X, and the four products of the attention backward on the packed q|k|v layout of the forward (what the
//   reference gets from torch.autograd: base_model.py:148-181 `_step`; attention.py:56-66).
// 64 x 64 tile per 256-thread workgroup (four waves 2 x 2, one 32 x 32 accumulator each), K-step 16, LDS double-buffered with the next
// K-step's operands fetched into registers BEFORE the current step's MFMAs (one barrier per step; the round-2 kernel loaded, synchronised,
// multiplied and synchronised again, so every step paid a full HBM/L2 latency -- 23 TFLOP/s at the DINOv2 shapes).  Operand loads are
// 16-byte vectors along whichever dimension is contiguous when strides and base allow (template modes), scalars otherwise.
#include "mst_common.h"

namespace {

struct GemmExArgs {
    const float* A; const float* B; float* C;
    int M, N, K, nb2;
    int64_t sam, sak, sbk, sbn, scm, scn;          // element strides of A[m][k], B[k][n], C[m][n]
    int64_t sa1, sa2, sb1, sb2, sc1, sc2;          // batch strides: batch index = b1 * nb2 + b2
    float alpha, beta;
};

constexpr int LDT = 68;                            // LDS row stride in floats: 16-byte aligned rows, k and k+1 four banks apart

// operand staging modes: 0 = four scalars per thread (any strides), 1 = one float4 along k, 2 = one float4 along m (A) / n (B)
template <int MODE>
struct Stage {
    float r[4];
    int i0 = 0, i1 = 0;                            // MODE 1: (row, k quad)   MODE 2: (k, row quad)
    // T[k][x] tile of a matrix X[x][k] with strides (sx, sk); x0 = tile origin, X = extent
    __device__ __forceinline__ void init(int tid) {
        if (MODE == 1) { i0 = tid >> 2; i1 = (tid & 3) * 4; }
        if (MODE == 2) { i0 = tid >> 4; i1 = (tid & 15) * 4; }
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t sx, int64_t sk, int x0, int X, int k0, int K, int tid, bool kfast) {
        if (MODE == 1) {
            const bool ok = x0 + i0 < X && k0 + i1 < K;
            const float4 v = ok ? *reinterpret_cast<const float4*>(P + (int64_t)(x0 + i0) * sx + (k0 + i1)) : make_float4(0.f, 0.f, 0.f, 0.f);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else if (MODE == 2) {
            const bool ok = x0 + i1 < X && k0 + i0 < K;
            const float4 v = ok ? *reinterpret_cast<const float4*>(P + (int64_t)(k0 + i0) * sk + (x0 + i1)) : make_float4(0.f, 0.f, 0.f, 0.f);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tid + 256 * e;
                const int x = kfast ? idx >> 4 : idx & 63, k = kfast ? idx & 15 : idx >> 6;
                r[e] = (x0 + x < X && k0 + k < K) ? P[(int64_t)(x0 + x) * sx + (int64_t)(k0 + k) * sk] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float (*T)[LDT], int tid, bool kfast) const {
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[i1 + e][i0] = r[e];
        } else if (MODE == 2) {
            *reinterpret_cast<float4*>(&T[i0][i1]) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tid + 256 * e;
                const int x = kfast ? idx >> 4 : idx & 63, k = kfast ? idx & 15 : idx >> 6;
                T[k][x] = r[e];
            }
        }
    }
};

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_ex_kernel(GemmExArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][16][LDT], Bs[2][16][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int b1 = blockIdx.z / g.nb2, b2 = blockIdx.z % g.nb2;
    const float* A = g.A + b1 * g.sa1 + b2 * g.sa2;
    const float* B = g.B + b1 * g.sb1 + b2 * g.sb2;
    float* C = g.C + b1 * g.sc1 + b2 * g.sc2;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const bool a_kfast = g.sak == 1, b_kfast = g.sbn != 1;       // scalar mode: which index runs over adjacent threads (coalescing)
    Stage<AMODE> sa;
    Stage<BMODE> sb;
    sa.init(tid);
    sb.init(tid);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    sa.load(A, g.sam, g.sak, m0, g.M, 0, g.K, tid, a_kfast);
    sb.load(B, g.sbn, g.sbk, n0, g.N, 0, g.K, tid, b_kfast);
    sa.store(As[0], tid, a_kfast);
    sb.store(Bs[0], tid, b_kfast);
    __syncthreads();
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        const int cur = (k0 >> 4) & 1;
        const bool more = k0 + 16 < g.K;
        if (more) {                                              // in flight across the MFMAs below
            sa.load(A, g.sam, g.sak, m0, g.M, k0 + 16, g.K, tid, a_kfast);
            sb.load(B, g.sbn, g.sbk, n0, g.N, k0 + 16, g.K, tid, b_kfast);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][k][wr * 32 + (lane & 31)], Bs[cur][k][wc * 32 + (lane & 31)], acc, 0, 0, 0);
        }
        if (more) {
            sa.store(As[cur ^ 1], tid, a_kfast);
            sb.store(Bs[cur ^ 1], tid, b_kfast);
        }
        __syncthreads();
    }
    const int col = n0 + wc * 32 + (lane & 31);
    if (col >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float* c = C + (int64_t)row * g.scm + (int64_t)col * g.scn;
        *c = g.alpha * acc[r] + (g.beta != 0.f ? g.beta * *c : 0.f);
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int AMODE>
int launch_b(const GemmExArgs& g, int bmode, dim3 grid, hipStream_t s) {
    switch (bmode) {
        case 1: gemm_ex_kernel<AMODE, 1><<<grid, dim3(256), 0, s>>>(g); break;
        case 2: gemm_ex_kernel<AMODE, 2><<<grid, dim3(256), 0, s>>>(g); break;
        default: gemm_ex_kernel<AMODE, 0><<<grid, dim3(256), 0, s>>>(g); break;
    }
    return mst_check_launch("gemm_ex");
}

}  // namespace

int launch_gemm_ex(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                   int64_t scm, int64_t scn, int nb1, int nb2, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1,
                   int64_t sc2, float alpha, float beta, hipStream_t s) {
    MST_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && nb1 > 0 && nb2 > 0, "gemm_ex: bad arguments");
    MST_CHECK_ARG((int64_t)nb1 * nb2 <= 65535, "gemm_ex: batch %d x %d exceeds the grid limit", nb1, nb2);
    GemmExArgs g{A, B, C, M, N, K, nb2, sam, sak, sbk, sbn, scm, scn, sa1, sa2, sb1, sb2, sc1, sc2, alpha, beta};
    // a float4 lies wholly inside or outside the operand when the vectorised extent is a multiple of 4, and is aligned when base and
    // every stride that moves it are multiples of 4 elements
    static const bool vec_ok = !(getenv("MST_GEMM_EX_SCALAR") && atoi(getenv("MST_GEMM_EX_SCALAR")) == 1);
    const bool a_al = vec_ok && al16(A) && sa1 % 4 == 0 && sa2 % 4 == 0, b_al = vec_ok && al16(B) && sb1 % 4 == 0 && sb2 % 4 == 0;
    int amode = 0, bmode = 0;
    if (a_al && sak == 1 && sam % 4 == 0 && K % 4 == 0) amode = 1;
    else if (a_al && sam == 1 && sak % 4 == 0 && M % 4 == 0) amode = 2;
    if (b_al && sbk == 1 && sbn % 4 == 0 && K % 4 == 0) bmode = 1;
    else if (b_al && sbn == 1 && sbk % 4 == 0 && N % 4 == 0) bmode = 2;
    const dim3 grid((N + 63) / 64, (M + 63) / 64, nb1 * nb2);
    switch (amode) {
        case 1: return launch_b<1>(g, bmode, grid, s);
        case 2: return launch_b<2>(g, bmode, grid, s);
    }
    return launch_b<0>(g, bmode, grid, s);
}
