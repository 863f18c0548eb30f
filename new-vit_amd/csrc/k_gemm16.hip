// C[M,N] = epi(A[M,K] . W[N,K]^T + bias)  with 16-bit (bf16 / fp16) MFMA operands, fp32 accumulate.
//
// Both operands are K-contiguous (A row-major activations, W as nn.Linear stores it), so MFMA
// fragments are 16-byte row segments.  Workgroup = 256 threads = 4 waves (2x2), tile 128x128x64,
// each wave 64x64 = 4x4 v_mfma_f32_16x16x32.  Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR
// round trip) into a 2-deep ring; the LDS image is lane-linear, so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and again on
// the read (cdna guide rule 21).  W plays the MFMA "A" role so that each lane ends up with 4
// consecutive output columns of one row: epilogue accesses are 16 B (fp32) / 8 B (16-bit) per lane.
// 1-D grid with an XCD-aware remap: tiles that share an A panel run on one XCD's L2.
//
// Reference arithmetic: F.linear at attention.py:58,67; mlp.py:35-38; block.py:90-94.
#include <stdlib.h>

#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage
// TN: the N extent of the tile, 128 (default) or 64.  64 halves a workgroup's work and doubles the grid: taken when the 128-wide grid
// would leave CUs idle (a few thousand token rows against N = 384: 99 workgroups on 256 CUs), wave tile 64 x 32.

// NS: stages of the LDS ring (2 = double buffer, what every launch uses).  A 4-stage ring behind a COUNTED vmcnt (the newer stages stay in
// flight) for the one-workgroup-per-CU grids of TN = 64 was measured: 0.994 vs 0.992 ms per 16 x 224^2 forward -- no gain, not dispatched.
template <typename T, int EPI, typename OutT, int TN = 128, int NS = 2>
__global__ __launch_bounds__(256) void gemm16_kernel(const T* __restrict__ A, int64_t lda,
                                                     const T* __restrict__ W, int64_t ldw,
                                                     const float* __restrict__ bias, OutT* C,
                                                     int64_t ldc, int M, int N, int K,
                                                     const float* __restrict__ gamma, float col_scale,
                                                     int scale_cols, int tiles_n, int nwg, int64_t split_stride) {
    typedef typename V8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // split-K (launch_gemm16_splitk): blockIdx.y owns the K columns [y K, (y + 1) K) of both operands and its own fp32 partial C
    A += (int64_t)blockIdx.y * K;
    W += (int64_t)blockIdx.y * K;
    C += (int64_t)blockIdx.y * split_stride;
    constexpr int NI = TN / 32;              // 16-column accumulator blocks per wave (wave tile 64 x TN/2) = W pieces per wave and stage
    constexpr int W_BYTES = TN * BK * 2;
    char* const As = smem;                    // [NS][128][64] T
    char* const Ws = smem + NS * TILE_BYTES;  // [NS][TN][64] T

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * TN;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- LDS-DMA source addresses: instruction i of this wave fills tile rows rb*8 .. rb*8+7
    const T* asrc[4];
    const T* wsrc[NI];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);  // chunk whose swizzled home is LDS slot lane&7
        int am = m0 + r;
        am = am < M ? am : M - 1;
        asrc[i] = A + (int64_t)am * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave * NI + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[i] = W + (int64_t)(n0 + r) * ldw + c * 8;
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i]), LDS_PTR(As + buf * TILE_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
            asrc[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i]), LDS_PTR(Ws + buf * W_BYTES + (wave * NI + i) * 1024), 16, 0, 0);
            wsrc[i] += BK;
        }
    };

    // accumulators start at the bias (no bias load / vmcnt drain left in the epilogue)
    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias) b0 = *reinterpret_cast<const f32x4*>(bias + n0 + wn * (TN / 2) + i * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = b0;
    }

    const int sw = (lane >> 1) & 7;  // ((row >> 1) & 7) for row = 16*k + (lane & 15)
    const int a_row_off = (wm * 64 + (lane & 15)) * 128;
    const int w_row_off = (wn * (TN / 2) + (lane & 15)) * 128;

    const int nk = K / BK;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) stage(p);
    for (int t = 0; t < nk; ++t) {
        // stage t has landed once at most the NS - 2 stages behind it are outstanding (LDS-DMA completes in issue order); near the
        // end fewer are in flight: wait for everything
        if (NS > 2 && t + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (4 + NI)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // stage t is visible to every wave; every wave is done reading the buffer the next stage overwrites
        if (t + NS - 1 < nk) stage((t + NS - 1) % NS);
        const char* Ab = As + (t % NS) * TILE_BYTES;
        const char* Wb = Ws + (t % NS) * W_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + (lane >> 4)) ^ sw) * 16;
            vec8 af[4], wf[NI];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                af[j] = *reinterpret_cast<const vec8*>(Ab + a_row_off + j * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                wf[i] = *reinterpret_cast<const vec8*>(Wb + w_row_off + i * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], af[j], acc[i][j]);
        }
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = m0+wm*64+j*16+(lane&15), n = n0+wn*64+i*16+(lane>>4)*4.
    // Interior tiles run branch-free (per-lane `m < M` branches make hipcc drain vmcnt(0) before every store).
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * (TN / 2) + i * 16 + (lane >> 4) * 4;
            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (EPI == MST_EPI_RESIDUAL && gamma) gv = *reinterpret_cast<const float4*>(gamma + n);
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = (n + r < scale_cols) ? col_scale : 1.0f;
            float4 xv[4];
            if constexpr (EPI == MST_EPI_RESIDUAL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 64 + j * 16 + (lane & 15);
                    if (FULL || m < M) xv[j] = *reinterpret_cast<const float4*>(C + (int64_t)m * ldc + n);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 64 + j * 16 + (lane & 15);
                if (!FULL && m >= M) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] *= sc[r];
                    if (EPI == MST_EPI_BIAS_GELU) v[r] = (sizeof(OutT) == 2) ? gelu_fast(v[r]) : gelu_erf(v[r]);
                    if (EPI == MST_EPI_BIAS_RELU) v[r] = fmaxf(v[r], 0.f);
                }
                OutT* cp = C + (int64_t)m * ldc + n;
                if constexpr (EPI == MST_EPI_RESIDUAL) {
                    float4 o;
                    o.x = xv[j].x + gv.x * v[0];
                    o.y = xv[j].y + gv.y * v[1];
                    o.z = xv[j].z + gv.z * v[2];
                    o.w = xv[j].w + gv.w * v[3];
                    *reinterpret_cast<float4*>(cp) = o;
                } else if constexpr (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    typedef __attribute__((ext_vector_type(4))) OutT o4;
                    o4 pk;
                    pk[0] = (OutT)v[0];
                    pk[1] = (OutT)v[1];
                    pk[2] = (OutT)v[2];
                    pk[3] = (OutT)v[3];
                    *reinterpret_cast<o4*>(cp) = pk;
                }
            }
        }
    };
    if (m0 + BM <= M) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

template <typename T, int EPI, typename OutT>
int launch_t(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
             int64_t ldc, int64_t M, int N, int K, const float* gamma, float col_scale, int scale_cols,
             hipStream_t s) {
    const int tiles_m = (int)((M + BM - 1) / BM);
    static const bool narrow_ok = !(getenv("MST_GEMM16_NARROW") && atoi(getenv("MST_GEMM16_NARROW")) == 0);
    if (narrow_ok && tiles_m * (N / BN) < 224) {          // the 128-wide grid would leave CUs idle: 128 x 64 tiles, twice the workgroups
        const int tiles_n = N / 64, nwg = tiles_m * tiles_n;
        static mst_lds_once lds_once_n;
        auto kern = gemm16_kernel<T, EPI, OutT, 64, 2>;
        constexpr int lds = 2 * TILE_BYTES + 2 * 64 * BK * 2;
        mst_allow_lds((const void*)kern, lds, &lds_once_n);
        kern<<<dim3(nwg), dim3(256), lds, s>>>((const T*)A, lda, (const T*)W, ldw, bias, (OutT*)C, ldc, (int)M, N, K, gamma, col_scale,
                                                scale_cols, tiles_n, nwg, 0);
        return mst_check_launch("gemm16 (128 x 64)");
    }
    static mst_lds_once lds_once;
    auto kern = gemm16_kernel<T, EPI, OutT>;
    mst_allow_lds((const void*)kern, 4 * TILE_BYTES, &lds_once);
    const int tiles_n = N / BN;
    const int nwg = tiles_m * tiles_n;
    kern<<<dim3(nwg), dim3(256), 4 * TILE_BYTES, s>>>((const T*)A, lda, (const T*)W, ldw, bias, (OutT*)C, ldc,
                                                       (int)M, N, K, gamma, col_scale, scale_cols, tiles_n, nwg, 0);
    return mst_check_launch("gemm16");
}

template <typename T>
int launch_split(const void* A, int64_t lda, const void* W, int64_t ldw, float* Cpart, int64_t ldc, int64_t M, int N, int Kc, int splits,
                 int64_t split_stride, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = gemm16_kernel<T, MST_EPI_BIAS, float>;
    mst_allow_lds((const void*)kern, 4 * TILE_BYTES, &lds_once);
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = N / BN;
    const int nwg = tiles_m * tiles_n;
    kern<<<dim3(nwg, splits), dim3(256), 4 * TILE_BYTES, s>>>((const T*)A, lda, (const T*)W, ldw, nullptr, Cpart, ldc, (int)M, N, Kc, nullptr,
                                                               1.f, 0, tiles_n, nwg, split_stride);
    return mst_check_launch("gemm16_splitk");
}

template <typename T>
int dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int cdt,
             int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma, float cs, int sc,
             hipStream_t s) {
    const bool f32out = (cdt == MST_F32);
    switch (epi) {
        case MST_EPI_BIAS:
            return f32out ? launch_t<T, MST_EPI_BIAS, float>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s)
                          : launch_t<T, MST_EPI_BIAS, T>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
        case MST_EPI_BIAS_GELU:
            return f32out ? launch_t<T, MST_EPI_BIAS_GELU, float>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s)
                          : launch_t<T, MST_EPI_BIAS_GELU, T>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
        case MST_EPI_BIAS_RELU:
            return f32out ? launch_t<T, MST_EPI_BIAS_RELU, float>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s)
                          : launch_t<T, MST_EPI_BIAS_RELU, T>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
        case MST_EPI_RESIDUAL:
            return launch_t<T, MST_EPI_RESIDUAL, float>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
    }
    mst_set_error("gemm16: bad epilogue %d", epi);
    return MST_EINVAL;
}

}  // namespace

// Cpart[z] = A[:, z Kc : (z + 1) Kc] . W[:, z Kc : (z + 1) Kc]^T for z < splits (fp32 partial products, split_stride elements apart; the caller
// sums them: mst_colsum over a [splits, M * N] view).  The shape of d weight = dY^T . X: a small [N_out, N_in] output reduced over
// thousands of token rows, which one workgroup per output tile would walk on a handful of CUs.
int launch_gemm16_splitk(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, float* Cpart, int64_t ldc, int64_t M, int N, int K,
                         int splits, int64_t split_stride, hipStream_t s) {
    MST_CHECK_ARG(A && W && Cpart && M > 0 && splits > 0 && splits <= 65535, "gemm16_splitk: bad arguments");
    MST_CHECK_ARG(K % splits == 0 && (K / splits) % BK == 0, "gemm16_splitk: K=%d must split into %d multiples of %d", K, splits, BK);
    MST_CHECK_ARG(N > 0 && N % BN == 0, "gemm16_splitk: N=%d must be a multiple of %d", N, BN);
    MST_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)Cpart & 15) == 0,
                  "gemm16_splitk: lda/ldw must be multiples of 8, ldc of 4, bases 16-byte aligned");
    MST_CHECK_ARG(split_stride >= M * ldc && split_stride % 4 == 0, "gemm16_splitk: split_stride");
    if (dt == MST_BF16) return launch_split<bf16_t>(A, lda, W, ldw, Cpart, ldc, M, N, K / splits, splits, split_stride, s);
    if (dt == MST_F16) return launch_split<f16_t>(A, lda, W, ldw, Cpart, ldc, M, N, K / splits, splits, split_stride, s);
    mst_set_error("gemm16_splitk: bad operand dtype %d", dt);
    return MST_EINVAL;
}

int launch_gemm16(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias,
                  void* C, int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                  float col_scale, int scale_cols, hipStream_t s) {
    MST_CHECK_ARG(K > 0 && K % BK == 0, "gemm16: K=%d must be a multiple of %d", K, BK);
    MST_CHECK_ARG(N > 0 && N % BN == 0, "gemm16: N=%d must be a multiple of %d", N, BN);
    MST_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "gemm16: lda/ldw must be multiples of 8, ldc of 4");
    MST_CHECK_ARG(M < (1ll << 31) - BM, "gemm16: M too large");
    MST_CHECK_ARG(cdt == MST_F32 || cdt == dt, "gemm16: C dtype must be f32 or the operand dtype");
    MST_CHECK_ARG(epi != MST_EPI_RESIDUAL || cdt == MST_F32, "gemm16: residual epilogue needs f32 C");
    MST_CHECK_ARG(epi != MST_EPI_RESIDUAL_RELU, "gemm16: the residual + ReLU epilogue exists for f32 operands only");
    if (M <= 0) return MST_OK;
    static const bool big_ok = !(getenv("MST_GEMM_BIG") && atoi(getenv("MST_GEMM_BIG")) == 0);
    // measured on MI355X (tools/bench_gemm.py): the read-modify-write epilogue of the residual GEMMs is
    // HBM-bound and overlaps better with two small workgroups per CU; the others favour the big tile
    static const bool mid_ok = !(getenv("MST_GEMM_MID") && atoi(getenv("MST_GEMM_MID")) == 0);
    static const bool wreg_ok = !(getenv("MST_GEMM_WREG") && atoi(getenv("MST_GEMM_WREG")) == 0);
    // the weights-in-registers kernel moves 16-byte pieces on every operand (LDS-DMA of A, vector loads of W, vector stores of C): a
    // caller of the C ABI with an odd row pitch or a sliced pointer falls back to the tiled kernels
    const bool al16 = lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0;
    if (wreg_ok && al16 && gemm16_wreg_applicable(M, N, K, dt, cdt, epi, scale_cols, lda, ldc))
        return launch_gemm16_wreg(A, dt, lda, W, ldw, bias, C, ldc, M, N, col_scale, scale_cols, s);
    if (mid_ok && gemm16_mid_applicable(M, N, K, dt, cdt, epi))
        return launch_gemm16_mid(A, dt, lda, W, ldw, bias, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    if (big_ok && epi != MST_EPI_RESIDUAL && gemm16_big_applicable(M, N, K))
        return launch_gemm16_big(A, dt, lda, W, ldw, bias, C, cdt, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    if (dt == MST_BF16) return dispatch<bf16_t>(A, lda, W, ldw, bias, C, cdt, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    if (dt == MST_F16) return dispatch<f16_t>(A, lda, W, ldw, bias, C, cdt, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    mst_set_error("gemm16: bad operand dtype %d", dt);
    return MST_EINVAL;
}
