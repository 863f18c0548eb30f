// Attention rollout (reference dino.py:204-212, `get_attention_cls`): R = A_last; for A in reversed(maps[:-1]):
// R = A @ R, on the full [n, heads, N, N] fp32 maps.  One launch per chain link: a batched row-major
// C[b] = A[b] . B[b] in exact fp32 (v_mfma_f32_32x32x2_f32), N arbitrary (257, 1297, 1370 ...: no alignment
// of the row pitch can be assumed, so the staging loads are scalar and fully bounds-checked, zero-filled).
//
// Tile 128x128x16, 4 waves (2x2) of 2x2 MFMA tiles.  The MFMA computes D = B^T . A^T (rows of D = output
// columns), so each lane owns 4 consecutive output columns per accumulator quad and the B tile can sit
// in LDS exactly as it lies in memory ([k][n], lanes read consecutive n: conflict-free).
#include "mst_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDA = 17, LDB = BN + 4;

__global__ __launch_bounds__(256) void bmm32_nn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, int tiles_m,
                                                       int tiles_n, int nwg) {
    __shared__ float As[BM * LDA];   // [m][k]
    __shared__ float Bs[BK * LDB];   // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int per_batch = tiles_m * tiles_n;
    const int b = tile / per_batch, tb = tile % per_batch;
    const int m0 = (tb / tiles_n) * BM, n0 = (tb % tiles_n) * BN;
    const int wm = wave >> 1, wn = wave & 1;
    const float* Ab = A + (int64_t)b * M * K;
    const float* Bb = B + (int64_t)b * K * N;
    float* Cb = C + (int64_t)b * M * N;

    // staging: A tile 128 x 16 -> thread (k = tid & 15, rows (tid >> 4) + 16 u); B tile 16 x 128 -> thread
    // (n = tid & 127, k rows (tid >> 7) + 2 u): consecutive lanes read consecutive addresses.
    const int ak = tid & 15, ar = tid >> 4;
    const int bn = tid & 127, bk = tid >> 7;
    float ra[8], rb[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + ar + 16 * u, k = k0 + ak;
            ra[u] = (m < M && k < K) ? Ab[(int64_t)m * K + k] : 0.f;
            const int kk = k0 + bk + 2 * u, n = n0 + bn;
            rb[u] = (kk < K && n < N) ? Bb[(int64_t)kk * N + n] : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            As[(ar + 16 * u) * LDA + ak] = ra[u];
            Bs[(bk + 2 * u) * LDB + bn] = rb[u];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    gload(0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (t + 1 < nk) gload((t + 1) * BK);
        const float* ap = As + (wm * 64 + (lane & 31)) * LDA + (lane >> 5);
        const float* bp = Bs + (lane >> 5) * LDB + wn * 64 + (lane & 31);
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const float a0 = ap[kp * 2], a1 = ap[32 * LDA + kp * 2];
            const float b0 = bp[kp * 2 * LDB], b1 = bp[kp * 2 * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
        }
    }

    // D[row = n_local = (r&3) + 8 (r>>2) + 4 (lane>>5)][col = m_local = lane & 31]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
                float* cp = Cb + (int64_t)m * N + n;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) cp[e] = acc[i][j][g * 4 + e];
            }
        }
}

}  // namespace

int launch_bmm32_nn(const float* A, const float* B, float* C, int64_t batch, int M, int N, int K, hipStream_t s) {
    MST_CHECK_ARG(M > 0 && N > 0 && K > 0, "bmm32: M=%d N=%d K=%d", M, N, K);
    if (batch <= 0) return MST_OK;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = batch * tiles_m * tiles_n;
    MST_CHECK_ARG(nwg < (1ll << 31), "bmm32: grid too large (%lld work-groups)", (long long)nwg);
    bmm32_nn_kernel<<<dim3((unsigned)nwg), dim3(256), 0, s>>>(A, B, C, M, N, K, tiles_m, tiles_n, (int)nwg);
    return mst_check_launch("bmm32_nn");
}
