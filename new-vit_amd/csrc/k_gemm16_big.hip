// Large-tile persistent variant of the 16-bit MFMA GEMM (same contract as k_gemm16.hip), for the
// encoder's streaming shapes (M = slices x tokens ~ 10^5, N in {1152, 384, 1536}, K in {384, 1536}).
//
// Why: a CU's global->LDS fill rate (~55-90 GB/s per CU on MI355X, microarch guide "ring-gemm")
// bounds a 128x128 tile at ~64 FLOP/B to well under half the MFMA rate; a 256 x 384 tile needs
// 153 FLOP per staged byte, and 384 divides every N of the ViT (QKV 3x, proj/fc2 1x, fc1 4x).
//   * workgroup = 8 waves (2 x 4), two per SIMD so one wave's waits hide under the other's MFMAs
//     (a 4-wave / 512-register build of the same tile measured 25 % slower); wave tile 128 x 96 =
//     8 x 6 v_mfma_f32_16x16x32 (192 accumulator registers), 48 MFMAs per 14 ds_read_b128
//   * BK = 32, 4-deep LDS ring of 40 KiB stages (160 KiB = the whole LDS, one workgroup per CU),
//     filled by global_load_lds_dwordx4; three stages stay in flight across raw s_barriers behind
//     counted s_waitcnt vmcnt(10/5/0) (each wave issues exactly 5 LDS-DMA per stage)
//   * 64-byte LDS rows: slot = chunk ^ g[(row>>2)&3], g = {0,3,2,1} -> every ds_read_b128 lane
//     group touches 16 distinct 16-byte slots of one 256-byte bank row; the swizzle is applied on the
//     DMA source address and on the read (LDS image stays lane-linear)
//   * persistent: 256 workgroups walk the tiles; the next tile's first three stages are issued
//     before the current tile's epilogue, so the fill latency hides under the stores
//   * bias is folded into the accumulator init and interior tiles run a branch-free epilogue (a
//     per-lane `m < M` branch makes hipcc drain vmcnt(0) before every single store)
//   * tile ids are dealt so that the N-tiles of one M-panel run on one XCD (shared A in its L2)
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int BM = 256, BN = 384, BK = 32, NSTAGE = 4;
constexpr int A_BYTES = BM * BK * 2;              // 16 KiB
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 40 KiB
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;   // 160 KiB
constexpr int NI = 6, NJ = 8;                     // 16-wide sub-tiles per wave: N, M
constexpr int PA = 2, PW = 3, PS = PA + PW;       // LDS-DMA pieces per wave per stage: A, W, total

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <typename T, int EPI, typename OutT>
__global__ __launch_bounds__(512) void gemm16_big_kernel(const T* __restrict__ A, int64_t lda,
                                                         const T* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, OutT* C,
                                                         int64_t ldc, int M, int N, int K,
                                                         const float* __restrict__ gamma, float col_scale,
                                                         int scale_cols, int tiles_n, int ntiles) {
    typedef typename V8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / BK;

    // block -> tile schedule: XCD x (= bid & 7) owns ids [x*32, x*32+32) of every round of 256
    const int nblk = gridDim.x;
    const int bslot = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);  // nblk is a multiple of 8

    // per-lane staging geometry: LDS-DMA piece = 16 rows x 64 B; lane -> row lane>>2, slot lane&3
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);   // chunk whose home is slot lane&3 (g[(row>>2)&3])
    // fragment read geometry (16x16x32: lane -> row lane&15, k-chunk lane>>4)
    const int frow = lane & 15;
    const int fslot = ((lane >> 4) ^ ((0 - (frow >> 2)) & 3)) * 16;
    const int a_frag_off = (wm * 128 + frow) * 64 + fslot;
    const int w_frag_off = A_BYTES + (wn * 96 + frow) * 64 + fslot;

    // A: pieces PA*wave .. (16 rows each); W: pieces PW*wave ..
    const T* a_src[PA];
    const T* w_src;   // piece u of this wave is at w_src + u*16*ldw
    auto set_tile = [&](int tile) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            int r = m0 + (wave * PA + u) * 16 + srow;
            r = r < M ? r : M - 1;
            a_src[u] = A + (int64_t)r * lda + schunk * 8;
        }
        w_src = W + (int64_t)(n0 + wave * (PW * 16) + srow) * ldw + schunk * 8;
    };
    auto stage = [&](int buf, int kt) {   // PS LDS-DMA per wave
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < PA; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[u] + kt * BK), LDS_PTR(base + (wave * PA + u) * 1024), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < PW; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_src + (int64_t)u * 16 * ldw + kt * BK),
                                             LDS_PTR(base + A_BYTES + (wave * PW + u) * 1024), 16, 0, 0);
    };

    int tile = bslot;
    if (tile >= ntiles) return;
    set_tile(tile);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s);

    // NB: every wave must issue exactly PS LDS-DMA per stage and nothing else may sit between a stage and
    // the counted waits of the K-loop except older memory operations (vmcnt retires in order).
    while (true) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const int nb = n0 + wn * 96 + (lane >> 4) * 4;
        const int mb = m0 + wm * 128 + (lane & 15);
        // accumulators start at the bias (its loads retire here, behind the previous tile's stores)
        f32x4 acc[NI][NJ];
        {
            f32x4 b0[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {   // all loads first: one wait, not one per column group
                b0[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (bias) b0[i] = *reinterpret_cast<const f32x4*>(bias + nb + i * 16);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = b0[i];
        }

        for (int t = 0; t < nk; ++t) {
            const int rem = nk - 1 - t;   // stages issued after stage t
            if (rem >= 2) wait_vm_barrier<2 * PS>();
            else if (rem == 1) wait_vm_barrier<PS>();
            else wait_vm_barrier<0>();
#ifndef ABL_NOLOAD
            if (t + NSTAGE - 1 < nk) stage((t + NSTAGE - 1) & (NSTAGE - 1), t + NSTAGE - 1);
#endif
            const char* sb = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
            vec8 wf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const vec8*>(sb + w_frag_off + i * 16 * 64);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const vec8 af = *reinterpret_cast<const vec8*>(sb + a_frag_off + j * 16 * 64);
#ifndef ABL_NOMFMA
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i][j] = mfma16(wf[i], af, acc[i][j]);
#else
                asm volatile("" ::"v"(af));
#endif
            }
#ifdef ABL_NOMFMA
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wf[i]));
#endif
        }

        // Every wave is past its last LDS read of this tile once it arrives here.  The next tile's first
        // two stages are issued now (ring slots 0,1); slots 2,3 become the epilogue's staging area and the
        // third stage follows after the epilogue (behind a barrier: its DMA lands in the staging area).
        const int next = tile + nblk;
        asm volatile("s_barrier" ::: "memory");
        if (next < ntiles) {
            set_tile(next);
            if (0 < nk) stage(0, 0);
            if (1 < nk) stage(1, 1);
        }

        // ---- epilogue, coalesced through LDS.  In registers a lane owns C[mb+16j][nb+16i .. +3]: a direct
        // store touches 16 rows x 32-64 B per instruction (measured: 2 TB/s, 2x the whole K-loop).  Instead
        // each wave transposes one 16-row x 96-column slab at a time through its private staging area and
        // moves it to / from global memory as 16-byte lane accesses along whole row segments.
        {
            constexpr int OB = (int)sizeof(OutT);
            constexpr int ROWB = 96 * OB + 16;            // padded staging row (400 B fp32 / 208 B 16-bit)
            constexpr int CPR = 96 * OB / 16;             // 16-byte chunks per row (24 / 12)
            constexpr int NCH = 16 * CPR / 64;            // chunk instructions per slab (6 / 3)
            char* stg = smem + 2 * STAGE_BYTES + wave * (16 * ROWB);
            const int wr_off = (lane & 15) * ROWB + (lane >> 4) * 4 * OB;
            const int n_w = n0 + wn * 96;                 // first column of this wave
            const int m_w = m0 + wm * 128;                // first row of this wave
            auto slabs = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;   // interior tile: no per-lane row guards
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (nb + i * 16 + r < scale_cols) v[r] *= col_scale;
                            if (EPI == MST_EPI_BIAS_GELU) v[r] = (OB == 2) ? gelu_fast(v[r]) : gelu_erf(v[r]);
                            if (EPI == MST_EPI_BIAS_RELU) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (OB == 4) {
                            *reinterpret_cast<float4*>(stg + wr_off + i * 64) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            typedef __attribute__((ext_vector_type(4))) OutT o4;
                            o4 pk;
                            pk[0] = (OutT)v[0];
                            pk[1] = (OutT)v[1];
                            pk[2] = (OutT)v[2];
                            pk[3] = (OutT)v[3];
                            *reinterpret_cast<o4*>(stg + wr_off + i * 32) = pk;
                        }
                    }
                    float4 xv[NCH];
                    if constexpr (EPI == MST_EPI_RESIDUAL) {   // all read-modify-write loads of the slab first
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int q = c * 64 + lane;
                            const int row = q / CPR, ch = q - row * CPR;
                            const int m = m_w + j * 16 + row;
                            if (FULL || m < M) xv[c] = *reinterpret_cast<const float4*>(C + (int64_t)m * ldc + n_w + ch * 4);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int q = c * 64 + lane;
                        const int row = q / CPR, ch = q - row * CPR;
                        const int m = m_w + j * 16 + row;
                        const u32x4 t = *reinterpret_cast<const u32x4*>(stg + row * ROWB + ch * 16);
#ifdef ABL_NOSTORE
                        asm volatile("" ::"v"(t));
                        continue;
#endif
                        if (!FULL && m >= M) continue;
                        OutT* cp = C + (int64_t)m * ldc + n_w + ch * (16 / OB);
                        if constexpr (EPI == MST_EPI_RESIDUAL) {
                            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
                            if (gamma) gv = *reinterpret_cast<const float4*>(gamma + n_w + ch * 4);
                            float4 o;
                            o.x = xv[c].x + gv.x * __uint_as_float(t[0]);
                            o.y = xv[c].y + gv.y * __uint_as_float(t[1]);
                            o.z = xv[c].z + gv.z * __uint_as_float(t[2]);
                            o.w = xv[c].w + gv.w * __uint_as_float(t[3]);
                            *reinterpret_cast<float4*>(cp) = o;
                        } else {
                            *reinterpret_cast<u32x4*>(cp) = t;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep each slab's loads/temps from being hoisted over the others
                }
            };
            if (m0 + BM <= M) slabs(std::true_type{});
            else slabs(std::false_type{});
        }
        if (next < ntiles) {
            asm volatile("s_barrier" ::: "memory");   // staging area is free again on every wave
            if (2 < nk) stage(2, 2);
        }

        if (next >= ntiles) break;
        tile = next;
    }
}

template <typename T, int EPI, typename OutT>
int launch_t(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
             int64_t M, int N, int K, const float* gamma, float cs, int sc, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = gemm16_big_kernel<T, EPI, OutT>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = N / BN;
    const int ntiles = tiles_m * tiles_n;
    const int cus = mst_persistent_grid();
    int nblk = ntiles < cus ? ((ntiles + 7) / 8) * 8 : cus;
    kern<<<dim3(nblk), dim3(512), LDS_BYTES, s>>>((const T*)A, lda, (const T*)W, ldw, bias, (OutT*)C, ldc, (int)M, N, K,
                                                   gamma, cs, sc, tiles_n, ntiles);
    return mst_check_launch("gemm16_big");
}

template <typename T>
int dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int cdt, int64_t ldc,
             int64_t M, int N, int K, int epi, const float* gamma, float cs, int sc, hipStream_t s) {
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
    mst_set_error("gemm16_big: bad epilogue %d", epi);
    return MST_EINVAL;
}

}  // namespace

// true when the big-tile kernel applies and fills the chip (else the 128x128 kernel is used)
bool gemm16_big_applicable(int64_t M, int N, int K) {
    return (N % BN == 0) && (K % BK == 0) && ((M + BM - 1) / BM) * (int64_t)(N / BN) >= 192;
}

int launch_gemm16_big(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                      int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma, float col_scale,
                      int scale_cols, hipStream_t s) {
    if (dt == MST_BF16) return dispatch<bf16_t>(A, lda, W, ldw, bias, C, cdt, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    if (dt == MST_F16) return dispatch<f16_t>(A, lda, W, ldw, bias, C, cdt, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    mst_set_error("gemm16_big: bad operand dtype %d", dt);
    return MST_EINVAL;
}
