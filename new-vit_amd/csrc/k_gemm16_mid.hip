// Mid-tile 16-bit MFMA GEMM (contract of k_gemm16.hip; 16-bit outputs with bias / ReLU epilogues, fp32 output with the residual epilogue) for the QKV projection
// (M ~ 10^5, N = 1152, K = 384): tile 128 x 384, ONE 4-wave workgroup per CU, one wave per SIMD.
//
// Why: with K = 384 the persistent 256 x 384 kernel (k_gemm16_big.hip, 8 waves at 256 VGPRs) and the 128 x 128 kernel
// (64 FLOP per staged byte, fill-bound) both took 0.65 ms on this shape.  Four waves with the whole register file
// (the compiler uses ~440 VGPRs: every fragment of a k-step and the epilogue's temporaries stay in registers, no
// spills) measured 0.58 ms; the same tile capped at 256 VGPRs so that two workgroups fit a CU did not (0.66 ms).
//   * wave tile 128 x 96 = 8 x 6 MFMA 16x16x32 (192 accumulator registers), 48 MFMAs per 14 ds_read_b128;
//   * BK = 32, ring of 32 KiB stages filled by global_load_lds_dwordx4 (8 pieces per wave and stage); stage t+NSTAGE-1 is
//     issued right behind the barrier of step t;
//   * 64-byte LDS rows with the big kernel's swizzle (slot = chunk ^ g[(row>>2)&3]) applied on the DMA source address;
//   * persistent over a strided tile list (XCD-grouped); the next tile's first stages are issued before the epilogue;
//   * epilogue through a private 16 x 96 staging slab per wave (outside the ring), bias folded into the accumulator init.
#include <type_traits>

#include "mst_common.h"

namespace {

#ifndef MID_BK
#define MID_BK 32         // 64: whole 128-byte lines per row and stage (8-row LDS-DMA pieces, 128-byte LDS rows, slot = chunk ^ ((row>>1)&7))
#endif
constexpr int BM = 128, BN = 384, BK = MID_BK;
constexpr int A_BYTES = BM * BK * 2;              // 8 KiB
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
#ifndef MID_NSTAGE
#define MID_NSTAGE 2      // 2 and 4 stages measured the same (0.58 ms on the QKV shape): the fill latency is not the limit
#endif
constexpr int NSTAGE = MID_NSTAGE;
constexpr int STG_OFF = NSTAGE * STAGE_BYTES;     // epilogue staging behind the ring
constexpr int LDS_BYTES = STG_OFF + 4 * 16 * (96 * 4 + 16);   // ring + staging (25.6 KiB covers fp32 rows)
constexpr int NI = 6, NJ = 8;                     // 16-wide sub-tiles per wave: N, M
constexpr int PA = BK / 16, PW = 3 * BK / 16, PS = PA + PW;   // LDS-DMA pieces (1 KiB) per wave per stage: A, W, total
constexpr int RPP = 1024 / (BK * 2);              // rows per piece (16 / 8)
constexpr int ROWB_K = BK * 2;                    // LDS row bytes (64 / 128)
#ifndef MID_NBLK
#define MID_NBLK 0        // 0 = one workgroup per compute unit of the device
#endif
constexpr int NBLK = MID_NBLK;

template <typename T, int EPI, typename OutT>
__global__ __launch_bounds__(256) void gemm16_mid_kernel(const T* __restrict__ A, int64_t lda,
                                                         const T* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, OutT* C, int64_t ldc, int M,
                                                         int N, int K, const float* __restrict__ gamma, float col_scale,
                                                         int scale_cols, int tiles_n, int ntiles) {
    typedef typename V8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = column quarter of the tile
    const int nk = K / BK;
    const int nblk = gridDim.x;
    // XCD x (= bid & 7) owns a contiguous eighth of every round of tile ids (n fastest): the N-tiles of an M-panel share
    // one L2
    const int bslot = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);   // nblk is a multiple of 8

    const int frow = lane & 15;
    int srow, schunk, schunk_odd, fslot;
    if constexpr (BK == 32) {
        srow = lane >> 2;
        schunk = schunk_odd = (lane & 3) ^ ((0 - (lane >> 4)) & 3);   // chunk whose home is slot lane&3
        fslot = ((lane >> 4) ^ ((0 - (frow >> 2)) & 3)) * 16;
    } else {
        srow = lane >> 3;                                 // pieces of 8 rows: piece u holds LDS rows 8u + srow, (row>>1)&7 = 4(u&1) + (srow>>1)
        schunk = (lane & 7) ^ (srow >> 1);
        schunk_odd = schunk ^ 4;
        fslot = ((lane >> 4) ^ (frow >> 1)) * 16;         // k-half h: chunk (lane>>4) + 4h -> byte offset ^ 64
    }
    const int a_frag_off = frow * ROWB_K + fslot;
    const int w_frag_off = A_BYTES + (wave * 96 + frow) * ROWB_K + fslot;

    const T* a_src[PA];
    const T* w_src;   // piece u of this wave is at w_src + u*RPP*ldw (+- the chunk swap of odd pieces)
    const int odd_fix = (schunk_odd - schunk) * 8;
    auto set_tile = [&](int tile) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            int r = m0 + (wave * PA + u) * RPP + srow;
            r = r < M ? r : M - 1;
            a_src[u] = A + (int64_t)r * lda + ((u & 1) ? schunk_odd : schunk) * 8;
        }
        w_src = W + (int64_t)(n0 + wave * (PW * RPP) + srow) * ldw + schunk * 8;
    };
    auto stage = [&](int buf, int kt) {   // PS LDS-DMA per wave
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < PA; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[u] + kt * BK), LDS_PTR(base + (wave * PA + u) * 1024), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < PW; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_src + (int64_t)u * RPP * ldw + kt * BK + ((u & 1) ? odd_fix : 0)),
                                             LDS_PTR(base + A_BYTES + (wave * PW + u) * 1024), 16, 0, 0);
    };

    int tile = bslot;
    if (tile >= ntiles) return;
    set_tile(tile);
    int ring = 0;                                     // ring slot of this tile's stage 0
#pragma unroll
    for (int u = 0; u < NSTAGE - 1; ++u)
        if (u < nk) stage(u, u);

    while (true) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const int nb = n0 + wave * 96 + (lane >> 4) * 4;
        f32x4 acc[NI][NJ];
        {
            f32x4 b0[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                b0[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (bias) b0[i] = *reinterpret_cast<const f32x4*>(bias + nb + i * 16);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = b0[i];
        }

        for (int t = 0; t < nk; ++t) {
            // stage t has landed: only the stages issued after it (at most NSTAGE-2) may still be in flight; behind the
            // barrier every wave is done reading the slot of step t-1, which stage t+NSTAGE-1 now overwrites
            const int rem = nk - 1 - t;
            if (NSTAGE >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PS) : "memory");
            else if (NSTAGE >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (t + NSTAGE - 1 < nk) stage((ring + t + NSTAGE - 1) % NSTAGE, t + NSTAGE - 1);
            const char* sb = smem + ((ring + t) % NSTAGE) * STAGE_BYTES;
#pragma unroll
            for (int h = 0; h < BK / 32; ++h) {
            vec8 wf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const vec8*>(sb + ((w_frag_off + i * 16 * ROWB_K) ^ (h * 64)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const vec8 af = *reinterpret_cast<const vec8*>(sb + ((a_frag_off + j * 16 * ROWB_K) ^ (h * 64)));
#pragma unroll
                for (int i = 0; i < NI; ++i) {
#ifdef MID_ABL_NO_MFMA
                    asm volatile("" : "+v"(acc[i][j]) : "v"(wf[i]), "v"(af));
#else
                    acc[i][j] = mfma16(wf[i], af, acc[i][j]);
#endif
                }
            }
            }
        }

        // the next tile's first NSTAGE-1 stages go into the slots after this tile's last one: none of them is the slot of
        // step nk-1 (possibly still being read by a slower wave); all others were released by the barrier of step nk-1
        const int next = tile + nblk;
        ring = (ring + nk) % NSTAGE;
        if (next < ntiles) {
            set_tile(next);
#pragma unroll
            for (int u = 0; u < NSTAGE - 1; ++u)
                if (u < nk) stage((ring + u) % NSTAGE, u);
        }

#ifdef MID_ABL_NO_STORE
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
        if (next >= ntiles) break;
        tile = next;
        continue;
#endif
#ifdef MID_ABL_DIRECT                                  // 8-byte lane pieces straight from the accumulators (32 B per token per instruction)
        if constexpr (sizeof(OutT) == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = m0 + j * 16 + (lane & 15);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    typedef __attribute__((ext_vector_type(4))) OutT o4;
                    o4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r];
                        if (nb + i * 16 + r < scale_cols) v *= col_scale;
                        pk[r] = (OutT)v;
                    }
                    if (m < M) *reinterpret_cast<o4*>(C + (int64_t)m * ldc + nb + i * 16) = pk;
                }
            }
            if (next >= ntiles) break;
            tile = next;
            continue;
        }
#endif
        // ---- epilogue: each wave transposes one 16-row x 96-column slab at a time through its private staging area and
        // moves it to / from global memory as 16-byte lane accesses along whole row segments (fp32: 384 B = three lines)
        {
            constexpr int OB = (int)sizeof(OutT);
            constexpr int ROWB = 96 * OB + 16;            // padded staging row (400 B fp32 / 208 B 16-bit)
            constexpr int CPR = 96 * OB / 16;             // 16-byte chunks per row (24 / 12)
            constexpr int NCH = 16 * CPR / 64;            // chunk instructions per slab (6 / 3)
            char* stg = smem + STG_OFF + wave * (16 * ROWB);
            const int wr_off = (lane & 15) * ROWB + (lane >> 4) * 4 * OB;
            const int n_w = n0 + wave * 96;
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
                            const int m = m0 + j * 16 + row;
                            if (FULL || m < M) xv[c] = *reinterpret_cast<const float4*>(C + (int64_t)m * ldc + n_w + ch * 4);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int q = c * 64 + lane;
                        const int row = q / CPR, ch = q - row * CPR;
                        const int m = m0 + j * 16 + row;
                        const u32x4 tv = *reinterpret_cast<const u32x4*>(stg + row * ROWB + ch * 16);
                        if (!FULL && m >= M) continue;
                        OutT* cp = C + (int64_t)m * ldc + n_w + ch * (16 / OB);
                        if constexpr (EPI == MST_EPI_RESIDUAL) {
                            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
                            if (gamma) gv = *reinterpret_cast<const float4*>(gamma + n_w + ch * 4);
                            float4 o;
                            o.x = xv[c].x + gv.x * __uint_as_float(tv[0]);
                            o.y = xv[c].y + gv.y * __uint_as_float(tv[1]);
                            o.z = xv[c].z + gv.z * __uint_as_float(tv[2]);
                            o.w = xv[c].w + gv.w * __uint_as_float(tv[3]);
                            *reinterpret_cast<float4*>(cp) = o;
                        } else {
                            *reinterpret_cast<u32x4*>(cp) = tv;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (m0 + BM <= M) slabs(std::true_type{});
            else slabs(std::false_type{});
        }
        if (next >= ntiles) break;
        tile = next;
        // the K-loop's first barrier orders this tile's last ring reads (all before the epilogue) against nothing: the
        // slot stage 1 will overwrite was last read at step nk-2 or nk-1, both behind barriers every wave has passed
    }
}

template <typename T, int EPI, typename OutT>
int launch_t(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
             int N, int K, const float* gamma, float cs, int sc, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = gemm16_mid_kernel<T, EPI, OutT>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = N / BN;
    const int ntiles = tiles_m * tiles_n;
    const int cus = NBLK > 0 ? NBLK : mst_persistent_grid();
    int nblk = ntiles < cus ? ((ntiles + 7) / 8) * 8 : cus;
    kern<<<dim3(nblk), dim3(256), LDS_BYTES, s>>>((const T*)A, lda, (const T*)W, ldw, bias, (OutT*)C, ldc, (int)M, N, K, gamma,
                                                   cs, sc, tiles_n, ntiles);
    return mst_check_launch("gemm16_mid");
}

template <typename T>
int dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
             int N, int K, int epi, const float* gamma, float cs, int sc, hipStream_t s) {
    switch (epi) {
        case MST_EPI_BIAS: return launch_t<T, MST_EPI_BIAS, T>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
        case MST_EPI_BIAS_RELU: return launch_t<T, MST_EPI_BIAS_RELU, T>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
        case MST_EPI_RESIDUAL: return launch_t<T, MST_EPI_RESIDUAL, float>(A, lda, W, ldw, bias, C, ldc, M, N, K, gamma, cs, sc, s);
    }
    mst_set_error("gemm16_mid: bad epilogue %d", epi);
    return MST_EINVAL;
}

}  // namespace

// true when the two-per-CU kernel applies: 16-bit output of the operand type, shallow K (the epilogue weighs as much as
// the K-loop), enough tiles to fill the chip twice
bool gemm16_mid_applicable(int64_t M, int N, int K, int dt, int cdt, int epi) {
    // the residual (fp32 read-modify-write) epilogue is implemented too, but the HBM-bound proj GEMM measured the same on
    // this kernel and on the 128 x 128 one (0.33 ms), so it stays there
    const bool out_ok = dt == cdt && (epi == MST_EPI_BIAS || epi == MST_EPI_BIAS_RELU);
    return out_ok && (N % BN == 0) && (K % BK == 0) && K <= 512 &&
           ((M + BM - 1) / BM) * (int64_t)(N / BN) >= 512;
}

int launch_gemm16_mid(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                      int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma, float col_scale, int scale_cols,
                      hipStream_t s) {
    if (dt == MST_BF16) return dispatch<bf16_t>(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    if (dt == MST_F16) return dispatch<f16_t>(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gamma, col_scale, scale_cols, s);
    mst_set_error("gemm16_mid: bad operand dtype %d", dt);
    return MST_EINVAL;
}
