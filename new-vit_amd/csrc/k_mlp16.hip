// Fused MLP block of a ViT layer for E = 384, 16-bit MFMA operands:
//     x <- x + ls2 * ( fc2( gelu( fc1( LayerNorm2(x) ) ) ) )          block.py:93-94,113; mlp.py:34-40
//     xn <- normalise(x)   (optional: the NEXT block's norm1, its affine folded into that block's QKV weights)
// in ONE kernel.  The 4E-wide hidden activations never leave the CU, LayerNorm needs no pass of its own, and per
// token the kernel moves ~3x1536 B in + 1536 B (+768 B) out instead of ~12 KB for LN + fc1 + fc2 as three launches
// (the unfused layer is HBM-bound: DESIGN.md section 5).
//
// Structure: persistent, one 8-wave workgroup per CU, 128 token rows per tile, PRODUCER / CONSUMER wave roles
// (waves w and w+4 share a SIMD, so each SIMD hosts one of each and their instruction mixes complement):
//   * producer k (waves 0-3): LayerNorm of rows [32k, 32k+32) kept as GEMM1 B-operand fragments in registers (the
//     LN affine is folded into W1 / b1 on the host); per 32-wide hidden chunk 48 MFMAs (H^T = W1c . xhat^T),
//     GELU on 16 values per lane, and the accumulators -- converted in place, they ARE GEMM2's B operand -- go to
//     a 2 KB LDS slot at the lane's own index;
//   * consumer k (waves 4-7): one chunk behind, 48 MFMAs per chunk (y^T[384 x 32] += W2c . H^T, 192 accumulator
//     registers), all LDS-DMA issue, and per tile the epilogue x += y + b2 (LayerScale folded into W2 / b2 on the host) plus the next LayerNorm straight
//     from the registers -- while the producers already normalise the next tile;
//   * every weight fragment read from LDS feeds two MFMAs; reads are hand-pipelined DEPTH ahead behind counted
//     lgkmcnt waits (hipcc otherwise emits read -> wait(0) -> 2 MFMAs);
//   * weights stream through two 3-deep LDS rings (W1 rows / W2 columns of a chunk, 24 KB each), pre-packed on the
//     host as the exact swizzled LDS image (1 KiB contiguous per LDS-DMA piece); the rings run across tile
//     boundaries; one step's DMA group stays in flight behind `s_waitcnt vmcnt(12)`; ONE `s_barrier` per step orders
//     ring slots and the H hand-off;
//   * W2's columns are permuted on the host to the accumulator's k order and its rows so that a lane ends up with
//     8-column groups (128-B row segments across the 4 lane groups) for the residual read-modify-write.
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int E = 384, HID = 1536, CH = 32;            // CH hidden units per chunk
constexpr int NCHUNK = HID / CH;                        // 48
constexpr int W1_BYTES = CH * E * 2;                    // 24 KiB
constexpr int W2_BYTES = E * CH * 2;                    // 24 KiB
constexpr int CHUNK_BYTES = W1_BYTES + W2_BYTES;        // packed weights per chunk in global memory
constexpr int NSLOT = 3;
constexpr int W2_RING = NSLOT * W1_BYTES;               // byte offset of the W2 ring
constexpr int HBUF = 2 * NSLOT * W1_BYTES;              // byte offset of the H hand-off buffer (2 x 8 KiB)
constexpr int LDS_BYTES = HBUF + 2 * 8192;              // 163,840 = all of the LDS
constexpr int KS = E / 32;                              // 12 k-steps of GEMM1
constexpr int NT = E / 16;                              // 24 output tiles (16 columns) of GEMM2
constexpr int PC_CONS = 12;                             // LDS-DMA pieces per step issued by each consumer wave (of 12 per wave pair)
constexpr int PC_PROD = 12 - PC_CONS;
constexpr int DEPTH = 4;                                // fragment reads in flight ahead of the MFMAs
#ifndef TILE_CYCLES
#define TILE_CYCLES 150000                              // ~one 128-row tile (48 steps) in shader clocks: the de-phasing window
#endif

template <int OFF, typename V> __device__ __forceinline__ void lds_read_b128(V& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {       // + fence: no MFMA above the wait (rule 18)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifdef MLP_STAMPS
// diagnostic build only (never shipped): per-wave cycle sums of the step phases, read back by mst_debug_mlp_stamps
__device__ unsigned long long g_stamps[256 * 8 * 4];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp()
#define ACCUM(slot, a, b) st[slot] += (b) - (a)
#else
#define STAMP(var)
#define ACCUM(slot, a, b)
#endif

template <typename T>
__global__ __launch_bounds__(512) void mlp16_kernel(float* __restrict__ x, T* __restrict__ xn_out,
                                                    const char* __restrict__ wpack,
                                                    const float* __restrict__ b1f, const float* __restrict__ b2,
                                                    const float* __restrict__ ls2, int M, int ntiles, float eps) {
    typedef typename V8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    const int pr = wave & 3;                             // rows [32*pr, 32*pr + 32) of the tile
    const int frow = lane & 15, g = lane >> 4;
    const int fslot = (g ^ ((0 - (frow >> 2)) & 3)) * 16;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int frag_off = frow * 64 + fslot;              // + 1024*tile16 (+ ks*2048 in the W1 image)
    char* const hslot = smem + HBUF + pr * 2048 + lane * 16;       // + mt*1024 + parity*8192

    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int last = my_tiles * NCHUNK;                  // steps 0 .. last+1: GEMM1 of chunk s in [0,last), its GELU + hand-off at s+1, GEMM2 at s+2

    // DMA group of step s: everything step s+2 needs = W1 image of producer chunk (s+2)%48 -> W1 slot (s+2)%3 and
    // W2 image of the chunk consumers see at step s+2, i.e. s%48 -> W2 slot (s+2)%3.  Issued by consumers only.
    // Branch-free: piece u of consumer k is image piece 4u+k (u < 6: W1, else W2); groups past either end of the
    // wave's work load a valid but unused chunk into a slot nobody reads again (cheaper than 24 scalar branches).
    const char* const wp_lane = wpack + pr * 1024 + lane * 16;
    char* const ring_wave = smem + pr * 1024;
    struct DmaGroup { const char* s1; const char* s2; char* d1; char* d2; };
    auto dma_addr = [&](int s) {
        const int sp = s + 2;
        const int slot = (sp + NSLOT) % NSLOT;
        DmaGroup a;
        a.s1 = wp_lane + (size_t)((sp + NCHUNK) % NCHUNK) * CHUNK_BYTES;
        a.s2 = wp_lane + (size_t)((sp + NCHUNK - 2) % NCHUNK) * CHUNK_BYTES + W1_BYTES;
        a.d1 = ring_wave + slot * W1_BYTES;
        a.d2 = ring_wave + W2_RING + slot * W2_BYTES;
        return a;
    };
    auto dma_piece = [&](const DmaGroup& a, int u) {     // image piece 4u + (wave & 3), u = 0..11 (compile-time): u < 6 W1, else W2
        if (u < 6) __builtin_amdgcn_global_load_lds(GLB_PTR(a.s1 + u * 4096), LDS_PTR(a.d1 + u * 4096), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(GLB_PTR(a.s2 + (u - 6) * 4096), LDS_PTR(a.d2 + (u - 6) * 4096), 16, 0, 0);
    };
    // The issue cost of a piece (~85 cycles inside these loops) is paid by the issuing wave only, so the 48 pieces of a
    // step are split to level the two roles' step times: consumers take PC_CONS each, producers the rest.
    auto dma_group = [&](int s) {
        const DmaGroup a = dma_addr(s);
        if (producer) {
#pragma unroll
            for (int u = PC_CONS; u < 12; ++u) dma_piece(a, u);
        } else {
#pragma unroll
            for (int u = 0; u < PC_CONS; ++u) dma_piece(a, u);
        }
    };
    dma_group(-2);
    dma_group(-1);

    // De-phase the persistent workgroups.  All of them walk the same 48-step tiles, so without this every CU reads its
    // LayerNorm rows and writes its epilogue at the same moment: HBM sees bursts at its full rate while the MFMAs
    // idle, then nothing for 48 steps.  A one-off start delay spreads the tile boundaries over the tile period;
    // the workgroups that own one tile fewer take the later half of the phases, so the tail stays balanced.
    if (my_tiles >= 4) {
        const int min_tiles = ntiles / (int)gridDim.x;
        const unsigned u = (((unsigned)blockIdx.x >> 3) + 5u * ((unsigned)blockIdx.x & 7u)) & 15u;   // 0..15
        const unsigned long long delay = (unsigned long long)TILE_CYCLES * ((my_tiles > min_tiles ? 0u : 16u) + u) / 32u;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 512 && __builtin_readcyclecounter() - t0 < delay; ++it) __builtin_amdgcn_s_sleep(32);   // bounded: every wave leaves
    }

    // The two roles run two separate step loops (same number of s_barriers: one per step 0..last), so the
    // register allocator sees the producer's fragments and the consumer's accumulators in disjoint live ranges.
    // barrier(s): the DMA group of step s-2 has landed (consumers waited for theirs) and every H(s-1) is written.
#define ACC(t, mt) R[2 * (t) + (mt)]
#define XA(mt, ks) R[KS * (mt) + (ks)]
    if (producer) {
        u32x4 R[2 * KS];                                 // xa[mt][ks]: normalised rows, GEMM1 B operand
        f32x4 bias_next[2];
        bias_next[0] = *reinterpret_cast<const f32x4*>(b1f + 4 * g);
        bias_next[1] = *reinterpret_cast<const f32x4*>(b1f + 16 + 4 * g);
#ifdef MLP_STAMPS
        unsigned long long st[4] = {0, 0, 0, 0};
#endif
        f32x4 hp[2][2] = {};                             // pre-activations of the previous chunk [hidden tile][row tile]
        auto store_h = [&](const float (&gv)[16], int sc) {   // k order of GEMM2 = (hidden tile, register) order of GEMM1
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                vec8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (T)gv[8 * mt + j];
                *reinterpret_cast<vec8*>(hslot + mt * 1024 + (sc & 1) * 8192) = hv;
            }
        };
#pragma unroll 1
        for (int s = 0; s <= last + 1; ++s) {
            STAMP(t0);
            // this wave's pieces of group s-2 have landed; group s-1 (the newest PC_PROD VMEM ops) may stay in flight
            if constexpr (PC_PROD > 0) {
                if (s <= last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PC_PROD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // H(s-2) writes drained before the hand-off
            STAMP(t1);
            ACCUM(0, t0, t1);
            if (s < last) {
                const int c = s % NCHUNK;
                if (c == 0) {
                    // ---- LayerNorm of this pair's 32 rows of tile s/48 -> GEMM1 B-operand fragments
                    const int tile = blockIdx.x + (s / NCHUNK) * gridDim.x;
                    float4 raw[2][2 * KS];               // both row tiles' 48 loads in flight together (one HBM round trip)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int m = tile * 128 + pr * 32 + 16 * mt + frow;
                        const float* xr = x + (size_t)(m < M ? m : M - 1) * E + 8 * g;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            raw[mt][2 * ks] = *reinterpret_cast<const float4*>(xr + 32 * ks);
                            raw[mt][2 * ks + 1] = *reinterpret_cast<const float4*>(xr + 32 * ks + 4);
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float v[KS * 8];
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const float4 a = raw[mt][2 * ks], b = raw[mt][2 * ks + 1];
                            v[8 * ks + 0] = a.x; v[8 * ks + 1] = a.y; v[8 * ks + 2] = a.z; v[8 * ks + 3] = a.w;
                            v[8 * ks + 4] = b.x; v[8 * ks + 5] = b.y; v[8 * ks + 6] = b.z; v[8 * ks + 7] = b.w;
                        }
                        float sum = 0.f;
#pragma unroll
                        for (int i = 0; i < KS * 8; ++i) sum += v[i];
                        sum += __shfl_xor(sum, 16, 64);
                        sum += __shfl_xor(sum, 32, 64);
                        const float mean = sum * (1.0f / E);
                        float sq = 0.f;
#pragma unroll
                        for (int i = 0; i < KS * 8; ++i) { const float d = v[i] - mean; sq = fmaf(d, d, sq); }
                        sq += __shfl_xor(sq, 16, 64);
                        sq += __shfl_xor(sq, 32, 64);
                        const float rstd = rsqrtf(sq * (1.0f / E) + eps);
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                        {
                            vec8 t8;
#pragma unroll
                            for (int j = 0; j < 8; ++j) t8[j] = (T)((v[8 * ks + j] - mean) * rstd);
                            XA(mt, ks) = __builtin_bit_cast(u32x4, t8);
                        }
                    }
                }
                STAMP(t2);
                ACCUM(1, t1, t2);                        // LayerNorm prologue (c == 0 only)
                // ---- GEMM1 of chunk c: 32 hidden units x 32 rows (accumulators start at b1), GELU, hand-off
                const f32x4 b0 = bias_next[0], b1v = bias_next[1];
                {
                    const int cn = (c + 1) % NCHUNK;     // b1 of the next step's chunk, one step ahead
                    bias_next[0] = *reinterpret_cast<const f32x4*>(b1f + cn * CH + 4 * g);
                    bias_next[1] = *reinterpret_cast<const f32x4*>(b1f + cn * CH + 16 + 4 * g);
                }
                dma_group(s);                            // after this step's other loads: they stay older than the pieces
                f32x4 h[2][2] = {{b0, b0}, {b1v, b1v}};  // [hidden tile][row tile]
                float gv[16];                            // GELU of the PREVIOUS chunk, interleaved with this chunk's MFMAs
                const unsigned w1a = lds_base + (s % NSLOT) * W1_BYTES + frag_off;
                vec8 w[2 * KS];                          // fragment index q = 2*ks + ht  ->  byte offset ks*2048 + ht*1024
                wait_lgkm<0>();
                static_for<0, DEPTH>([&](auto i) { constexpr int q = decltype(i)::value; lds_read_b128<(q >> 1) * 2048 + (q & 1) * 1024>(w[q], w1a); });
                static_for<0, 2 * KS>([&](auto i) {
                    constexpr int q = decltype(i)::value;
                    if constexpr (q + DEPTH < 2 * KS) lds_read_b128<((q + DEPTH) >> 1) * 2048 + ((q + DEPTH) & 1) * 1024>(w[q + DEPTH], w1a);
                    wait_lgkm<(2 * KS - 1 - q < DEPTH) ? (2 * KS - 1 - q) : DEPTH>();
                    h[q & 1][0] = mfma16(w[q], __builtin_bit_cast(vec8, XA(0, q >> 1)), h[q & 1][0]);
                    h[q & 1][1] = mfma16(w[q], __builtin_bit_cast(vec8, XA(1, q >> 1)), h[q & 1][1]);
                    if constexpr (q % 3 != 2) {          // 16 of the 24 iterations carry one GELU each (VALU under the MFMAs;
                        constexpr int e = q - q / 3;     // packed v_pk_fma_f32 pairs measured 4 % slower than scalar FMAs here)
                        gv[e] = gelu_sig<T>(hp[(e >> 2) & 1][e >> 3][e & 3]);
                    }
                });
                STAMP(t3);
                ACCUM(2, t2, t3);                        // GEMM1 (+ GELU of the previous chunk)
                store_h(gv, s - 1);                      // H(s-1); at s == 0 a dummy into the parity nobody reads yet
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) hp[i][j] = h[i][j];
                STAMP(t4);
                ACCUM(3, t3, t4);                        // hand-off
            } else if (s == last) {                      // drain: GELU + hand-off of the last chunk
                float gv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) gv[e] = gelu_sig<T>(hp[(e >> 2) & 1][e >> 3][e & 3]);
                store_h(gv, s - 1);
            }
        }
#ifdef MLP_STAMPS
        if (lane == 0) for (int i = 0; i < 4; ++i) g_stamps[(blockIdx.x * 8 + wave) * 4 + i] = st[i];
#endif
    } else {
        u32x4 R[2 * NT];                                 // acc[t][mt]: y^T accumulators
#ifdef MLP_STAMPS
        unsigned long long st[4] = {0, 0, 0, 0};
#endif
#pragma unroll 1
        for (int s = 0; s <= last + 1; ++s) {
            STAMP(t0);
            // this wave's pieces of group s-2 have landed (group s-1, the newest PC_CONS VMEM ops, may stay in flight; at a tile
            // boundary the epilogue's stores are newer still, which only makes this wait stricter)
            if (s >= 2 && s <= last - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PC_CONS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            STAMP(t1);
            ACCUM(0, t0, t1);                            // DMA wait + barrier
            dma_group(s);
            STAMP(t2);
            ACCUM(1, t1, t2);                            // DMA addresses
            if (s >= 2) {
                const int sc = s - 2, c = sc % NCHUNK;
                if (c == 0) {
                    // new tile: accumulators start at x + b2 (the residual rides in the accumulators, so all 48 row
                    // pieces are in flight at once here and the epilogue is store-only: no 12 dependent HBM round trips)
                    const int tile = blockIdx.x + (sc / NCHUNK) * gridDim.x;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int m = tile * 128 + pr * 32 + 16 * mt + frow;
                        const float* xr = x + (size_t)(m < M ? m : M - 1) * E + 8 * g;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            ACC(t, mt) = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(xr + 32 * (t >> 1) + 4 * (t & 1)));
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 32 * (t >> 1) + 8 * g + 4 * (t & 1));
                        ACC(t, 0) = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, ACC(t, 0)) + bv);
                        ACC(t, 1) = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, ACC(t, 1)) + bv);
                    }
                }
                // ---- GEMM2 of chunk c
                const vec8 hf0 = *reinterpret_cast<const vec8*>(hslot + (sc & 1) * 8192);
                const vec8 hf1 = *reinterpret_cast<const vec8*>(hslot + 1024 + (sc & 1) * 8192);
                const unsigned w2a = lds_base + W2_RING + (s % NSLOT) * W2_BYTES + frag_off;
                vec8 w[NT];
                wait_lgkm<0>();
                static_for<0, DEPTH>([&](auto i) { constexpr int q = decltype(i)::value; lds_read_b128<q * 1024>(w[q], w2a); });
                static_for<0, NT>([&](auto i) {
                    constexpr int q = decltype(i)::value;
                    if constexpr (q + DEPTH < NT) lds_read_b128<(q + DEPTH) * 1024>(w[q + DEPTH], w2a);
                    wait_lgkm<(NT - 1 - q < DEPTH) ? (NT - 1 - q) : DEPTH>();
                    ACC(q, 0) = __builtin_bit_cast(u32x4, mfma16(w[q], hf0, __builtin_bit_cast(f32x4, ACC(q, 0))));
                    ACC(q, 1) = __builtin_bit_cast(u32x4, mfma16(w[q], hf1, __builtin_bit_cast(f32x4, ACC(q, 1))));
                });
                STAMP(t3);
                ACCUM(2, t2, t3);                        // GEMM2
                if (c == NCHUNK - 1) {
                    // ---- epilogue of tile sc/48: store x (residual already inside), next LayerNorm from registers
                    const int tile = blockIdx.x + (sc / NCHUNK) * gridDim.x;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int m = tile * 128 + pr * 32 + 16 * mt + frow;
                        float* xw = x + (size_t)(m < M ? m : M - 1) * E + 8 * g;
                        if (m < M) {
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                *reinterpret_cast<f32x4*>(xw + 32 * (t >> 1) + 4 * (t & 1)) = __builtin_bit_cast(f32x4, ACC(t, mt));
                        }
                        if (xn_out) {
                            float sum = 0.f;
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const f32x4 a = __builtin_bit_cast(f32x4, ACC(t, mt));
                                sum += a[0] + a[1] + a[2] + a[3];
                            }
                            sum += __shfl_xor(sum, 16, 64);
                            sum += __shfl_xor(sum, 32, 64);
                            const float mean = sum * (1.0f / E);
                            float sq = 0.f;
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const f32x4 a = __builtin_bit_cast(f32x4, ACC(t, mt));
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const float d = a[r] - mean; sq = fmaf(d, d, sq); }
                            }
                            sq += __shfl_xor(sq, 16, 64);
                            sq += __shfl_xor(sq, 32, 64);
                            const float rstd = rsqrtf(sq * (1.0f / E) + eps);
                            if (m < M) {
                                T* xo = xn_out + (size_t)m * E + 8 * g;
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks) {
                                    vec8 o;
                                    const f32x4 a0 = __builtin_bit_cast(f32x4, ACC(2 * ks, mt));
                                    const f32x4 a1 = __builtin_bit_cast(f32x4, ACC(2 * ks + 1, mt));
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        o[r] = (T)((a0[r] - mean) * rstd);
                                        o[4 + r] = (T)((a1[r] - mean) * rstd);
                                    }
                                    *reinterpret_cast<vec8*>(xo + 32 * ks) = o;
                                }
                            }
                        }
                    }
                }
                STAMP(t4);
                ACCUM(3, t3, t4);                        // epilogue (c == 47 only)
            }
        }
#ifdef MLP_STAMPS
        if (lane == 0) for (int i = 0; i < 4; ++i) g_stamps[(blockIdx.x * 8 + wave) * 4 + i] = st[i];
#endif
    }
}

template <typename T>
int launch_t(float* x, void* xn_out, const void* wpack, const float* b1f, const float* b2, int64_t M, float eps,
             hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = mlp16_kernel<T>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int ntiles = (int)((M + 127) / 128);
    const int cus = mst_persistent_grid();
    const int nblk = ntiles < cus ? ntiles : cus;
    kern<<<dim3(nblk), dim3(512), LDS_BYTES, s>>>(x, (T*)xn_out, (const char*)wpack, b1f, b2, nullptr, (int)M, ntiles, eps);
    return mst_check_launch("mlp16");
}

}  // namespace

#ifdef MLP_STAMPS
extern "C" int mst_debug_mlp_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

int launch_mlp16(float* x, void* xn_out, int dt, const void* wpack, const float* b1f, const float* b2, int64_t M, int E_,
                 float eps, hipStream_t s) {
    MST_CHECK_ARG(E_ == E, "mlp_fused: embed_dim=%d unsupported (384)", E_);
    MST_CHECK_ARG(M > 0 && M < (1ll << 31) - 128, "mlp_fused: bad M");
    if (dt == MST_BF16) return launch_t<bf16_t>(x, xn_out, wpack, b1f, b2, M, eps, s);
    if (dt == MST_F16) return launch_t<f16_t>(x, xn_out, wpack, b1f, b2, M, eps, s);
    mst_set_error("mlp_fused: dtype %d unsupported (f16 / bf16)", dt);
    return MST_EINVAL;
}
