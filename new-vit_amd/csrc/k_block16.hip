// Everything of a ViT block after the attention kernel, for E = 384 and 16-bit MFMA operands, in ONE launch:
//     x  <- x + ls1 * (proj(attn_out) + b_proj)                       attention.py:67-68; block.py:90-91,112
//     x  <- x + ls2 * (fc2(gelu(fc1(LayerNorm2(x)))) + b2)            block.py:93-94,113; mlp.py:34-40
//     xn <- normalise(x)      (optional: the NEXT block's norm1, its affine folded into that block's QKV weights)
// The residual stream x is read ONCE and written ONCE per block (the separate out-projection launch re-read and re-wrote
// all of it: an HBM-bound 1.35 GB round trip per block), the 4E-wide hidden activations never leave the CU, and neither
// LayerNorm needs a pass of its own.
//
// Structure (k_mlp16.hip's, extended): persistent, one 8-wave workgroup per CU, 128 token rows per tile, PRODUCER /
// CONSUMER wave roles (waves w and w+4 share a SIMD).  A tile takes P = 62 steps, one `s_barrier` each:
//   step  0..11  consumers: out-projection as 12 more "GEMM2-type" chunks of 32 k: the accumulators (y^T [384 x 32 rows],
//                192 registers) start at x + ls1*b_proj and take W_proj chunk j (streamed through the W2 ring) times the
//                attention-output columns [32j, 32j+32) of their 32 rows, which LDS-DMA delivers (per-lane SOURCE
//                addresses) in exactly the hand-off layout the producers use for H; after step 11 the accumulators hold
//                x_mid: LayerNorm2 from the registers, the normalised rows go to the producer twin wave through a
//                96 KB per-workgroup global scratch (lane-private, 1 KiB per store), and b2 is added.   producers: idle
//   step 12..59  producers: GEMM1 of hidden chunk c = step-12 (32 units) against the normalised rows held as B fragments
//                in registers, GELU of the previous chunk under the MFMAs, hand-off through the 16 KB LDS slot
//   step 14..61  consumers: GEMM2 of chunk step-14; step 61: store x, next LayerNorm from the registers -> xn
// Weight images (W_proj chunks, W1 rows, W2 columns) are pre-packed on the host as swizzled LDS images and stream through
// two 3-deep LDS rings by LDS-DMA two steps ahead; the ring slot of whatever a step consumes is (global step) % 3, and the
// attention-output pieces of the out-projection steps use the W1 ring, which GEMM1 does not need during those steps.
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int E = 384, HID = 1536, CH = 32;            // CH hidden units (or k of the out-projection) per chunk
constexpr int NCHUNK = HID / CH;                        // 48
constexpr int PJ = E / CH;                              // 12 out-projection steps
constexpr int J_G1 = PJ;                                // first GEMM1 step (chunk c at step J_G1 + c)
constexpr int J_G2 = PJ + 2;                            // first GEMM2 step (chunk c at step J_G2 + c)
constexpr int P = J_G2 + NCHUNK;                        // 62 steps per tile
constexpr int W1_BYTES = CH * E * 2;                    // 24 KiB
constexpr int W2_BYTES = E * CH * 2;                    // 24 KiB
constexpr int CHUNK_BYTES = W1_BYTES + W2_BYTES;        // packed MLP weights per chunk in global memory
constexpr int NSLOT = 3;
constexpr int W2_RING = NSLOT * W1_BYTES;               // byte offset of the W2 ring
constexpr int HBUF = 2 * NSLOT * W1_BYTES;              // byte offset of the H hand-off buffer (2 x 8 KiB)
constexpr int LDS_BYTES = HBUF + 2 * 8192;              // 163,840 = all of the LDS
constexpr int KS = E / 32;                              // 12 k-steps of GEMM1
constexpr int NT = E / 16;                              // 24 output tiles (16 columns) of GEMM2 / the out-projection
#ifndef BLOCK_DEPTH_P
#define BLOCK_DEPTH_P 4
#endif
#ifndef BLOCK_DEPTH_C
#define BLOCK_DEPTH_C 4
#endif
constexpr int DEPTH_P = BLOCK_DEPTH_P, DEPTH_C = BLOCK_DEPTH_C;   // fragment reads in flight ahead of the MFMAs (producer / consumer loop)
constexpr int SCRATCH_WG = 4 * 2 * KS * 1024;           // 96 KiB of normalised rows per workgroup
#ifndef BLOCK_TILE_CYCLES
#define BLOCK_TILE_CYCLES 190000                        // ~one 128-row tile (62 steps) in shader clocks: the de-phasing window
#endif

template <int OFF, typename V> __device__ __forceinline__ void lds_read_b128(V& dst, unsigned addr) {
#ifndef ABL_NO_READS
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
#else
    asm volatile("" : "=v"(dst) : "v"(addr));            // timing-only ablation: fragments are whatever the registers hold
#endif
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
template <typename T> __device__ __forceinline__ float gelu_blk(float v) {
#ifdef ABL_NO_GELU
    return v;                                            // timing-only ablation
#else
    return gelu_sig<T>(v);
#endif
}
// LDS-DMA pieces per wave in the group that lands for local step t (issued two steps earlier)
__device__ __forceinline__ int group_size(int t) {
    return t < PJ ? 8 : (t < J_G2 ? 6 : (t < J_G1 + NCHUNK ? 12 : 6));
}
__device__ __forceinline__ void wait_vm(int n) {        // n wave-uniform: all but the n newest vector-memory operations are done
    if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef BLOCK_STAMPS
// diagnostic build only (never shipped): per-wave cycle sums by phase, read back by mst_debug_block_stamps
__device__ unsigned long long g_bstamps[256 * 8 * 8];
__device__ __forceinline__ unsigned long long bstamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define BST(var) const unsigned long long var = bstamp()
#define BACC(slot, a, b) st[slot] += (b) - (a)
#else
#define BST(var)
#define BACC(slot, a, b)
#endif

#ifdef BLOCK_DEBUG
// diagnostic build only (never shipped): what each role saw of the LayerNorm2 hand-off, [wg][role][pr][i][lane] x 16 B
__device__ char* g_block_dbg = nullptr;
#endif

template <typename T>
__global__ __launch_bounds__(512) void block16_kernel(float* x, const T* attn, T* xn_out,
                                                      const char* __restrict__ wproj, const float* __restrict__ bproj,
                                                      const char* __restrict__ wpack, const float* __restrict__ b1f,
                                                      const float* __restrict__ b2, char* scratch, int M, int ntiles,
                                                      float eps) {
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
    // Global addresses are (wave-uniform 64-bit base, built with scalar arithmetic where it is used) + (32-bit lane offset):
    // 64-bit per-lane pointers precomputed outside the step loop would be hoisted by the compiler and spilled around it.
    const unsigned lane16 = lane * 16;
    // LayerNorm2 hand-off region of this wave pair (24 KiB, lane-private 16-byte slots: piece i at i*1024 + lane*16).
    // Plain global stores / loads on purpose.  `buffer_store_dwordx4 ... s<N> offen` (SGPR soffset) was tried first and is
    // WRONG on gfx950 with hipcc 7.2: the compiler pads the "VALU overwrites the data of a >64-bit store" hazard only for the
    // immediate-soffset form, the next v_pk_add reused the data registers at once, and lanes 12..15 of every 16 (the last of
    // the four passes in which the store reads its data) shipped fp32 bit patterns instead of the 16-bit rows.
    char* const xs_base = scratch + (size_t)blockIdx.x * SCRATCH_WG + (size_t)pr * (2 * KS * 1024);

    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    // ---- LDS-DMA group that lands for local step t of this workgroup's k-th tile, into ring slot `slot` (= global step % 3).
    // Consumers issue it two steps ahead.
#ifdef BLOCK_DMA_STRIDED                                 // round-2 first form: piece u of wave pr = image piece 4u + pr (1 KiB each)
    constexpr int WAVE_PIECES = 1024;
#else                                                    // wave pr takes the six CONSECUTIVE pieces 6 pr .. 6 pr + 5 (see dma_w)
    constexpr int WAVE_PIECES = 6144;
#endif
    char* const ring_wave = smem + pr * WAVE_PIECES;
    auto dma_w = [&](const char* src, char* dst) {       // a 24 KiB image (src wave-uniform): this wave's 6 pieces
#ifdef BLOCK_NODMA
        return;                                          // timing-only ablation: weights are whatever the LDS holds
#endif
#ifdef BLOCK_DMA_STRIDED
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds(GLB_PTR((src + u * 4096) + lane16), LDS_PTR(dst + u * 4096), 16, 0, 0);
#else
        // the instruction's immediate offset moves BOTH the global and the LDS address: one lane address, one M0, six immediates
        // (no 64-bit vector add and no M0 write per piece)
        const auto g = GLB_PTR((src + 2048) + lane16);
        const auto l = LDS_PTR(dst + 2048);
        __builtin_amdgcn_global_load_lds(g, l, 16, -2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, -1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
#endif
    };
    // rows of a tile this lane touches: row (16*mt + frow) of the pair's 32, clamped to the last valid row of the matrix
    auto row_off = [&](int tile, int mt) -> unsigned {
        const int last = M - 1 - tile * 128;             // >= 0
        int fr = frow;
        asm volatile("" : "+v"(fr));                     // lane addresses are rebuilt where they are used, never hoisted + spilled
        const int r = pr * 32 + 16 * mt + fr;
        return (unsigned)(r < last ? r : last);
    };
    auto dma_group = [&](int t, int k, int slot) {
        if (k >= my_tiles) return;
        if (t < PJ) {
            // attention-output columns [32t, 32t+32) of this pair's 32 rows -> W1 ring slot, in the H hand-off layout
            const int tile = blockIdx.x + k * gridDim.x;
            const char* const abase = (const char*)(attn + (size_t)tile * 128 * E + 32 * t);
            char* const d = smem + slot * W1_BYTES + pr * 2048;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                __builtin_amdgcn_global_load_lds(GLB_PTR(abase + (row_off(tile, mt) * (E * 2) + g * 16)), LDS_PTR(d + mt * 1024), 16, 0, 0);
            dma_w(wproj + (size_t)t * W2_BYTES + pr * WAVE_PIECES, ring_wave + W2_RING + slot * W2_BYTES);
            return;
        }
        if (t < J_G1 + NCHUNK) dma_w(wpack + (size_t)(t - J_G1) * CHUNK_BYTES + pr * WAVE_PIECES, ring_wave + slot * W1_BYTES);
        if (t >= J_G2) dma_w(wpack + (size_t)(t - J_G2) * CHUNK_BYTES + W1_BYTES + pr * WAVE_PIECES, ring_wave + W2_RING + slot * W2_BYTES);
    };
    if (!producer) {
        dma_group(0, 0, 0);
        dma_group(1, 0, 1);
    }

    // De-phase the persistent workgroups (k_mlp16.hip): spread the tile boundaries (HBM bursts) over the tile period.
    if (my_tiles >= 4) {
        const int min_tiles = ntiles / (int)gridDim.x;
        const unsigned u = (((unsigned)blockIdx.x >> 3) + 5u * ((unsigned)blockIdx.x & 7u)) & 15u;   // 0..15
        const unsigned long long delay = (unsigned long long)BLOCK_TILE_CYCLES * ((my_tiles > min_tiles ? 0u : 16u) + u) / 32u;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 640 && __builtin_readcyclecounter() - t0 < delay; ++it) __builtin_amdgcn_s_sleep(32);   // bounded: every wave leaves
    }

    const int nsteps = my_tiles * P;
#ifdef BLOCK_CONS_PRIO
    // static priority for the consumer half (waves 4-7: the younger half, loser of the VALU arbitration, and the critical path)
    if (!producer) __builtin_amdgcn_s_setprio(BLOCK_CONS_PRIO);
#endif
#ifdef BLOCK_PROD_PRIO
    if (producer) __builtin_amdgcn_s_setprio(BLOCK_PROD_PRIO);
#endif
#define ACC(t, mt) R[2 * (t) + (mt)]
#define XA(mt, ks) R[KS * (mt) + (ks)]
    if (producer) {
        u32x4 R[2 * KS];                                 // xa[mt][ks]: normalised rows, GEMM1 B operand
        f32x4 bias_next[2];
        bias_next[0] = *reinterpret_cast<const f32x4*>(b1f + 4 * g);
        bias_next[1] = *reinterpret_cast<const f32x4*>(b1f + 16 + 4 * g);
        f32x4 hp[2][2] = {};                             // pre-activations of the previous chunk [hidden tile][row tile]
        auto store_h = [&](const float (&gv)[16], int c) {   // k order of GEMM2 = (hidden tile, register) order of GEMM1
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                vec8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (T)gv[8 * mt + j];
                *reinterpret_cast<vec8*>(hslot + mt * 1024 + (c & 1) * 8192) = hv;
            }
        };
        int j = 0, k = 0, slot = 0;
#ifdef BLOCK_WARM
        float warm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
#ifdef BLOCK_STAMPS
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long tstart = bstamp();
#endif
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            BST(p0);
#ifdef BLOCK_WARM
            // GEMM1 steps: own loads (bias, normalised rows) and the H(s-2) writes are drained before the barrier.  In the idle
            // steps nothing of this wave's vector memory matters to anyone, and the pre-touch loads below must NOT be waited for.
            if (j >= J_G1 && j <= J_G1 + NCHUNK) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (j == J_G1) asm volatile("" ::"v"(warm[0]), "v"(warm[1]), "v"(warm[2]), "v"(warm[3]), "v"(warm[4]), "v"(warm[5]), "v"(warm[6]), "v"(warm[7]), "v"(warm[8]));
#else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // own loads / H(s-2) writes drained
#endif
            BST(p1);
            if (j > J_G1 && j <= J_G1 + NCHUNK) BACC(0, p0, p1); else BACC(2, p0, p1);
            if (j == J_G1) {
                // normalised rows of this tile from the consumer twin wave: written by the same lanes with plain stores that were
                // drained (vmcnt(0)) before this barrier; same CU, same L1: workgroup-scope visibility needs nothing more
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16));            // address = scalar base + this one 32-bit register, built here (not hoisted)
#pragma unroll
                for (int i = 0; i < 2 * KS; ++i)
                    R[i] = *reinterpret_cast<const u32x4*>((xs_base + i * 1024) + l16);
#ifdef BLOCK_DEBUG
                if (g_block_dbg)
                    for (int i = 0; i < 2 * KS; ++i)
                        *reinterpret_cast<u32x4*>(g_block_dbg + ((((size_t)blockIdx.x * 2 + 1) * 4 + pr) * 2 * KS + i) * 1024 + lane16) = R[i];
#endif
            }
            if (j >= J_G1 && j < J_G1 + NCHUNK) {
                const int c = j - J_G1;
                // ---- GEMM1 of chunk c: 32 hidden units x 32 rows (accumulators start at b1), GELU of chunk c-1, hand-off
                const f32x4 b0 = bias_next[0], b1v = bias_next[1];
                {
                    const int cn = (c + 1) % NCHUNK;     // b1 of the next step's chunk, one step ahead
                    bias_next[0] = *reinterpret_cast<const f32x4*>(b1f + cn * CH + 4 * g);
                    bias_next[1] = *reinterpret_cast<const f32x4*>(b1f + cn * CH + 16 + 4 * g);
                }
                f32x4 h[2][2] = {{b0, b0}, {b1v, b1v}};  // [hidden tile][row tile]
#ifdef BLOCK_GELU_FIRST
                // GELU + hand-off of the PREVIOUS chunk FIRST: its vector work (16 exp2 + 16 rcp + ~60 plain ops per lane) runs while
                // the consumer twin on this SIMD has the matrix pipe to itself for its 48 MFMAs; this wave's own 48 MFMAs follow
                // when the pipe is free again.  (hipcc otherwise sinks the GELU below the MFMA loop -- both waves then share the
                // pipe first and the GELU runs alone afterwards: 2,650 instead of ~1,700 cycles per step.)
                if (c != 0) {
                    float gv0[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) gv0[e] = gelu_blk<T>(hp[(e >> 2) & 1][e >> 3][e & 3]);
                    store_h(gv0, c - 1);                 // H(c-1)
                }
                __builtin_amdgcn_sched_barrier(0);
#endif
                float gv[16];                            // (BLOCK_GELU_LAST: GELU of the previous chunk after this chunk's MFMAs)
                const unsigned w1a = lds_base + slot * W1_BYTES + frag_off;
                vec8 w[2 * KS];                          // fragment index q = 2*ks + ht  ->  byte offset ks*2048 + ht*1024
                wait_lgkm<0>();
                static_for<0, DEPTH_P>([&](auto i) { constexpr int q = decltype(i)::value; lds_read_b128<(q >> 1) * 2048 + (q & 1) * 1024>(w[q], w1a); });
                static_for<0, 2 * KS>([&](auto i) {
                    constexpr int q = decltype(i)::value;
                    if constexpr (q + DEPTH_P < 2 * KS) lds_read_b128<((q + DEPTH_P) >> 1) * 2048 + ((q + DEPTH_P) & 1) * 1024>(w[q + DEPTH_P], w1a);
                    wait_lgkm<(2 * KS - 1 - q < DEPTH_P) ? (2 * KS - 1 - q) : DEPTH_P>();
#ifndef ABL_NO_MFMA_P
                    h[q & 1][0] = mfma16(w[q], __builtin_bit_cast(vec8, XA(0, q >> 1)), h[q & 1][0]);
                    h[q & 1][1] = mfma16(w[q], __builtin_bit_cast(vec8, XA(1, q >> 1)), h[q & 1][1]);
#else
                    asm volatile("" ::"v"(w[q]));
#endif
#ifndef BLOCK_GELU_FIRST
#if !defined(BLOCK_GELU_GROUP) || BLOCK_GELU_GROUP == 0
                    // (hipcc sinks these below the whole MFMA loop: the values are not needed before store_h)
                    if constexpr (q % 3 != 2) {
                        constexpr int e = q - q / 3;
                        gv[e] = gelu_blk<T>(hp[(e >> 2) & 1][e >> 3][e & 3]);
                    }
#else
                    // GELU of the previous chunk really interleaved with this chunk's MFMAs: GG values per carrying iteration,
                    // pinned there by an empty asm (independent chains of one iteration hide each other's latency)
                    constexpr int GG = BLOCK_GELU_GROUP, EVERY = 24 / (16 / GG);
                    if constexpr (GG == 1 ? (q % 3 != 2) : (q % EVERY == 0)) {
                        constexpr int e0 = GG == 1 ? q - q / 3 : GG * (q / EVERY);
#pragma unroll
                        for (int u = 0; u < GG; ++u) gv[e0 + u] = gelu_blk<T>(hp[((e0 + u) >> 2) & 1][(e0 + u) >> 3][(e0 + u) & 3]);
#pragma unroll
                        for (int u = 0; u < GG; ++u) asm volatile("" : "+v"(gv[e0 + u]));
                    }
#endif
#endif
                });
#ifndef BLOCK_GELU_FIRST
                if (c != 0) store_h(gv, c - 1);          // H(c-1)
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) hp[i][jj] = h[i][jj];
            } else if (j == J_G1 + NCHUNK) {             // drain: GELU + hand-off of the last chunk
                float gv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) gv[e] = gelu_blk<T>(hp[(e >> 2) & 1][e >> 3][e & 3]);
                store_h(gv, NCHUNK - 1);
#ifdef BLOCK_WARM
                if (k + 1 < my_tiles) {
                    // The NEXT tile's residual rows (196 KB) and attention-output rows (96 KB) are pulled towards the L2 now: the
                    // consumers read the former at their step 0 in one burst and the latter by LDS-DMA only two (short) steps
                    // ahead of each out-projection step -- cold, both are HBM round trips on the critical path (stamps: 3,900
                    // cycles per out-projection step, 17,700 for the x rows).  One dword per 128-byte line, 9 per lane, results
                    // consumed (= waited for) only at step J_G1 of the next tile.
                    const int tile = blockIdx.x + (k + 1) * gridDim.x;
                    const int rows = M - tile * 128 < 128 ? M - tile * 128 : 128;
                    const char* const xb = (const char*)(x + (size_t)tile * 128 * E);
                    const char* const ab = (const char*)(attn + (size_t)tile * 128 * E);
                    unsigned l = (unsigned)(pr * 64 + lane);
                    asm volatile("" : "+v"(l));
                    const unsigned xl = (unsigned)rows * 12u, al = (unsigned)rows * 6u;     // 128-byte lines of the two row sets
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const unsigned li = l + q * 256;
                        warm[q] = *reinterpret_cast<const float*>(xb + (size_t)(li < xl ? li : xl - 1) * 128);
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned li = l + q * 256;
                        warm[6 + q] = *reinterpret_cast<const float*>(ab + (size_t)(li < al ? li : al - 1) * 128);
                    }
                }
#endif
            }
            BST(p2);
            if (j > J_G1 && j <= J_G1 + NCHUNK) BACC(1, p1, p2); else if (j == J_G1) BACC(3, p1, p2);
            if (++j == P) { j = 0; ++k; }
            if (++slot == NSLOT) slot = 0;
        }
#ifdef BLOCK_WARM
        asm volatile("" ::"v"(warm[0]), "v"(warm[1]), "v"(warm[2]), "v"(warm[3]), "v"(warm[4]), "v"(warm[5]), "v"(warm[6]), "v"(warm[7]), "v"(warm[8]));
#endif
#ifdef BLOCK_STAMPS
        st[7] = bstamp() - tstart;
        if (lane == 0) for (int i = 0; i < 8; ++i) g_bstamps[(blockIdx.x * 8 + wave) * 8 + i] = st[i];
#endif
    } else {
        u32x4 R[2 * NT];                                 // acc[t][mt]: y^T accumulators (x rides inside)
        auto load_x = [&](int tile) {                    // all 48 row pieces in flight at once (one HBM round trip)
            const char* const xbase = (const char*)(x + (size_t)tile * 128 * E);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const char* xr = xbase + (row_off(tile, mt) * (E * 4) + g * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#if defined(BLOCK_NT) || defined(BLOCK_NT_XLD)
                    ACC(t, mt) = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 128 * (t >> 1) + 16 * (t & 1))));
#else
                    ACC(t, mt) = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(xr + 128 * (t >> 1) + 16 * (t & 1)));
#endif
            }
        };
        auto add_bias = [&](const float* b) {            // 8 vectors in flight per round (the accumulators leave ~40 registers)
            const float* const bl = b + 8 * g;
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += 8) {
                f32x4 bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = *reinterpret_cast<const f32x4*>(bl + 32 * ((t0 + i) >> 1) + 4 * ((t0 + i) & 1));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ACC(t0 + i, 0) = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, ACC(t0 + i, 0)) + bv[i]);
                    ACC(t0 + i, 1) = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, ACC(t0 + i, 1)) + bv[i]);
                }
            }
        };
        // Complete rows sit in the accumulators.  FINAL: store x and (optionally) the row-major normalised rows of the next block;
        // otherwise: normalised x_mid to the producer twin's lane-private scratch, fragment by fragment.
        auto rows_done = [&](auto final_tag, int tile) {
            constexpr bool FINAL = decltype(final_tag)::value;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                unsigned fr = frow, l16 = lane16;
                asm volatile("" : "+v"(fr), "+v"(l16));  // (see row_off)
                const unsigned rloc = pr * 32 + 16 * mt + fr;
                const bool valid = tile * 128 + (int)rloc < M;
                if (FINAL && valid) {
                    char* xw = (char*)(x + (size_t)tile * 128 * E) + (rloc * (E * 4) + g * 32);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#if defined(BLOCK_NT) || defined(BLOCK_NT_XST)
                        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, ACC(t, mt)), reinterpret_cast<f32x4*>(xw + 128 * (t >> 1) + 16 * (t & 1)));
#else
                        *reinterpret_cast<f32x4*>(xw + 128 * (t >> 1) + 16 * (t & 1)) = __builtin_bit_cast(f32x4, ACC(t, mt));
#endif
                }
                if (FINAL && !xn_out) continue;
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
                float mean_o = mean;                     // opaque copy: otherwise the 96 centred values of the variance pass are
                asm volatile("" : "+v"(mean_o));         // kept (and spilled) for the output pass instead of being recomputed
                char* const xo = (char*)(xn_out + (size_t)tile * 128 * E) + (rloc * (E * 2) + g * 16);
                if (FINAL && !valid) continue;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    vec8 o;
                    const f32x4 a0 = __builtin_bit_cast(f32x4, ACC(2 * ks, mt));
                    const f32x4 a1 = __builtin_bit_cast(f32x4, ACC(2 * ks + 1, mt));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (T)((a0[r] - mean_o) * rstd);
                        o[4 + r] = (T)((a1[r] - mean_o) * rstd);
                    }
#if defined(BLOCK_NT) || defined(BLOCK_NT_XN)
                    if constexpr (FINAL) __builtin_nontemporal_store(o, reinterpret_cast<vec8*>(xo + 64 * ks));
#else
                    if constexpr (FINAL) *reinterpret_cast<vec8*>(xo + 64 * ks) = o;
#endif
                    else {
                        *reinterpret_cast<vec8*>((xs_base + (KS * mt + ks) * 1024) + l16) = o;
#ifdef BLOCK_DEBUG
                        if (g_block_dbg)
                            *reinterpret_cast<vec8*>(g_block_dbg + ((((size_t)blockIdx.x * 2 + 0) * 4 + pr) * 2 * KS + (KS * mt + ks)) * 1024 + lane16) = o;
#endif
                    }
                }
            }
        };
        int j = 0, k = 0, slot = 0;
#ifdef BLOCK_STAMPS
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long tstart = bstamp();
#endif
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            BST(c0);
            int jt_late = 0, kt_late = 0, st_late = 0;
            // this wave's pieces of the group for THIS step have landed; the group for the next step may stay in flight.
            // Step J_G1 is different: the normalised rows stored at the end of step J_G1-1 must have left before the barrier.
            {
                int jn = j + 1, kn = k;
                if (jn == P) { jn = 0; ++kn; }
                wait_vm((j == J_G1 || kn >= my_tiles) ? 0 : group_size(jn));
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            BST(c1);
            if (j >= J_G2 && j < P - 1) BACC(0, c0, c1);
            {
                int jt = j + 2, kt = k;
                if (jt >= P) { jt -= P; ++kt; }
                int sl = slot + 2;
                if (sl >= NSLOT) sl -= NSLOT;
#ifndef BLOCK_DMA_LATE
                dma_group(jt, kt, sl);
#else
                jt_late = jt; kt_late = kt; st_late = sl;
#endif
            }
            const int tile = blockIdx.x + k * gridDim.x;
            BST(c2);
            if (j == 0) {
                // new tile: accumulators start at x + ls1*b_proj (the residual rides in the accumulators: x is read once).
                load_x(tile);
                add_bias(bproj);
            }
            BST(c3);
            if (j == 0) BACC(3, c2, c3);
            if (j < PJ || j >= J_G2) {
                // ---- one 32-deep chunk: y^T[384 x 32 rows] += Wc[384 x 32] . B^T, B = attention output columns (from the W1 ring)
                // or H of MLP chunk j - J_G2 (from the hand-off slot)
                const unsigned hb = lds_base + pr * 2048 + lane * 16 + ((j < PJ) ? slot * W1_BYTES : HBUF + ((j - J_G2) & 1) * 8192);
                vec8 hf0, hf1;
                lds_read_b128<0>(hf0, hb);               // older than every W read below: the first counted wait covers them
                lds_read_b128<1024>(hf1, hb);
                const unsigned w2a = lds_base + W2_RING + slot * W2_BYTES + frag_off;
                vec8 w[NT];
                static_for<0, DEPTH_C>([&](auto i) { constexpr int q = decltype(i)::value; lds_read_b128<q * 1024>(w[q], w2a); });
                static_for<0, NT>([&](auto i) {
                    constexpr int q = decltype(i)::value;
                    if constexpr (q + DEPTH_C < NT) lds_read_b128<(q + DEPTH_C) * 1024>(w[q + DEPTH_C], w2a);
                    wait_lgkm<(NT - 1 - q < DEPTH_C) ? (NT - 1 - q) : DEPTH_C>();
#ifndef ABL_NO_MFMA_C
                    ACC(q, 0) = __builtin_bit_cast(u32x4, mfma16(w[q], hf0, __builtin_bit_cast(f32x4, ACC(q, 0))));
                    ACC(q, 1) = __builtin_bit_cast(u32x4, mfma16(w[q], hf1, __builtin_bit_cast(f32x4, ACC(q, 1))));
#else
                    asm volatile("" ::"v"(w[q]), "v"(hf0), "v"(hf1));
#endif
                });
            }
#ifdef BLOCK_DMA_LATE
            dma_group(jt_late, kt_late, st_late);        // issue AFTER this step's MFMAs (it lands for step + 2 either way)
#endif
            BST(c4);
            if (j >= J_G2 && j < P - 1) BACC(1, c1, c4);
            else if (j < PJ) BACC(2, c0, c4);
            else if (j < J_G2) BACC(6, c0, c4);
            if (j == PJ - 1) {
                // the accumulators hold x_mid (after the out-projection): LayerNorm2 for the producer twin, then + ls2*b2
                rows_done(std::false_type{}, tile);
                add_bias(b2);
            } else if (j == P - 1) {
                // the block's output rows: x, the next block's normalised rows; then request the next tile's rows
                rows_done(std::true_type{}, tile);
            }
            BST(c5);
            if (j == PJ - 1) BACC(4, c4, c5);
            else if (j == P - 1) { BACC(5, c4, c5); BACC(5, c0, c4); }
            if (++j == P) { j = 0; ++k; }
            if (++slot == NSLOT) slot = 0;
        }
#ifdef BLOCK_STAMPS
        st[7] = bstamp() - tstart;
        if (lane == 0) for (int i = 0; i < 8; ++i) g_bstamps[(blockIdx.x * 8 + wave) * 8 + i] = st[i];
#endif
    }
}

template <typename T>
int launch_t(float* x, const void* attn, void* xn_out, const void* wproj, const float* bproj, const void* wpack,
             const float* b1f, const float* b2, void* scratch, int64_t M, float eps, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = block16_kernel<T>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int ntiles = (int)((M + 127) / 128);
    const int cus = mst_persistent_grid();
    const int nblk = ntiles < cus ? ntiles : cus;
    kern<<<dim3(nblk), dim3(512), LDS_BYTES, s>>>(x, (const T*)attn, (T*)xn_out, (const char*)wproj, bproj, (const char*)wpack,
                                                  b1f, b2, (char*)scratch, (int)M, ntiles, eps);
    return mst_check_launch("block16");
}

}  // namespace

#ifdef BLOCK_STAMPS
extern "C" int mst_debug_block_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * n);
}
#endif

#ifdef BLOCK_DEBUG
extern "C" int mst_debug_block_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_block_dbg), &p, sizeof(p)); }
#endif

size_t block16_scratch_bytes(void) { return (size_t)mst_persistent_grid() * SCRATCH_WG; }

int launch_block16(float* x, const void* attn, void* xn_out, int dt, const void* wproj, const float* bproj, const void* wpack,
                   const float* b1f, const float* b2, void* scratch, int64_t M, int E_, float eps, hipStream_t s) {
    MST_CHECK_ARG(E_ == E, "block_fused: embed_dim=%d unsupported (384)", E_);
    MST_CHECK_ARG(M > 0 && M < (1ll << 31) - 128, "block_fused: bad M");
    if (dt == MST_BF16) return launch_t<bf16_t>(x, attn, xn_out, wproj, bproj, wpack, b1f, b2, scratch, M, eps, s);
    if (dt == MST_F16) return launch_t<f16_t>(x, attn, xn_out, wproj, bproj, wpack, b1f, b2, scratch, M, eps, s);
    mst_set_error("block_fused: dtype %d unsupported (f16 / bf16)", dt);
    return MST_EINVAL;
}
