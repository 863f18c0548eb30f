// Flash-style softmax(q k^T) v for the per-slice ViT blocks: head_dim 64, non-causal, arbitrary N
// (257 at 224^2, 1370 at 518^2) with tail masking; 16-bit MFMA operands, fp32 scores/softmax/acc.
// The [N,N] score matrix of attention.py:61-63 is never materialised.
//
// Workgroup = 4 waves x 32 query rows; K/V tiles of 64 keys are shared through a 2-deep LDS ring,
// register-staged (global loads for tile t+1 are issued before the MFMAs of tile t and written to
// LDS after them: one barrier per tile).  Per wave and tile:
//   S^T[key][q] = K . Q^T   (v_mfma_f32_32x32x16, K rows via ds_read_b128 from a 144-byte-row
//                            image, Q fragments resident in registers, pre-multiplied by log2 e; the chain starts with one
//                            extra MFMA that puts -r[q], the per-query softmax reference point, into every key row)  -> the query sits on the
//                            lane, so the row max / sum are in-lane plus ONE cross-half shuffle;
//   O^T[d][q]  += V^T . P^T  the S^T accumulator registers, converted to 16-bit in place, ARE the
//                            B operand (k order 16s+8(j>>2)+4h+(j&3)); the matching V^T fragment is
//                            fetched with ds_read_b64_tr_b16 (hardware transpose) from a row-major
//                            V image swizzled so the 32-lane halves cover one full bank row.
// q arrives pre-scaled by head_dim^-0.5 (folded into the QKV projection epilogue).
//
// Reference arithmetic: attention.py:56-66 (and dino.py:226-243 for the stored probabilities).
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KV_TILE_BYTES = 64 * 128;  // V tile: 64 keys x 64 dims x 2 B
// K rows are PADDED to 144 B instead of XOR-swizzled: 36-dword rows put the 16 lanes of a ds_read_b128 group on 16 distinct
// 4-bank groups (conflict-free), and every fragment address of a tile becomes ONE lane offset plus an immediate (the
// XOR made each of the 8 reads its own lane function: ~25 address VALU ops per tile in an issue-bound loop).
constexpr int K_ROW = 144, K_TILE_BYTES = 64 * K_ROW;
// -DATTN_NO_DMA restores the register-staged K/V tiles (global_load -> ds_write_b128 behind the MFMAs).  Default: LDS-DMA, 4 pieces
// of 1 KiB per wave and tile (waves 0-1 the K tile, 2-3 the V tile), both tiles as 128-byte rows whose 16-byte slots are permuted on
// the SOURCE side (K: slot = chunk ^ ((row >> 1) & 7), V: chunk ^ (((row >> 1) & 1) << 2)), so a piece is still 8 whole rows.
#ifndef ATTN_NO_DMA
#define ATTN_DMA 1
#endif
// (Tried in round 2 and removed in round 3: fragment reads in asm, four in flight behind counted lgkmcnt waits instead of the compiler's
// one read per MFMA -- 0.978 vs 0.970 ms: with four waves per SIMD the exposed LDS latency is already covered by the other waves.)

// c + a.x + a.y for a pair of 16-bit values (v_dot2_f32_bf16 / v_dot2_f32_f16 against (1, 1)): the row sums of the ROUNDED probabilities,
// two per instruction
__device__ __forceinline__ float sum2(bf16_t x, bf16_t y, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    const v2 a = {x, y}, one = {(bf16_t)1.0f, (bf16_t)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(a, one, c, false);
}
__device__ __forceinline__ float sum2(f16_t x, f16_t y, float c) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    const v2 a = {x, y}, one = {(f16_t)1.0f, (f16_t)1.0f};
    return __builtin_amdgcn_fdot2(a, one, c, false);
}

template <typename T>
__device__ __forceinline__ typename V8<T>::type tr_pair(const char* p_lo, const char* p_hi) {
    typedef typename V8<T>::type vec8;
    union { struct { s16x4 lo, hi; } s; vec8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_lo));
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_hi));
    return u.v;
}

// The softmax reference point is subtracted by the matrix pipe (one extra MFMA per 32 keys instead of 32 vector subtractions:
// measured -5 % on the bench shape when it went in, -9 % against the fma form re-measured with the LDS-DMA tiles).
#ifndef ATTN_WG_WAVES
#define ATTN_WG_WAVES 4      // query rows per workgroup = 32 x waves (K/V tiles shared by the workgroup); 8 waves measured 1.5 % slower at N = 1370 (6 x 256 rows pad 12 %, 11 x 128 pad 3 %)
#endif
constexpr int WGW = ATTN_WG_WAVES, WGT = 64 * WGW, QB = 32 * WGW;
#ifndef ATTN_WAVES_PER_EU
#define ATTN_WAVES_PER_EU 4   // 128 VGPRs (5 spilled dwords): 4 waves per SIMD measured 3.4 % faster than 3 at 148
#endif
template <typename T>
__global__ __launch_bounds__(WGT) __attribute__((amdgpu_waves_per_eu(ATTN_WAVES_PER_EU, ATTN_WAVES_PER_EU))) void attn16_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                     int N, int heads, int log2q, int out_blocked) {
    typedef typename V8<T>::type vec8;
#ifdef ATTN_DMA
    constexpr int KT_BYTES = KV_TILE_BYTES;              // unpadded K rows
#else
    constexpr int KT_BYTES = K_TILE_BYTES;
#endif
    __shared__ __attribute__((aligned(16))) char smem[2 * KT_BYTES + 2 * KV_TILE_BYTES];
    char* const Ks = smem;
    char* const Vs = smem + 2 * KT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: the q-blocks of one (sequence, head) get consecutive tile ids on ONE XCD, so its
    // K/V (re-read by every q-block) stay in that XCD's L2 (plain (x,y,z) order deals them over all 8 XCDs:
    // rocprofv3 FETCH_SIZE showed 5.7x the algorithmic bytes).
    const int nqb = (N + QB - 1) / QB;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = tile % nqb, h = (tile / nqb) % heads, seq = tile / (nqb * heads);
    const int E = heads * 64, ld = 3 * E;
    const T* base = qkv + (int64_t)seq * N * ld;
    const int h2 = lane >> 5;

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q = lane&31][d = 16*ds + 8*h2 + j]
    const int q = qb * QB + wave * 32 + (lane & 31);
    const bool wave_active = (qb * QB + wave * 32) < N;
    vec8 bq[4];
    {
        const int qc = q < N ? q : N - 1;
        const T* qp = base + (int64_t)qc * ld + h * 64 + h2 * 8;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) bq[ds] = *reinterpret_cast<const vec8*>(qp + ds * 16);
    }

#ifdef ATTN_DMA
    static_assert(WGW == 4 || WGW == 8, "the LDS-DMA staging map deals 16 pieces per tile to 4 or 8 waves");
    // piece p = PPW wave + u: p < 8 -> K rows 8p .. 8p+7, else V rows 8(p-8) ..; lane l -> row + (l >> 3), LDS slot l & 7
    constexpr int PPW = 16 / WGW;
    const int d_p0 = wave * PPW;                         // first piece of this wave (even: K odd-piece fix-up goes by u & 1)
    const int d_row0 = (d_p0 & 7) * 8 + (lane >> 3);     // + 8u
    const int d_isv = d_p0 >> 3;
    const int d_chunk0 = d_isv ? ((lane & 7) ^ (((d_row0 >> 1) & 1) << 2)) : ((lane & 7) ^ ((d_row0 >> 1) & 7));
    const int d_col0 = (1 + d_isv) * (heads * 64) + h * 64 + d_chunk0 * 8;       // element column of piece u even; odd K pieces: ^ 32
    // steady state: a scalar tile base + one precomputed 32-bit lane offset per piece -- no vector ALU work per piece (per-piece
    // address arithmetic was ~40 issue cycles x 4 pieces of a ~1300-cycle tile: -8 % with half the pieces in a timing ablation)
    unsigned d_off[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) d_off[u] = (unsigned)(((d_row0 + 8 * u) * ld + ((u & 1) && !d_isv ? (d_col0 ^ 32) : d_col0)) * 2);
    auto dma = [&](int t, int buf) {
        char* dst = (d_isv ? Vs + buf * KV_TILE_BYTES : Ks + buf * KT_BYTES) + (d_p0 & 7) * 1024;
#ifdef ATTN_DMA_VADDR
        if (false) {
#else
        if (t * 64 + 64 <= N) {
#endif
            const char* tb = reinterpret_cast<const char*>(base) + (int64_t)t * 64 * ld * 2;
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
#ifdef ATTN_ABL_HALF_DMA
                if (u & 1) continue;
#endif
                // saddr + 32-bit lane offset, spelled out: hipcc widens the offsets to 64-bit register pairs and adds the
                // base per piece otherwise (8 more registers -> spills, one more vector instruction per piece)
                const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(dst + u * 1024);
                // (m0 is written here exactly as the builtin of the other branch writes it -- immediately before its one use;
                // nothing else in this kernel reads it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :
                             : "v"(d_off[u]), "s"(tb), "s"(m0v)
                             : "memory", "m0");
#pragma clang diagnostic pop
            }
        } else {                                          // ragged last tile: rows beyond N re-read row N-1 (masked later)
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                int key = t * 64 + d_row0 + 8 * u;
                key = key < N ? key : N - 1;
                const int col = (u & 1) && !d_isv ? (d_col0 ^ 32) : d_col0;
                __builtin_amdgcn_global_load_lds(GLB_PTR(base + (int64_t)key * ld + col), LDS_PTR(dst + u * 1024), 16, 0, 0);
            }
        }
    };
#endif
    // ---- K/V staging map: 64 key rows x 8 chunks of 16 B per operand; thread -> key row sr, CPT consecutive chunks
    constexpr int CPT = 512 / WGT;                       // chunks per thread and operand: 2 (4 waves) or 1 (8 waves)
    const int sr = tid / (8 / CPT), sc0 = (tid % (8 / CPT)) * CPT;
    int k_off[CPT], v_off[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        k_off[c] = sr * K_ROW + (sc0 + c) * 16;
        v_off[c] = sr * 128 + (((sc0 + c) ^ (((sr >> 1) & 1) << 2)) * 16);
    }
    u32x4 rk[CPT], rv[CPT];
    auto gload = [&](int t) {
        int key = t * 64 + sr;
        key = key < N ? key : N - 1;
        const T* kp = base + (int64_t)key * ld + E + h * 64 + sc0 * 8;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            rk[c] = *reinterpret_cast<const u32x4*>(kp + 8 * c);
            rv[c] = *reinterpret_cast<const u32x4*>(kp + E + 8 * c);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            *reinterpret_cast<u32x4*>(Ks + buf * K_TILE_BYTES + k_off[c]) = rk[c];
            *reinterpret_cast<u32x4*>(Vs + buf * KV_TILE_BYTES + v_off[c]) = rv[c];
        }
    };

    // ---- per-lane LDS read offsets
    // K (ds_read_b128): row = kb*32 + (lane&31), 16-byte chunk 2*ds + h2 of a 144-byte row
    const int krow = lane & 31;
#ifdef ATTN_DMA
    int k_off4[4];                                       // chunk 2 ds + h2 of row krow: + kb*4096 immediate
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off4[ds] = krow * 128 + (((2 * ds + h2) ^ ((krow >> 1) & 7)) << 4);
#endif
    const int k_lane_off = krow * K_ROW + h2 * 16;       // + kb*32*K_ROW + ds*32: immediates
    // V (ds_read_b64_tr_b16): 16-lane group g: rows key0 + (i>>2), key0 = ks*16 + 4*h2,
    // columns db*32 + 16*(g&1) + 4*(i&3) .. +3  ->  chunk = db*4 + (g&1)*2 + ((i&3)>>1), +8 B if i odd
    const int vi = lane & 15;
    const int vrow = 4 * h2 + (vi >> 2);
    const int vsw = ((vrow >> 1) & 1) << 2;  // ks*16 and +8 do not change ((row>>1)&1)
    const int vchunk = ((lane >> 4) & 1) * 2 + ((vi & 3) >> 1);
    const int v_lane_off = vrow * 128 + (vi & 1) * 8;

    f32x16 oT[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[0][r] = oT[1][r] = 0.f;
    // Scores in the log2 domain (Q fragments pre-multiplied by log2 e) and RELATIVE to a per-query reference point r that the
    // matrix pipe subtracts: the chain of a score tile starts with one extra MFMA  ones[key][k] . (-r)[k][q]  instead of a
    // zero accumulator, so p = exp2(s) needs no per-element fma / sub in this issue-bound loop.  Softmax is invariant to the
    // reference point; r only has to stay within 2^RT of the running maximum, so it is a 16-bit value (exactly representable
    // as an MFMA operand) that moves only when a tile's maximum exceeds it by more than RT.
    if (!log2q) {                                        // stand-alone contract: q carries head_dim^-0.5 only
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
#pragma unroll
            for (int j = 0; j < 8; ++j) bq[ds][j] = (T)((float)bq[ds][j] * LOG2E);
    }
    float r_ref = 0.f, l_run = 0.f;                       // (the reference MFMA's operands are rebuilt from r_ref where a classic tile needs them:
    //                                                       eight registers that the fast loop does not carry)
    constexpr float RT = 8.0f;

    const int nt = (N + 63) >> 6;
#ifdef ATTN_DMA
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    gload(0);
    lstore(0);
#endif
    __syncthreads();

    // FAST mode.  The reference point only exists to keep exp2 inside the range of its type: when the first tile's maxima are small
    // (|max| <= TH0 in the log2 domain for every query of the wave) the reference stays 0 and the later tiles run WITHOUT the running
    // maximum (32 v_max + a cross-half swap) and WITHOUT the reference MFMA of each score chain; the tile's own row sum guards the
    // range instead (one compare: sum < GUARD means every p < GUARD), and a tile that fails it is recomputed the classic way and the
    // wave stays classic from there on.  Softmax is invariant to the reference point, so both modes give the same result up to
    // rounding; bf16 has fp32's exponent range (GUARD 2^100), fp16 needs p <= 65504 (GUARD 2^15, TH0 8 as RT).
    constexpr float TH0 = std::is_same<T, f16_t>::value ? 8.0f : 60.0f;
    constexpr float GUARD = std::is_same<T, f16_t>::value ? 32768.0f : 1.2676506e30f;
    bool fast = false;                                   // wave-uniform
#ifdef ATTN_NO_FAST
    constexpr bool fast_allowed = false;
#else
    constexpr bool fast_allowed = true;
#endif

    // one KV tile; MASK = the ragged last tile (keys >= N get -inf; its upper 32 keys are skipped altogether when none is valid).
    // Peeled so the 32 selects per tile that the compiler otherwise if-converts into EVERY iteration stay out of the steady-state loop.
    // A FAST tile whose guard trips returns false BEFORE it has touched l_run / oT or reached its barrier; the caller re-enters the same
    // tile in classic mode with skip_dma (the next tile's DMA is already in flight).  The recomputation lives outside the fast loop so
    // that the loop's register allocation is its own (with it inside, the Q fragments were spilled and their reloads' vmcnt(0) waited
    // for the DMA of the next tile: 1.64 ms instead of 0.95).
    auto tile_step = [&](auto mask_tag, auto fast_tag, int t, bool skip_dma) -> bool {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        const int buf = t & 1;
#ifndef ATTN_ABL_NO_GLOAD
#ifdef ATTN_DMA
        if (t + 1 < nt && !skip_dma) dma(t + 1, buf ^ 1);  // that buffer was last read in tile t-1: every wave is behind its barrier
#else
        if (t + 1 < nt && !skip_dma) gload(t + 1);
#endif
#endif
        if (wave_active) {
            const char* Kb = Ks + buf * KT_BYTES;
            const char* Vb = Vs + buf * KV_TILE_BYTES;
            f32x16 s[2];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool upper = !MASK || (t * 64 + 32 < N);           // wave-uniform: the tile's keys 32..63 hold a valid key
            auto scores = [&](bool with_ref) {
                vec8 kone, qneg;
                if (with_ref) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) kone[j] = qneg[j] = (T)0.f;
                    if (h2 == 0) {
                        kone[0] = (T)1.f;
                        qneg[0] = (T)(-r_ref);           // 16-bit representable by construction
                    }
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (MASK && kb == 1 && !upper) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[1][r] = -INFINITY;
                        continue;
                    }
                    s[kb] = with_ref ? mfma32(kone, qneg, zero16) : zero16;      // -r[q] in every key row
#ifndef ATTN_ABL_NO_QK
#pragma unroll
                    for (int ds = 0; ds < 4; ++ds) {
#ifdef ATTN_DMA
                        const vec8 a = *reinterpret_cast<const vec8*>(Kb + k_off4[ds] + kb * 4096);
#else
                        const vec8 a = *reinterpret_cast<const vec8*>(Kb + k_lane_off + kb * 32 * K_ROW + ds * 32);
#endif
                        s[kb] = mfma32(a, bq[ds], s[kb]);
                    }
#endif
                }
                if constexpr (MASK) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                            if (key >= N) s[kb][r] = -INFINITY;
                        }
                }
            };
            float lsum;
            vec8 pf[4];
            auto exps = [&]() {                          // s <- p = exp2(s), pf = p in 16 bits, lsum = this lane's partial row sum
                lsum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
#ifdef ATTN_ABL_NO_EXP
                        float p = s[kb][r];
                        asm volatile("" : "+v"(p));
#else
                        const float p = __builtin_amdgcn_exp2f(s[kb][r]);
#endif
                        s[kb][r] = p;
#if !defined(ATTN_ABL_NO_SUM) && !defined(ATTN_SUM_DOT2)
                        lsum += p;                       // one chain: four partial sums measured 3 % slower (registers)
#endif
                    }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[ks][j] = (T)s[ks >> 1][(ks & 1) * 8 + j];
#if !defined(ATTN_ABL_NO_SUM) && defined(ATTN_SUM_DOT2)
                // -DATTN_SUM_DOT2: the sum of the probabilities as the P.V MFMA sees them (rounded to 16 bits), two per v_dot2: 16 instructions
                // instead of 32 adds -- measured 1.5 % SLOWER (0.958 vs 0.944 ms), off
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) lsum = sum2(pf[ks][j], pf[ks][j + 1], lsum);
#endif
            };
            auto classic = [&]() {                       // running maximum, reference point, probabilities
#ifdef ATTN_ABL_NO_MAX
                float mx = 0.f;
                asm volatile("" : "+v"(mx));
                if (t < 0) {
#else
                float mx = s[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
#ifdef ATTN_BPERMUTE
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
                {   // the other half of the query's keys sits 32 lanes away: one v_permlane32_swap instead of an LDS round trip
                    // (ds_bpermute + a wait for every outstanding LDS operation) in the middle of every tile
                    const unsigned mu = __float_as_uint(mx);
                    const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
#endif
                if (t == 0 && fast_allowed && __all(fabsf(mx) <= TH0)) {
                    fast = true;                          // the reference point stays 0
                } else if (t == 0 || !__all(mx <= RT)) {   // rare after the first tiles: move the reference point
#endif
                    const bool mv = (t == 0) || (mx > RT);
                    const float r_new = mv ? (float)(T)(r_ref + mx) : r_ref;   // 16-bit representable
                    const float delta = r_new - r_ref;
                    const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // t > 0: the reference only moves up (t = 0: nothing accumulated yet, and 2^-delta may be inf)
                    r_ref = r_new;
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        oT[0][r] *= alpha;
                        oT[1][r] *= alpha;
                        s[0][r] -= delta;
                        s[1][r] -= delta;
                    }
                }
                exps();
            };
            if constexpr (FAST) {
                scores(false);
                exps();
                if (!__all(lsum < GUARD)) return false;  // out of range without a maximum: this tile again, the classic way
            } else {
                scores(true);                            // (after a tripped guard the reference point is still 0)
                classic();
            }
            l_run += lsum;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (MASK && ks >= 2 && !upper) continue;  // p = 0 for all of those keys
#ifdef ATTN_ABL_NO_PV
                asm volatile("" : "+v"(oT[0]), "+v"(oT[1]) : "v"(pf[ks]));
#else
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* p = Vb + ks * 16 * 128 + v_lane_off + (((db * 4 + vchunk) ^ vsw) * 16);
                    const vec8 vf = tr_pair<T>(p, p + 8 * 128);
                    oT[db] = mfma32(vf, pf[ks], oT[db]);
                }
#endif
            }
        }
#ifndef ATTN_ABL_NO_GLOAD
#ifdef ATTN_DMA
#ifndef ATTN_ABL_NO_WAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t+1 have landed
#endif
#else
        if (t + 1 < nt) lstore(buf ^ 1);
#endif
#endif
#ifndef ATTN_ABL_NO_BARRIER
        __syncthreads();
#endif
        return true;
    };
    const int nt_full = (N & 63) ? nt - 1 : nt;
    const std::false_type FULL{}, CLASSIC{};
    const std::true_type RAGGED{}, FASTM{};
    if (nt_full > 0) tile_step(FULL, CLASSIC, 0, false);  // the first tile always computes its maxima: it decides the mode
    else tile_step(RAGGED, CLASSIC, 0, false);
    int t = 1;
    bool resume = false;                                 // the classic loop re-enters a tile whose DMA has been issued
    if (fast) {
        for (; t < nt_full; ++t)
            if (!tile_step(FULL, FASTM, t, false)) { resume = true; break; }
#ifdef ATTN_FAST_RAGGED
        if (!resume && t < nt) {
            if (tile_step(RAGGED, FASTM, t, false)) t = nt;
            else resume = true;
        }
#endif
    }
    for (; t < nt_full; ++t) {
        tile_step(FULL, CLASSIC, t, resume);
        resume = false;
    }
    if (t < nt) tile_step(RAGGED, CLASSIC, t, resume);

    if (wave_active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < N) {
            // row-major: 8-byte pieces, 16 contiguous bytes per row and instruction.  Blocked (include/mst_hip.h, what the single-role
            // block kernel reads): feature f = 64 h + 32 db + 8 g + 4 h2 + e of row R sits in piece f / 16, slot (R % 32) + 32 ((f / 8) & 1),
            // so the 32 query rows of an instruction fill 512 contiguous bytes (two runs where they straddle a 32-row group).
            const int64_t R = (int64_t)seq * N + q;
            T* op = out_blocked ? out + (R >> 5) * (32 * E) + (4 * h) * 512 + (R & 31) * 8 + 4 * h2
                                : out + R * E + h * 64 + 4 * h2;
            const int sdb = out_blocked ? 1024 : 32, sg2 = out_blocked ? 512 : 16, sg1 = out_blocked ? 256 : 8;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __attribute__((ext_vector_type(4))) T o4;
                    o4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (T)(oT[db][g * 4 + e] * inv);
                    *reinterpret_cast<o4*>(op + db * sdb + (g >> 1) * sg2 + (g & 1) * sg1) = pk;
                }
        }
    }
}

// ---- CLS-row probabilities: softmax_j(q_0 . k_j) per (sequence, head) -> probs[seq][h][N] fp32.
template <typename T>
__global__ __launch_bounds__(256) void cls_probs_kernel(const T* __restrict__ qkv, float* __restrict__ probs,
                                                        int N, int heads, int hd, int log2q, T* __restrict__ attn_out = nullptr) {
    extern __shared__ float sm[];  // [hd] q0 | [N] scores | [8] reduce
    float* q0 = sm;
    float* sc = sm + hd;
    float* red = sc + N;
    const int h = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
    const int E = heads * hd, ld = 3 * E;
    const T* base = qkv + (int64_t)seq * N * ld;
    for (int d = tid; d < hd; d += 256) q0[d] = to_f32(base[h * hd + d]);
    __syncthreads();
    float mx = -INFINITY;
    if (hd == 64) {
        // eight lanes per key row: each loads 8 consecutive dims (16 B for the 16-bit types: whole 128-byte K rows per 8 lanes)
        // and the partial dot products meet in three shuffles; 32 keys per pass of the workgroup
        const int sub = tid & 7;
        float qr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[i] = q0[sub * 8 + i];
        for (int j0 = 0; j0 < N; j0 += 32) {
            const int j = j0 + (tid >> 3);
            const T* kp = base + (int64_t)(j < N ? j : N - 1) * ld + E + h * 64 + sub * 8;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) a = fmaf(qr[i], to_f32(kp[i]), a);
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            if (j < N) {
                if (sub == 0) sc[j] = a;
                mx = fmaxf(mx, a);
            }
        }
    } else {
        for (int j = tid; j < N; j += 256) {
            const T* kp = base + (int64_t)j * ld + E + h * hd;
            float a = 0.f;
            for (int d = 0; d < hd; ++d) a = fmaf(q0[d], to_f32(kp[d]), a);
            sc[j] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < N; j += 256) {
        const float p = log2q ? exp2f(sc[j] - mx) : __expf(sc[j] - mx);   // log2q: the scores are already in the log2 domain
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    if (probs) {
        float* po = probs + ((int64_t)seq * heads + h) * N;
        for (int j = tid; j < N; j += 256) po[j] = sc[j] * inv;
    }
    if (attn_out) {   // the CLS query's attention output (hd = 64): o[d] = sum_j p_j v[j][d]; four key ranges x 64 dims, then one add tree
        float* part = red + 8;                             // [4][64]
        const int d = tid & 63, pt = tid >> 6;
        const int per = (N + 3) >> 2;
        const int j1 = (pt + 1) * per < N ? (pt + 1) * per : N;
        float a = 0.f;
        for (int j = pt * per; j < j1; ++j) a = fmaf(sc[j], to_f32(base[(int64_t)j * ld + 2 * E + h * 64 + d]), a);
        part[pt * 64 + d] = a;
        __syncthreads();
        if (tid < 64) attn_out[(int64_t)seq * E + h * 64 + tid] = (T)(((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) * inv);
    }
}

// ---- full probabilities (API parity for the `attention_maps` list / rollout): one wave per query.
template <typename T>
__global__ __launch_bounds__(256) void probs_full_kernel(const T* __restrict__ qkv, float* __restrict__ probs,
                                                         int N, int heads, int hd, int log2q) {
    extern __shared__ float sm[];  // per wave: [hd] q | [N] scores
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* qv = sm + wave * (hd + N);
    float* sc = qv + hd;
    const int h = blockIdx.y, seq = blockIdx.z;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int E = heads * hd, ld = 3 * E;
    const T* base = qkv + (int64_t)seq * N * ld;
    for (int d = lane; d < hd; d += 64) qv[d] = to_f32(base[(int64_t)qi * ld + h * hd + d]);
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
        const T* kp = base + (int64_t)j * ld + E + h * hd;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a = fmaf(qv[d], to_f32(kp[d]), a);
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float p = log2q ? exp2f(sc[j] - mx) : __expf(sc[j] - mx);   // log2q: the scores are already in the log2 domain
        sc[j] = p;
        sum += p;
    }
    const float inv = 1.0f / wave_sum(sum);
    float* po = probs + (((int64_t)seq * heads + h) * N + qi) * N;
    for (int j = lane; j < N; j += 64) po[j] = sc[j] * inv;
}

// ---- full probabilities on the matrix pipe (16-bit inputs, head_dim 64): two passes over the K tiles of a (sequence, head) per
// 128-query workgroup.  Pass 1 is the flash loop without V: S^T = K . Q^T puts a query on a lane, so the running max / sum are
// in-lane.  Pass 2 recomputes the tile as S = Q . K^T (the same two register fragments, operands swapped): now a lane holds ONE
// key of 16 query rows, i.e. 32 lanes write 128 contiguous bytes of a probability row.  The kernel is bound by its 4 N^2-byte
// output (2.9 GB per layer at 64 x 518^2): the one-wave-per-row VALU version took ~140 ms per layer.
template <typename T>
__global__ __launch_bounds__(256) void probs_full16_kernel(const T* __restrict__ qkv, float* __restrict__ probs, int N,
                                                           int heads, int log2q) {
    typedef typename V8<T>::type vec8;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 128 * 8];
    char* const Ks = smem;
    float* const stat = reinterpret_cast<float*>(smem + 2 * K_TILE_BYTES);   // [128 q][2]: max * log2e, 1 / sum
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (N + 127) >> 7;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = tile % nqb, h = (tile / nqb) % heads, seq = tile / (nqb * heads);
    const int E = heads * 64, ld = 3 * E;
    const T* base = qkv + (int64_t)seq * N * ld;
    const int h2 = lane >> 5;
    const int q = qb * 128 + wave * 32 + (lane & 31);
    const float sc = log2q ? 1.0f : LOG2E;               // scores -> log2 domain
    vec8 bq[4];
    {
        const int qc = q < N ? q : N - 1;
        const T* qp = base + (int64_t)qc * ld + h * 64 + h2 * 8;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) bq[ds] = *reinterpret_cast<const vec8*>(qp + ds * 16);
    }
    const int sr = tid >> 2, sc0 = (tid & 3) * 2;        // K staging: thread -> key row sr, chunks sc0, sc0 + 1
    u32x4 rk0, rk1;
    auto gload = [&](int t) {
        int key = t * 64 + sr;
        key = key < N ? key : N - 1;
        const T* kp = base + (int64_t)key * ld + E + h * 64 + sc0 * 8;
        rk0 = *reinterpret_cast<const u32x4*>(kp);
        rk1 = *reinterpret_cast<const u32x4*>(kp + 8);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<u32x4*>(Ks + buf * K_TILE_BYTES + sr * K_ROW + sc0 * 16) = rk0;
        *reinterpret_cast<u32x4*>(Ks + buf * K_TILE_BYTES + sr * K_ROW + sc0 * 16 + 16) = rk1;
    };
    const int k_lane_off = (lane & 31) * K_ROW + h2 * 16;
    const int nt = (N + 63) >> 6;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- pass 1: row statistics
    float m_run = -INFINITY, l_run = 0.f;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        const char* Kb = Ks + buf * K_TILE_BYTES;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = zero16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const vec8 a = *reinterpret_cast<const vec8*>(Kb + k_lane_off + kb * 32 * K_ROW + ds * 32);
                s[kb] = mfma32(a, bq[ds], s[kb]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                s[kb][r] = key < N ? s[kb][r] * sc : -INFINITY;
                mx = fmaxf(mx, s[kb][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        float lsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum += __builtin_amdgcn_exp2f(s[kb][r] - m_new);
        l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + lsum;
        m_run = m_new;
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (h2 == 0) {
            stat[(wave * 32 + (lane & 31)) * 2] = m_run;
            stat[(wave * 32 + (lane & 31)) * 2 + 1] = 1.0f / l_tot;
        }
    }
    __syncthreads();

    // ---- pass 2: S = Q . K^T, lane = key, registers = the wave's 16 query rows (r&3) + 8 (r>>2) + 4 h2 of this half
    float mrow[16], irow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ql = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        mrow[r] = stat[ql * 2];
        irow[r] = stat[ql * 2 + 1];
    }
    float* const pbase = probs + (((int64_t)seq * heads + h) * N) * N;
    gload(0);
    __syncthreads();                                     // every wave is done with pass 1's last tile and with `stat`
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        const char* Kb = Ks + buf * K_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = zero16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const vec8 a = *reinterpret_cast<const vec8*>(Kb + k_lane_off + kb * 32 * K_ROW + ds * 32);
                s = mfma32(bq[ds], a, s);                // operands swapped: D[q][key]
            }
            const int key = t * 64 + kb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qb * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                if (key < N && qq < N) pbase[(int64_t)qq * N + key] = __builtin_amdgcn_exp2f(s[r] * sc - mrow[r]) * irow[r];
            }
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }
}

}  // namespace

int launch_attn16(const void* qkv, int dt, int n_seq, int N, int heads, void* out, int log2q, hipStream_t s, int out_blocked) {
    MST_CHECK_ARG(n_seq > 0 && N > 0 && heads > 0, "attention: bad sizes n_seq=%d N=%d heads=%d", n_seq, N, heads);
    const int64_t nwg = (int64_t)((N + QB - 1) / QB) * heads * n_seq;
    MST_CHECK_ARG(nwg < (1ll << 31), "attention: grid too large");
    const dim3 grid((unsigned)nwg), block(WGT);
    if (dt == MST_BF16) attn16_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)qkv, (bf16_t*)out, N, heads, log2q, out_blocked);
    else if (dt == MST_F16) attn16_kernel<f16_t><<<grid, block, 0, s>>>((const f16_t*)qkv, (f16_t*)out, N, heads, log2q, out_blocked);
    else { mst_set_error("attention16: bad dtype %d", dt); return MST_EINVAL; }
    return mst_check_launch("attention16");
}

int launch_cls_probs(const void* qkv, int dt, int n_seq, int N, int heads, int hd, float* probs, int log2q, hipStream_t s) {
    MST_CHECK_ARG(N > 0 && N <= 12000 && hd > 0 && hd <= 256, "cls_probs: N=%d hd=%d unsupported", N, hd);
    if (n_seq > 65535) {   // gridDim.y limit: walk the sequences in pieces
        const size_t esz = dt == MST_F32 ? 4 : 2;
        for (int s0 = 0; s0 < n_seq; s0 += 65535) {
            const int c = (n_seq - s0 < 65535) ? n_seq - s0 : 65535;
            int rc = launch_cls_probs((const char*)qkv + (size_t)s0 * N * 3 * heads * hd * esz, dt, c, N, heads, hd,
                                      probs + (size_t)s0 * heads * N, log2q, s);
            if (rc) return rc;
        }
        return MST_OK;
    }
    const dim3 grid(heads, n_seq), block(256);
    const size_t sh = (size_t)(hd + N + 8 + 256) * sizeof(float);
    if (dt == MST_BF16) cls_probs_kernel<bf16_t><<<grid, block, sh, s>>>((const bf16_t*)qkv, probs, N, heads, hd, log2q);
    else if (dt == MST_F16) cls_probs_kernel<f16_t><<<grid, block, sh, s>>>((const f16_t*)qkv, probs, N, heads, hd, log2q);
    else if (dt == MST_F32) cls_probs_kernel<float><<<grid, block, sh, s>>>((const float*)qkv, probs, N, heads, hd, log2q);
    else { mst_set_error("cls_probs: bad dtype %d", dt); return MST_EINVAL; }
    return mst_check_launch("cls_probs");
}

// The CLS query's attention output only ([n_seq, heads*64] in the operand type; optionally its probabilities too): what the LAST
// block needs when nothing but the class token is read behind it (mst_vit_weights.prune_last_block).
int launch_cls_attn(const void* qkv, int dt, int n_seq, int N, int heads, float* probs, void* out, int log2q, hipStream_t s) {
    MST_CHECK_ARG(N > 0 && N <= 12000 && n_seq > 0 && n_seq <= 65535 && out, "cls_attn: N=%d n_seq=%d unsupported", N, n_seq);
    const dim3 grid(heads, n_seq), block(256);
    const size_t sh = (size_t)(64 + N + 8 + 256) * sizeof(float);
    if (dt == MST_BF16) cls_probs_kernel<bf16_t><<<grid, block, sh, s>>>((const bf16_t*)qkv, probs, N, heads, 64, log2q, (bf16_t*)out);
    else if (dt == MST_F16) cls_probs_kernel<f16_t><<<grid, block, sh, s>>>((const f16_t*)qkv, probs, N, heads, 64, log2q, (f16_t*)out);
    else if (dt == MST_F32) cls_probs_kernel<float><<<grid, block, sh, s>>>((const float*)qkv, probs, N, heads, 64, log2q, (float*)out);
    else { mst_set_error("cls_attn: bad dtype %d", dt); return MST_EINVAL; }
    return mst_check_launch("cls_attn");
}

int launch_probs_full(const void* qkv, int dt, int n_seq, int N, int heads, int hd, float* probs, int log2q, hipStream_t s) {
    if (hd == 64 && (dt == MST_BF16 || dt == MST_F16)) {   // matrix-pipe path
        MST_CHECK_ARG(N > 0 && n_seq > 0 && heads > 0, "probs_full: bad sizes");
        const int64_t nwg = (int64_t)((N + 127) / 128) * heads * n_seq;
        MST_CHECK_ARG(nwg < (1ll << 31), "probs_full: grid too large");
        if (dt == MST_BF16) probs_full16_kernel<bf16_t><<<dim3((unsigned)nwg), dim3(256), 0, s>>>((const bf16_t*)qkv, probs, N, heads, log2q);
        else probs_full16_kernel<f16_t><<<dim3((unsigned)nwg), dim3(256), 0, s>>>((const f16_t*)qkv, probs, N, heads, log2q);
        return mst_check_launch("probs_full16");
    }
    MST_CHECK_ARG(N > 0 && N <= 3800 && hd > 0 && hd <= 256, "probs_full: N=%d hd=%d unsupported", N, hd);
    if (n_seq > 65535) {   // gridDim.z limit: walk the sequences in pieces
        const size_t esz = dt == MST_F32 ? 4 : 2;
        for (int s0 = 0; s0 < n_seq; s0 += 65535) {
            const int c = (n_seq - s0 < 65535) ? n_seq - s0 : 65535;
            int rc = launch_probs_full((const char*)qkv + (size_t)s0 * N * 3 * heads * hd * esz, dt, c, N, heads, hd,
                                       probs + (size_t)s0 * heads * N * N, log2q, s);
            if (rc) return rc;
        }
        return MST_OK;
    }
    const dim3 grid((N + 3) / 4, heads, n_seq), block(256);
    const size_t sh = (size_t)4 * (hd + N) * sizeof(float);
    if (dt == MST_BF16) probs_full_kernel<bf16_t><<<grid, block, sh, s>>>((const bf16_t*)qkv, probs, N, heads, hd, log2q);
    else if (dt == MST_F16) probs_full_kernel<f16_t><<<grid, block, sh, s>>>((const f16_t*)qkv, probs, N, heads, hd, log2q);
    else if (dt == MST_F32) probs_full_kernel<float><<<grid, block, sh, s>>>((const float*)qkv, probs, N, heads, hd, log2q);
    else { mst_set_error("probs_full: bad dtype %d", dt); return MST_EINVAL; }
    return mst_check_launch("probs_full");
}
