// Everything of a ViT block after the attention kernel, for E = 384 and 16-bit MFMA operands, in ONE launch -- the
// SINGLE-ROLE form (round 3; k_block16.hip is the producer/consumer form it replaces):
//     x  <- x + ls1 * (proj(attn_out) + b_proj)                       attention.py:67-68; block.py:90-91,112
//     x  <- x + ls2 * (fc2(gelu(fc1(LayerNorm2(x)))) + b2)            block.py:93-94,113; mlp.py:34-40
//     xn <- normalise(x)      (optional: the NEXT block's norm1, its affine folded into that block's QKV weights)
//
// Structure: persistent, one 4-wave workgroup per CU = ONE wave per SIMD with the whole 512-register file, 128 token rows per
// tile, every wave owns 32 rows end to end.  Everything a row needs stays in the registers of the lane pair (l, l+32) that owns it:
//   * y^T accumulators [384 features x 32 rows] = 12 tiles of the 32x32x16 MFMA (192 registers); the residual x rides inside;
//   * the out-projection's B operand (attention-output rows) is loaded from global memory straight into fragment registers;
//   * LayerNorm2 runs on the accumulators and leaves the normalised rows as 24 B fragments (96 registers): a 32x32 accumulator
//     tile IS the next MFMA's B operand after a pairwise 16-bit pack (k order permuted; the weight images are packed to match);
//   * per hidden chunk of 32 units: GEMM1 (24 MFMAs) -> GELU in registers -> the same lane-private pack -> GEMM2 (24 MFMAs).
// No hand-off of activations through LDS or global memory exists at all (k_block16.hip: a 16 KiB LDS hand-off per chunk and a
// 96 KiB global scratch per tile), and no wave waits for another one except at the ring barrier.
//
// Weights: every 24 KiB weight image (12 out-projection chunks, 48 W1 chunks, 48 W2 chunks) is ONE element of a single stream
// packed on the host in consumption order (108 elements per tile, [24 fragments][64 lanes][16 B] each: a fragment read is a
// conflict-free ds_read_b128 at lane*16).  The stream runs through a 6-slot LDS ring by LDS-DMA four elements ahead; a "phase"
// consumes one element in 24 slots of [fragment read 6 ahead | counted lgkmcnt | MFMA | a few vector instructions of the GELU
// of a neighbouring chunk], with one s_barrier per phase.  Fragment reads run across the phase boundaries (the barrier of phase e
// also publishes element e + 1), so the LDS latency never surfaces.
#include <type_traits>

#include "mst_common.h"

namespace {

constexpr int E = 384, HID = 1536, CH = 32;
constexpr int NCHUNK = HID / CH;                        // 48
constexpr int PJ = E / CH;                              // 12 out-projection phases
constexpr int ELEM_BYTES = 24 * 1024;                   // one ring element: 24 fragments of 1 KiB
constexpr int ELEMS = PJ + 2 * NCHUNK;                  // 108 per tile
constexpr int NSLOT = 6;
constexpr int AHEAD = 4;                                // phase e issues the LDS-DMA of element e + 4
constexpr int RING_BYTES = NSLOT * ELEM_BYTES;          // 147,456
constexpr int B1_OFF = RING_BYTES;                      // fp32 b1f [1536]
constexpr int BP_OFF = B1_OFF + HID * 4;                // fp32 b_proj [384]  (ls1 folded)
constexpr int B2_OFF = BP_OFF + E * 4;                  // fp32 b2     [384]  (ls2 folded)
constexpr int LDS_BYTES = B2_OFF + E * 4;               // 156,672
constexpr int NF = 24;                                  // fragments (= MFMAs) per phase
#ifndef BLOCKS_DEPTH
#define BLOCKS_DEPTH 4
#endif
constexpr int D = BLOCKS_DEPTH;                         // fragment reads in flight ahead of the MFMAs
// Register sets of the fragment ring: D + 2, not D + 1.  The read issued in slot i must not target the set MFMA i-1 took its A
// operand from: that MFMA is still executing, and the in-order wave then stalls AT THE READ until it has finished (write-after-read
// on its source registers) -- measured 44 instead of 32 cycles per slot with D + 1 sets (profiles/r04b_*).
constexpr int R = D + 2;
static_assert(NF % R == 0, "fragment ring must tile the phase");
// Slot i of an out-projection / GEMM2 phase takes fragment FRAG(i) = (tile t = i % 12, k-step p = i / 12): consecutive MFMAs go to
// DIFFERENT accumulator tiles.  A dependent 32x32x16 MFMA (same accumulator as its predecessor) issued 48 cycles after it, an
// independent one 32 (stamps: 48.5 cycles per slot in the GEMM1 chain, 44 with dependent pairs; profiles/r04b_*).
#ifdef BLOCKS_PAIRS
constexpr int frag_of(int i) { return i; }
#else
constexpr int frag_of(int i) { return 2 * (i % 12) + i / 12; }
#endif
constexpr int BIAS_SLOT = 8;                            // slot of a phase in which the next chunk's b1 is read

template <int OFF, typename V> __device__ __forceinline__ void lds_read_b128(V& dst, unsigned addr) {
#if defined(BLOCKS_WACC)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"(OFF));
#elif !defined(BLOCKS_NOREADS)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
#else
    asm volatile("" : "=v"(dst) : "v"(addr));            // timing-only ablation: fragments are whatever the registers hold
#endif
}
template <int OFF, typename V> __device__ __forceinline__ void lds_read_b128_acc(V& dst, unsigned addr) {   // into the accumulator file
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {       // + fence: no MFMA above the wait (rule 18)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
#ifdef BLOCKS_STAMPS
// diagnostic build only (never shipped): per-wave cycle sums by phase kind, read back by mst_debug_blocks_stamps
__device__ unsigned long long g_bsstamps[256 * 4 * 16];
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

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x16 cat16(f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    const f32x8 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
template <int Q> __device__ __forceinline__ f32x4 sub4(f32x16 v) { return __builtin_shufflevector(v, v, 4 * Q, 4 * Q + 1, 4 * Q + 2, 4 * Q + 3); }

// ---- GELU as a list of micro-operations (one VALU / transcendental instruction each), so that a phase can spread the GELU of
// eight values over its MFMA slots.  Value v, scratch a, b; after the last stage `a` holds gelu(v).  Forms: mst_common.h gelu_sig.
template <typename T> struct Gelu;
template <> struct Gelu<bf16_t> {
    static constexpr int STAGES = 7;
    static constexpr int target(int) { return 0; }       // scratch register a stage writes (0 = a, 1 = b, 2 = c)
    template <int S> static __device__ __forceinline__ void stage(float v, float& a, float& b, float& c) {
        if constexpr (S == 0) a = v * v;
        else if constexpr (S == 1) a = fmaf(a, -0.06940179f * 1.4426950408889634f, -1.60031416f * 1.4426950408889634f);
        else if constexpr (S == 2) a = v * a;
        else if constexpr (S == 3) a = __builtin_amdgcn_exp2f(a);
        else if constexpr (S == 4) a = 1.0f + a;
        else if constexpr (S == 5) a = __builtin_amdgcn_rcpf(a);
        else a = v * a;
    }
};
template <> struct Gelu<f16_t> {
    static constexpr int STAGES = 9;
    static constexpr int target(int S) { return S == 0 ? 1 : (S == 2 || S == 3) ? 2 : 0; }
    template <int S> static __device__ __forceinline__ void stage(float v, float& a, float& b, float& c) {
        if constexpr (S == 0) b = __builtin_amdgcn_fmed3f(v, -8.0f, 8.0f);       // the quadratic P turns over at |v| = 8.35
        else if constexpr (S == 1) a = b * b;
        else if constexpr (S == 2) c = fmaf(a, 7.03033577e-04f * 1.4426950408889634f, -7.40112920e-02f * 1.4426950408889634f);
        else if constexpr (S == 3) c = fmaf(c, a, -1.59501577f * 1.4426950408889634f);
        else if constexpr (S == 4) a = b * c;
        else if constexpr (S == 5) a = __builtin_amdgcn_exp2f(a);
        else if constexpr (S == 6) a = 1.0f + a;
        else if constexpr (S == 7) a = __builtin_amdgcn_rcpf(a);
        else a = v * a;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void block16s_kernel(float* x, const T* attn, T* xn_out, const char* __restrict__ wseq,
                                                       const float* __restrict__ b1f, const float* __restrict__ bproj,
                                                       const float* __restrict__ b2, int M, int ntiles, float eps, int layout) {
    typedef typename V8<T>::type vec8;
    typedef typename V8<T>::half_type vec4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool x_in_img = layout & MST_LAYOUT_X_IN_IMAGE, x_out_img = layout & MST_LAYOUT_X_OUT_IMAGE, act_blk = layout & MST_LAYOUT_ACT_BLOCKED;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row32 = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane16 = lane * 16;

    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    // ---- biases -> LDS (once per workgroup)
    for (int i = tid; i < HID / 4; i += 256) *reinterpret_cast<f32x4*>(smem + B1_OFF + i * 16) = *reinterpret_cast<const f32x4*>(b1f + 4 * i);
    for (int i = tid; i < E / 4; i += 256) {
        *reinterpret_cast<f32x4*>(smem + BP_OFF + i * 16) = *reinterpret_cast<const f32x4*>(bproj + 4 * i);
        *reinterpret_cast<f32x4*>(smem + B2_OFF + i * 16) = *reinterpret_cast<const f32x4*>(b2 + 4 * i);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (the first phase barrier publishes them)

    // ---- weight stream.  Element ge (global index over this workgroup's tiles) = stream element ge % 108 -> ring slot ge % 6;
    // this wave's share: the six consecutive 1 KiB pieces 6 wave .. 6 wave + 5 (one lane address, one M0, six immediates).
    // The stream simply wraps: the four elements issued beyond the last tile land in slots nobody reads (drained before exit).
#ifndef BLOCKS_DMA_GLOBAL
    // buffer form: the piece base rides in soffset (SGPR), the lane part is ONE 32-bit VGPR: half the address traffic per issue
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wseq, 0, ELEMS * ELEM_BYTES, 0x00020000);
    const int wlane_off = wave * 6144 + lane16;
    char* const wdst_wave = smem + wave * 6144;
    auto dma_piece = [&](int src_off, int slot, auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
#ifndef BLOCKS_NODMA
        // (the 12-bit immediate moves the source AND the LDS address: pieces 4 and 5 take a second base 4 KiB further on both sides)
        constexpr int hi = u >= 4 ? 4096 : 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(wdst_wave + slot * ELEM_BYTES + hi), 16, wlane_off, src_off + hi, u * 1024 - hi, 0);
#endif
    };
#else
    const char* const wsrc_lane = wseq + (wave * 6144 + 2048) + lane16;
    char* const wdst_wave = smem + wave * 6144 + 2048;
    auto dma_piece = [&](int src_off, int slot, auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
#ifndef BLOCKS_NODMA
        __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc_lane + src_off), LDS_PTR(wdst_wave + slot * ELEM_BYTES), 16, (u - 2) * 1024, 0);
#endif
    };
#endif
    for (int e0 = 0; e0 < AHEAD; ++e0) static_for<0, 6>([&](auto u) { dma_piece(e0 * ELEM_BYTES, e0, u); });
    int dsrc = AHEAD * ELEM_BYTES, dslot = AHEAD;         // stream byte offset / ring slot of the next element to request

    f32x16 acc[12];
    u32x4 xa[24];                                        // out-projection: attention-output fragments; MLP: normalised rows
    f32x4 hq[2][4];                                      // hidden pre-activations of two chunks in flight, as register quarters
    vec8 hB[2][2];                                       // [chunk parity][k-step]: GELU outputs as GEMM2 B fragments
    vec8 w[R];                                           // weight fragment ring
    float ga[8], gb[8], gc[8];                           // GELU scratch of the eight values in flight
#ifdef BLOCKS_NOFILL
    hB[0][0] = hB[0][1] = hB[1][0] = hB[1][1] = vec8{};
#endif

    int slot = 0;                                        // ring slot of the element the next phase consumes
    const unsigned b1_lane = lds_base + B1_OFF + half * 16;     // + 128 * chunk + 32 * q

    // ---- one phase: 24 slots over ring element `slot` (prefetching the head of the next element), MFMA `mf(i, fragment)`,
    // vector filler `fill(i)`; BQ >= 0: b1 of chunk `bias_chunk` is read INTO hq[BQ] (dead at that point), where the GEMM1 of that
    // chunk accumulates on top of it a phase later.
#ifdef BLOCKS_STAMPS
    unsigned long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = bstamp();
#endif
    // ord_tag / nord_tag: 1 = this / the next phase walks its fragments tile-major (frag_of), 0 = in stream order (GEMM1 k-steps)
    auto phase = [&](auto head_tag, auto tail_tag, auto bq_tag, int bias_chunk, auto&& mf, auto&& fill, auto cat_tag, auto ord_tag, auto nord_tag) {
        constexpr bool HEAD = decltype(head_tag)::value, TAIL = decltype(tail_tag)::value;
        constexpr int ORD = decltype(ord_tag)::value, NORD = decltype(nord_tag)::value;
        auto fr = [](int i, int ord) constexpr { return ord ? frag_of(i) : i; };
        constexpr int BQ = decltype(bq_tag)::value, CAT = decltype(cat_tag)::value;
        BST(p0);
        // element ge landed for everybody at the previous barrier; this one publishes ge + 1: own pieces first (all but the 12
        // youngest vector-memory operations done: the pieces of ge + 2 and ge + 3 may stay in flight)
        asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        BST(p1);
        BACC(CAT, p0, p1);
        const unsigned cur = lds_base + slot * ELEM_BYTES + lane16;
        const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
        const unsigned nxt = lds_base + nslot * ELEM_BYTES + lane16;
        const int my_dsrc = dsrc, my_dslot = dslot;
        dsrc = dsrc + ELEM_BYTES == ELEMS * ELEM_BYTES ? 0 : dsrc + ELEM_BYTES;
        dslot = dslot + 1 == NSLOT ? 0 : dslot + 1;
        const unsigned baddr = b1_lane + 128 * bias_chunk;
        f32x4 bt0, bt1, bt2, bt3;
        if constexpr (HEAD) static_for<0, D>([&](auto q) { lds_read_b128<fr(decltype(q)::value, ORD) * 1024>(w[decltype(q)::value % R], cur); });
        static_for<0, NF>([&](auto it) {
            constexpr int i = decltype(it)::value;
            if constexpr (i + D < NF) lds_read_b128<fr(i + D, ORD) * 1024>(w[(i + D) % R], cur);
            else if constexpr (TAIL) lds_read_b128<fr(i + D - NF, NORD) * 1024>(w[(i + D) % R], nxt);
            if constexpr (BQ >= 0 && i == BIAS_SLOT) {
                lds_read_b128_acc<0>(bt0, baddr);
                lds_read_b128_acc<32>(bt1, baddr);
                lds_read_b128_acc<64>(bt2, baddr);
                lds_read_b128_acc<96>(bt3, baddr);
            }
#ifndef BLOCKS_WAIT1                                       // default since r04j: -1 % kernel time (1.074 -> 1.062 ms in the bench pipeline)
            // ONE counted wait per two slots: at an even slot fragments i and i + 1 have landed once only the reads behind
            // fragment i + 1 are outstanding (fragments i + 2 .. last issued, plus the four b1 reads where they are younger)
            if constexpr (i % 2 == 0) {
                constexpr int last = TAIL ? i + D : (i + D < NF - 1 ? i + D : NF - 1);
                constexpr int younger_frags = last - (i + 1) > 0 ? last - (i + 1) : 0;
                constexpr int younger_bias = (BQ >= 0 && i >= BIAS_SLOT && i + 1 <= BIAS_SLOT + D) ? 4 : 0;
                wait_lgkm<younger_frags + younger_bias>();
            } else {
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            constexpr int younger_frags = TAIL ? D : (NF - 1 - i < D ? NF - 1 - i : D);
            constexpr int younger_bias = (BQ >= 0 && i >= BIAS_SLOT && i <= BIAS_SLOT + D) ? 4 : 0;
            wait_lgkm<younger_frags + younger_bias>();
#endif
            if constexpr (BQ >= 0 && i == BIAS_SLOT + D + 1) {
                // this slot's counted wait covers the four b1 reads: only now do the values exist for hipcc (an asm output it may
                // copy at once -- it moved the in-flight registers into the accumulator file right behind the reads otherwise)
                asm volatile("" : "+a"(bt0), "+a"(bt1), "+a"(bt2), "+a"(bt3));
                hq[BQ < 0 ? 0 : BQ][0] = bt0;
                hq[BQ < 0 ? 0 : BQ][1] = bt1;
                hq[BQ < 0 ? 0 : BQ][2] = bt2;
                hq[BQ < 0 ? 0 : BQ][3] = bt3;
            }
            mf(it, w[i % R]);
#ifdef BLOCKS_DMA_FRONT
            if constexpr (i == 0) static_for<0, 6>([&](auto u) { dma_piece(my_dsrc, my_dslot, u); });
#else
            if constexpr (i % 4 == 3) dma_piece(my_dsrc, my_dslot, std::integral_constant<int, i / 4>{});
#endif
            fill(it);
        });
        slot = nslot;
        BST(p2);
        BACC(CAT + 1, p1, p2);
    };
    constexpr std::integral_constant<int, 0> C_PROJ{};
    constexpr std::integral_constant<int, 2> C_A{};
    constexpr std::integral_constant<int, 4> C_B{};
    constexpr std::integral_constant<int, 6> C_G10{};
    auto no_fill = [](auto) {};
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    constexpr std::integral_constant<int, -1> NOBIAS{};
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};

    // GEMM1 of one chunk into hq[HS] (which holds the chunk's b1): 24 dependent MFMAs on one 32x32 accumulator
    auto gemm1_mf = [&](auto hs_tag) {
        return [&](auto it, const vec8& wf) {
            constexpr int hs = decltype(hs_tag)::value, i = decltype(it)::value;
            f32x16 c = cat16(hq[hs][0], hq[hs][1], hq[hs][2], hq[hs][3]);
            c = mfma32(wf, __builtin_bit_cast(vec8, xa[i]), c);
            static_for<0, 4>([&](auto q) { hq[hs][decltype(q)::value] = sub4<decltype(q)::value>(c); });
        };
    };
    auto gemm2_mf = [&](auto hs_tag) {
        return [&](auto it, const vec8& wf) {
            constexpr int hs = decltype(hs_tag)::value, i = frag_of(decltype(it)::value);
            acc[i >> 1] = mfma32(wf, hB[hs][i & 1], acc[i >> 1]);
        };
    };

    // ---- GELU of eight values (registers 8 PART .. 8 PART + 7 of chunk buffer hs) -> hB[hs][PART], as micro-operations
    constexpr int GOPS = 8 * Gelu<T>::STAGES + 4;        // + 4 pairwise packs
    auto gelu_op = [&](auto hs_tag, auto part_tag, auto n_tag) {
        constexpr int hs = decltype(hs_tag)::value, PART = decltype(part_tag)::value, n = decltype(n_tag)::value;
        if constexpr (n < 8 * Gelu<T>::STAGES) {
            constexpr int grp = n / (4 * Gelu<T>::STAGES), m = n % (4 * Gelu<T>::STAGES);
            constexpr int st = m / 4, val = 4 * grp + m % 4;
            Gelu<T>::template stage<st>(hq[hs][2 * PART + (val >> 2)][val & 3], ga[val], gb[val], gc[val]);
            // pinned: without an ordered user hipcc gathers the whole list at the head of the phase
            if constexpr (Gelu<T>::target(st) == 0) asm volatile("" : "+v"(ga[val]));
            else if constexpr (Gelu<T>::target(st) == 1) asm volatile("" : "+v"(gb[val]));
            else asm volatile("" : "+v"(gc[val]));
        } else if constexpr (n < GOPS) {
            constexpr int pr = n - 8 * Gelu<T>::STAGES;  // values 2 pr, 2 pr + 1 -> elements 2 pr, 2 pr + 1 of the fragment
            hB[hs][PART][2 * pr] = (T)ga[2 * pr];
            hB[hs][PART][2 * pr + 1] = (T)ga[2 * pr + 1];
        }
    };
    constexpr int OPS_PER_SLOT = (GOPS + NF - 2) / (NF - 1);          // slots 1..23 carry the filler
    auto gelu_fill = [&](auto hs_tag, auto part_tag) {
        return [&, hs_tag, part_tag](auto it) {
            constexpr int i = decltype(it)::value;
#ifdef BLOCKS_NOFILL
            // timing-only ablation: GELU = identity (the packs stay, so GEMM1 stays live)
            if constexpr (i == 12) static_for<0, 8>([&](auto e) { hB[decltype(hs_tag)::value][decltype(part_tag)::value][decltype(e)::value] = (T)hq[decltype(hs_tag)::value][2 * decltype(part_tag)::value + (decltype(e)::value >> 2)][decltype(e)::value & 3]; });
            if constexpr (false)
#else
            if constexpr (i >= 1)
#endif
                static_for<0, OPS_PER_SLOT>([&](auto o) { gelu_op(hs_tag, part_tag, std::integral_constant<int, (i - 1) * OPS_PER_SLOT + decltype(o)::value>{}); });
        };
    };
    auto gelu_now = [&](auto hs_tag, auto part_tag) { static_for<0, GOPS>([&](auto n) { gelu_op(hs_tag, part_tag, n); }); };

    // ---- row traffic.  Blocked / image layouts (include/mst_hip.h): every instruction moves one contiguous KiB; row-major: lane =
    // row, 32 scattered 32-byte runs per instruction (what the first and the last block of an encoder still see).
    // The rows of tile k + 1 are requested from inside the epilogue of tile k: the attention fragments as soon as xa is dead, the x
    // pieces tile by tile right behind the stores that free their accumulator registers, so the loads queue behind nothing
    // (stamps: 14-17 k cycles per tile spent ISSUING 72 loads behind the 72 stores of the epilogue when they came afterwards).
    const int last_grp = (M - 1) >> 5;
    auto row_of = [&](int tile) { const int g = tile * 128 + wave * 32 + row32; return (unsigned)(g < M ? g : M - 1); };
    auto grp_of = [&](int tile) { return (size_t)((tile * 4 + wave) < last_grp ? (tile * 4 + wave) : last_grp); };
    auto load_attn = [&](int tile) {
        if (act_blk) {
            const char* ap = (const char*)attn + (grp_of(tile) * (32 * E * 2) + lane16);
#pragma unroll
            for (int i = 0; i < 24; ++i) xa[i] = *reinterpret_cast<const u32x4*>(ap + 1024 * i);
        } else {
            const char* ap = (const char*)attn + ((size_t)row_of(tile) * (E * 2) + half * 16);
#pragma unroll
            for (int i = 0; i < 24; ++i) xa[i] = *reinterpret_cast<const u32x4*>(ap + 32 * i);
        }
    };
    auto load_x = [&](int tile, auto t_tag) {            // accumulator tile t of the wave's 32 rows
        constexpr int t = decltype(t_tag)::value;
        const char* xp = x_in_img ? (const char*)x + (grp_of(tile) * (32 * E * 4) + lane16) + 4096 * t
                                  : (const char*)x + ((size_t)row_of(tile) * (E * 4) + half * 16) + 128 * t;
        const int qstep = x_in_img ? 1024 : 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + qstep * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][4 * q + r] = v[r];
        }
    };
    // De-phase the persistent workgroups (k_mlp16.hip): every workgroup runs the same program on the same amount of work, so all 256
    // tile boundaries (576 KB of row traffic per workgroup) would hit HBM in the same ~10 % of the tile period.  Workgroups with
    // the smaller tile count start up to a whole tile late for free, the others up to half a tile.
#ifndef BLOCKS_TILE_CYCLES
#define BLOCKS_TILE_CYCLES 170000
#endif
#ifndef BLOCKS_NO_DEPHASE
    if (my_tiles >= 4) {
        const int min_tiles = ntiles / (int)gridDim.x;
        const unsigned u = (((unsigned)blockIdx.x >> 3) + 5u * ((unsigned)blockIdx.x & 7u)) & 15u;   // 0..15
        const unsigned long long delay = (unsigned long long)BLOCKS_TILE_CYCLES * ((my_tiles > min_tiles ? 0u : 16u) + u) / 32u;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 640 && __builtin_readcyclecounter() - t0 < delay; ++it) __builtin_amdgcn_s_sleep(32);   // bounded: every wave leaves
    }
#endif
    load_attn(blockIdx.x);
    static_for<0, 12>([&](auto t) { load_x(blockIdx.x, t); });

    for (int k = 0; k < my_tiles; ++k) {
        const int tile = blockIdx.x + k * gridDim.x;
        const int next_tile = k + 1 < my_tiles ? tile + (int)gridDim.x : tile;   // (last tile: its own rows once more, unused)
        const int grow = tile * 128 + wave * 32 + row32;                 // this lane's token row
        const bool valid = grow < M;
        const unsigned crow = (unsigned)(valid ? grow : M - 1);
        BST(q0);
        BST(q1);
        BACC(8, q0, q1);
        // ---- out-projection: 12 phases, acc[t] += Wp chunk j (t, p) . attention columns
        static_for<0, PJ>([&](auto jt) {
            constexpr int j = decltype(jt)::value;
            auto mf = [&](auto it, const vec8& wf) {
                constexpr int i = frag_of(decltype(it)::value);
                acc[i >> 1] = mfma32(wf, __builtin_bit_cast(vec8, xa[2 * j + (i & 1)]), acc[i >> 1]);
            };
            if constexpr (j == 0) phase(YES, YES, NOBIAS, 0, mf, no_fill, C_PROJ, I1, I1);
            else if constexpr (j == PJ - 1) phase(NO, NO, I0, 0, mf, no_fill, C_PROJ, I1, I1);           // b1 of chunk 0 -> hq[0]; no prefetch across LayerNorm2
            else phase(NO, YES, NOBIAS, 0, mf, no_fill, C_PROJ, I1, I1);
        });
        BST(q2);
        // ---- LayerNorm2 on the accumulators (+ b_proj first), normalised rows -> xa, then + b2.  Vector-typed arithmetic on
        // purpose (v_pk_* at the boundary, where no MFMA competes for the issue port); the file is built with -fno-slp-vectorize.
        {
            // (no fragment read is in flight here: asm outputs that have not landed must not live across code hipcc schedules by
            // itself -- under LayerNorm2's register pressure it copied them into the accumulator file right behind the reads)
            const unsigned bp = lds_base + BP_OFF + half * 16, bb = lds_base + B2_OFF + half * 16;
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            static_for<0, 12>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                f32x4 b[4];
                static_for<0, 4>([&](auto q) { lds_read_b128<128 * t + 32 * decltype(q)::value>(b[decltype(q)::value], bp); });
                wait_lgkm<0>();
                acc[t] += cat16(b[0], b[1], b[2], b[3]);
                s4 += (sub4<0>(acc[t]) + sub4<1>(acc[t])) + (sub4<2>(acc[t]) + sub4<3>(acc[t]));
                asm volatile("" : "+v"(s4));
            });
            float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / E);
            const f32x4 mean4 = {mean, mean, mean, mean};
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
            static_for<0, 12>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                static_for<0, 4>([&](auto q) {
                    const f32x4 d = sub4<decltype(q)::value>(acc[t]) - mean4;
                    q4 = __builtin_elementwise_fma(d, d, q4);
                });
                asm volatile("" : "+v"(q4));
            });
            float sq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / E) + eps);
            const float nmr = -mean * rstd;
            const f32x4 rstd4 = {rstd, rstd, rstd, rstd}, nmr4 = {nmr, nmr, nmr, nmr};
            static_for<0, 12>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                f32x4 b[4];
                static_for<0, 4>([&](auto q) { lds_read_b128<128 * t + 32 * decltype(q)::value>(b[decltype(q)::value], bb); });
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x4 n0 = __builtin_elementwise_fma(p ? sub4<2>(acc[t]) : sub4<0>(acc[t]), rstd4, nmr4);
                    const f32x4 n1 = __builtin_elementwise_fma(p ? sub4<3>(acc[t]) : sub4<1>(acc[t]), rstd4, nmr4);
                    vec8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (T)n0[e]; o[4 + e] = (T)n1[e]; }
                    xa[2 * t + p] = __builtin_bit_cast(u32x4, o);
                    asm volatile("" : "+v"(xa[2 * t + p]));
                }
                wait_lgkm<0>();
                acc[t] += cat16(b[0], b[1], b[2], b[3]);
            });
        }
        BST(q3);
        BACC(9, q2, q3);
        // ---- GEMM1 of chunk 0 (b1 of chunk 1 -> hq[1]), then the first half of its GELU
        phase(YES, YES, I1, 1, gemm1_mf(I0), no_fill, C_G10, I0, I0);
        wait_lgkm<0>();
        gelu_now(I0, I0);
        // ---- chunks: A(c) = GEMM1(c + 1) beside the second half of GELU(c); B(c) = GEMM2(c) beside the first half of GELU(c + 1)
        // (+ b1 of chunk c + 2 into the buffer GELU(c) has just left)
        auto iter = [&](auto par_tag, auto bias_tag, int c) {
            constexpr int cur = decltype(par_tag)::value, nx = cur ^ 1;
            phase(NO, YES, NOBIAS, 0, gemm1_mf(std::integral_constant<int, nx>{}), gelu_fill(std::integral_constant<int, cur>{}, I1), C_A, I0, I1);
            if constexpr (decltype(bias_tag)::value)
                phase(NO, YES, std::integral_constant<int, cur>{}, c + 2, gemm2_mf(std::integral_constant<int, cur>{}), gelu_fill(std::integral_constant<int, nx>{}, I0), C_B, I1, I0);
            else                                         // chunk 46: the next phase is GEMM2(47)
                phase(NO, YES, NOBIAS, 0, gemm2_mf(std::integral_constant<int, cur>{}), gelu_fill(std::integral_constant<int, nx>{}, I0), C_B, I1, I1);
        };
#pragma unroll 1
        for (int c = 0; c < NCHUNK - 2; c += 2) {        // chunks 0 .. 45
            iter(I0, YES, c);
            iter(I1, YES, c + 1);
        }
        iter(I0, NO, NCHUNK - 2);                        // chunk 46
        wait_lgkm<0>();
        gelu_now(I1, I1);
        phase(NO, NO, NOBIAS, 0, gemm2_mf(I1), no_fill, C_B, I1, I1); // GEMM2 of chunk 47
        // ---- the block's output rows: x and the next block's normalised rows, interleaved with the requests for the next tile's rows
        BST(q4);
        {
            const size_t grp = (size_t)tile * 4 + wave;
            const bool grp_valid = (tile * 4 + wave) * 32 < M;           // wave-uniform: groups past M do not exist in the buffers
            load_attn(next_tile);                                        // xa is dead since the last GEMM1
            f32x4 rstd4 = {0.f, 0.f, 0.f, 0.f}, nmr4 = {0.f, 0.f, 0.f, 0.f};
            if (xn_out) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
                static_for<0, 12>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    s4 += (sub4<0>(acc[t]) + sub4<1>(acc[t])) + (sub4<2>(acc[t]) + sub4<3>(acc[t]));
                    asm volatile("" : "+v"(s4));
                });
                float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.0f / E);
                const f32x4 mean4 = {mean, mean, mean, mean};
                f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
                static_for<0, 12>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    static_for<0, 4>([&](auto q) {
                        const f32x4 d = sub4<decltype(q)::value>(acc[t]) - mean4;
                        q4 = __builtin_elementwise_fma(d, d, q4);
                    });
                    asm volatile("" : "+v"(q4));
                });
                float sq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                sq += __shfl_xor(sq, 32, 64);
                const float rstd = rsqrtf(sq * (1.0f / E) + eps);
                const float nmr = -mean * rstd;
                rstd4 = f32x4{rstd, rstd, rstd, rstd};
                nmr4 = f32x4{nmr, nmr, nmr, nmr};
            }
            char* const xo = x_out_img ? (char*)x + (grp * (32 * E * 4) + lane16) : (char*)x + ((size_t)grow * (E * 4) + half * 16);
            const int xt = x_out_img ? 4096 : 128, xq = x_out_img ? 1024 : 32;
            const bool x_store = x_out_img ? grp_valid : valid;
            char* const op = act_blk ? (char*)xn_out + (grp * (32 * E * 2) + lane16) : (char*)xn_out + ((size_t)crow * (E * 2) + half * 16);
            const int ostep = act_blk ? 1024 : 32;
            const bool n_store = act_blk ? grp_valid : valid;
            static_for<0, 12>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                if (x_store) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[t][4 * q + r];
                        *reinterpret_cast<f32x4*>(xo + xt * t + xq * q) = v;
                    }
                }
                if (xn_out) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        // registers 8p..8p+3 = features 32t + 16p + 4 half + 0..3, registers 8p+4..8p+7 = the same + 8: one
                        // v_permlane32_swap per register pair hands each lane eight CONSECUTIVE features (T21)
                        const f32x4 n0 = __builtin_elementwise_fma(p ? sub4<2>(acc[t]) : sub4<0>(acc[t]), rstd4, nmr4);
                        const f32x4 n1 = __builtin_elementwise_fma(p ? sub4<3>(acc[t]) : sub4<1>(acc[t]), rstd4, nmr4);
                        vec4 lo, hi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] = (T)n0[e]; hi[e] = (T)n1[e]; }
                        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                        u32x4 o;
#pragma unroll
                        for (int d2 = 0; d2 < 2; ++d2) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(l2[d2], h2[d2], false, false);
                            o[d2] = sw[0];
                            o[2 + d2] = sw[1];
                        }
                        if (n_store) *reinterpret_cast<u32x4*>(op + ostep * (2 * t + p)) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef BLOCKS_EPI_LATE_LOADS
                load_x(next_tile, tt);                                   // accumulator tile t is free: the next tile's x goes in
                __builtin_amdgcn_sched_barrier(0);
#endif
            });
#ifdef BLOCKS_EPI_LATE_LOADS
            static_for<0, 12>([&](auto tt) { load_x(next_tile, tt); });
#endif
        }
        BST(q5);
        BACC(10, q4, q5);
    }
#ifdef BLOCKS_STAMPS
    st[15] = bstamp() - tstart;
    if (lane == 0) for (int i = 0; i < 16; ++i) g_bsstamps[(blockIdx.x * 4 + wave) * 16 + i] = st[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead LDS-DMA must not outlive the workgroup's LDS allocation
}

template <typename T>
int launch_t(float* x, const void* attn, void* xn_out, const void* wseq, const float* b1f, const float* bproj, const float* b2,
             int64_t M, float eps, int layout, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = block16s_kernel<T>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int ntiles = (int)((M + 127) / 128);
    const int cus = mst_persistent_grid();
    const int nblk = ntiles < cus ? ntiles : cus;
    kern<<<dim3(nblk), dim3(256), LDS_BYTES, s>>>(x, (const T*)attn, (T*)xn_out, (const char*)wseq, b1f, bproj, b2, (int)M, ntiles, eps, layout);
    return mst_check_launch("block16s");
}

}  // namespace

#ifdef BLOCKS_STAMPS
extern "C" int mst_debug_blocks_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bsstamps), sizeof(unsigned long long) * n);
}
#endif

int launch_block16s(float* x, const void* attn, void* xn_out, int dt, const void* wseq, const float* b1f, const float* bproj,
                    const float* b2, int64_t M, int E_, float eps, int layout, hipStream_t s) {
    MST_CHECK_ARG(E_ == E, "block_fused_s: embed_dim=%d unsupported (384)", E_);
    MST_CHECK_ARG(M > 0 && M < (1ll << 31) - 128, "block_fused_s: bad M");
    if (dt == MST_BF16) return launch_t<bf16_t>(x, attn, xn_out, wseq, b1f, bproj, b2, M, eps, layout, s);
    if (dt == MST_F16) return launch_t<f16_t>(x, attn, xn_out, wseq, b1f, bproj, b2, M, eps, layout, s);
    mst_set_error("block_fused_s: dtype %d unsupported (f16 / bf16)", dt);
    return MST_EINVAL;
}
