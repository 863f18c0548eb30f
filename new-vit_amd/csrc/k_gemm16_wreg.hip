// Weights-in-registers 16-bit MFMA GEMM for the QKV projection (contract of k_gemm16.hip, bias epilogue with the q scaling):
// C[M, N] = (A[M, 384] . W[N, 384]^T + bias) * (col < scale_cols ? col_scale : 1), M ~ 10^5 token rows, N a multiple of 384.
//
// Why (profiles/r02u_qkv_*): on this shape the tiled kernels move 3.2 GB from L2 into LDS for 0.27 GB of unique operands -- the
// 128 x 384 tile (k_gemm16_mid.hip) re-streams the 288 KiB weight tile for every 128 token rows -- and that stream, not the MFMA
// pipe, sets their time (loads alone 0.27 ms at the ~20 B/clk a CU gets from L2 on this access shape; 0.41 ms with the MFMAs, whose
// wave cannot start while its eight LDS-DMA instructions wait for room in the memory pipeline; 0.58 ms with the epilogue).  Here a
// workgroup never re-reads a weight:
//   * one persistent 8-wave workgroup per CU owns ONE 384-column tile for the whole launch; wave w keeps W[48 columns][384] as 36
//     MFMA A-operand fragments in 144 registers, loaded once (2 waves per SIMD at 256 registers);
//   * the token rows stream through a ring of 32-row x 768-byte LDS slots filled by LDS-DMA (24 pieces per slot, 3 per wave, issued
//     BETWEEN the MFMAs of the chunk three ahead of their use); every A byte is fetched by the three workgroups of its XCD that own
//     the three column tiles (same XCD = same L2: HBM sees it once), 0.8 GB instead of 3.2 GB into LDS;
//   * per chunk and wave 72 MFMA 16x16x32 against 24 ds_read_b128 (software-pipelined one k-step ahead, counted lgkmcnt), one
//     s_barrier per chunk; 16-byte slot of chunk c of row r = c ^ (r & 15): conflict-free fragment reads, and because the XOR stays
//     inside 256-byte blocks every 1 KiB DMA piece still fetches eight whole 128-byte lines;
//   * the weight rows are fetched in the order that makes a lane's 12 accumulator columns CONSECUTIVE output columns
//     (tile t row 4q+i <-> column 12q + 4t + i): 24 contiguous bytes per lane and row go into an LDS staging tile (32 x 768 bytes
//     for the workgroup), and behind the next chunk's barrier every wave stores four whole 768-byte rows of it with three
//     16-byte-per-lane instructions (stores straight from the accumulators -- 8- and 16-byte pieces scattered over 16 rows per
//     instruction -- were slower still).
// Measured (tools/wreg_clock.py, steady state after 2 s of launches, M = 350,720): 0.297 ms against 0.523 ms for the mid-tile
// kernel.  The launch is power-limited: MFMAs + LDS reads alone hold 2.26 GHz (0.221 ms), the 808 MB store stream alone 2.36 GHz
// (0.161 ms), both together 1.75-1.9 GHz -- 11 % more cycles than the arithmetic alone, 35 % more time
// (profiles/r02v_qkv_wreg.txt).  Spreading the stores over the k-steps (-DWREG_SPREAD_ST) changes nothing.
// XCD x (= blockIdx & 7) sweeps the x-th eighth of the chunks; inside it CU g works on column tile g % tiles_n.
#include <cstdlib>

#include "mst_common.h"

namespace {

constexpr int KD = 384, BN = 384, CH = 32;
constexpr int ROWB = KD * 2;                       // 768-byte LDS rows
constexpr int SLOT = CH * ROWB;                    // 24 KiB per chunk
#ifndef WREG_NSLOT
#define WREG_NSLOT 4
#endif
constexpr int NSLOT = WREG_NSLOT;                  // chunks it+1 .. it+NSLOT-1 are in flight while chunk it is multiplied
constexpr int STG_PITCH = ROWB + 16;                // staging rows: 784 bytes (a row shifts by 4 banks)
constexpr int STG_BYTES = CH * STG_PITCH;          // one 32 x 384 output tile of the workgroup, 16-bit
constexpr int STG_OFF = NSLOT * SLOT;
constexpr int LDS_BYTES = STG_OFF + 2 * STG_BYTES; // ring + two staging tiles (144.5 KiB)
constexpr int NKT = KD / 32;                       // 12 k-steps
constexpr int PPW = SLOT / 1024 / 8;               // LDS-DMA pieces per wave and chunk (3)
static_assert(NSLOT == 4, "the counted vmcnt table below is written for three chunks in flight");

#define WREG_RD(dst, base, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(imm) : "memory")

// BLK: A arrives in the 16-bit "blocked" layout of include/mst_hip.h (what the single-role block kernel writes between two blocks of
// an encoder): a chunk of 32 rows IS 24 contiguous KiB pieces [k/16][row + 32 ((k/8)&1)][8], the LDS slot is a plain copy of it
// (LDS-DMA with contiguous sources), and a fragment (row 16 mt + frow, k = 32 kt + 8 kq ..+7) sits at one lane offset + immediates.
template <typename T, bool BLK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm16_wreg_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ bias, T* __restrict__ C,
    int64_t ldc, int M, int N, float col_scale, int scale_cols, int nchunks) {
    typedef typename V8<T>::type vec8;
    typedef __attribute__((ext_vector_type(4))) T out4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 15, kq = lane >> 4;

    // ---- which chunks, which columns
    const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int tiles_n = N / BN;
    const int nt = g % tiles_n, jcu = g / tiles_n;
    const int c_nt = (G - nt + tiles_n - 1) / tiles_n;              // CUs of this XCD on column tile nt
    const int per_xcd = (nchunks + 7) >> 3;
    const int lo = xcd * per_xcd;
    const int hi = lo + per_xcd < nchunks ? lo + per_xcd : nchunks;
    const int n0 = nt * BN + wave * 48;
    const float cs = nt * BN < scale_cols ? col_scale : 1.f;

    // ---- the wave's weights: tile t row m = frow  <->  output column n0 + 12 (m >> 2) + 4 t + (m & 3)
    vec8 w[3][NKT];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const T* wr = W + (int64_t)(n0 + 12 * (frow >> 2) + 4 * t + (frow & 3)) * ldw + kq * 8;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) w[t][kt] = *reinterpret_cast<const vec8*>(wr + kt * 32);
    }
    f32x4 b[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        b[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias) b[t] = *reinterpret_cast<const f32x4*>(bias + n0 + 12 * kq + 4 * t);
    }
    // a use of every register loaded above: the compiler waits for these loads HERE, not with a short vmcnt inside the loop
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        asm volatile("" : "+v"(b[t]));
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) asm volatile("" : "+v"(w[t][kt]));
    }

    // ---- LDS-DMA: piece p = 3 wave + u of a slot, lane l fills bytes [1024 p + 16 l, +16): row r, slot s <- chunk s ^ (r & 15)
    unsigned dma_off[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int o = (wave * PPW + u) * 1024 + lane * 16;
        const int r = o / ROWB;
        dma_off[u] = BLK ? (unsigned)o : (unsigned)(r * (int)lda * 2 + ((((o % ROWB) >> 4) ^ (r & 15)) << 4));
    }
    const int64_t chunk_bytes = BLK ? (int64_t)SLOT : (int64_t)CH * lda * 2;
    auto issue_piece = [&](int chunk, int slot, int u) {
#ifdef WREG_ABL_SAME_A
        const char* base = reinterpret_cast<const char*>(A) + (int64_t)(chunk & 7) * chunk_bytes;
#else
        const char* base = reinterpret_cast<const char*>(A) + (int64_t)chunk * chunk_bytes;
#endif
        unsigned off = dma_off[u];
        const int valid = M - chunk * CH;
        if (!BLK && valid < CH) {                   // ragged last chunk: clamp the row (blocked: whole groups are allocated, rows past M are never stored)
            const int o = (wave * PPW + u) * 1024 + lane * 16;
            int r = o / ROWB;
            const int ck = ((o % ROWB) >> 4) ^ (r & 15);
            r = r < valid ? r : valid - 1;
            off = (unsigned)(r * (int)lda * 2 + ck * 16);
        }
#ifdef WREG_ABL_NO_DMA
        asm volatile("" ::"v"(off), "s"(base));
#else
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(smem + slot * SLOT + (wave * PPW + u) * 1024), 16, 0, 0);
#endif
    };

    // ---- fragment reads: row 16 mt + frow, chunk 4 kt + kq  ->  byte r*768 + (kt >> 2)*256 + (((4 (kt & 3) + kq) ^ frow) << 4)
    int rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = BLK ? (kq >> 1) * 1024 + (frow + 32 * (kq & 1)) * 16 : frow * ROWB + (((4 * j + kq) ^ frow) << 4);
    // ---- epilogue, two parts.  put(): at the end of iteration `it` every wave converts its 32 x 48 results and writes them into
    // staging tile it & 1 (24 consecutive bytes per lane and row).  flush(): behind the NEXT chunk barrier wave w moves rows
    // 4w .. 4w+3 of that tile (768 contiguous bytes each = six whole lines) to C with three 16-byte-per-lane stores.  The tile
    // written in iteration it+2 is the one flushed in iteration it+1: every wave has passed that flush before it reaches barrier
    // it+2, which the writers are behind.
    const int put_off = frow * STG_PITCH + wave * 96 + kq * 24;
    int fl_lds[3];
    unsigned fl_glb[3];
    int fl_row[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = lane + 64 * j;                // 16-byte piece of the wave's 4 x 768 bytes
        fl_row[j] = wave * 4 + q / 48;
        fl_lds[j] = fl_row[j] * STG_PITCH + (q % 48) * 16;
        fl_glb[j] = (unsigned)(fl_row[j] * (int)ldc * 2 + nt * BN * 2 + (q % 48) * 16);
    }
    f32x4 acc[2][3];
    auto put = [&](int it) {
        char* stg = smem + STG_OFF + (it & 1) * STG_BYTES + put_off;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                out4 pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) pk[i] = (T)(acc[mt][t][i] * cs);
                *reinterpret_cast<out4*>(stg + mt * 16 * STG_PITCH + t * 8) = pk;
            }
    };
    auto flush_piece = [&](int chunk, int it, int j) {
        const char* stg = smem + STG_OFF + (it & 1) * STG_BYTES;
        char* cb = reinterpret_cast<char*>(C) + (int64_t)chunk * CH * ldc * 2;
        const int valid = M - chunk * CH;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + fl_lds[j]);
#ifdef WREG_ABL_NO_STORE
        asm volatile("" ::"v"(v));
#else
        if (valid >= CH || fl_row[j] < valid) {
#ifdef WREG_PLAIN_ST
            *reinterpret_cast<u32x4*>(cb + fl_glb[j]) = v;
#else
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(cb + fl_glb[j]));   // write-once stream: 0.297 vs 0.310 ms
#endif
        }
#endif
    };
    auto flush = [&](int chunk, int it) {
#pragma unroll
        for (int j = 0; j < 3; ++j) flush_piece(chunk, it, j);
    };

    int c = lo + jcu;
    if (c >= hi) return;
#ifdef WREG_CLOCK                                   // diagnostic build: shader clock of this launch (s_memtime / s_memrealtime)
    const uint64_t clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int d = 0; d < NSLOT - 1; ++d)
        if (c + d * c_nt < hi) {
#pragma unroll
            for (int u = 0; u < PPW; ++u) issue_piece(c + d * c_nt, d, u);
        }

    int prev = -1, last_it = 0;                     // chunk waiting in the staging tile
    for (int it = 0; c < hi; c += c_nt, ++it) {
        // chunk c has landed once at most the operations issued behind its pieces are outstanding (vector memory operations
        // complete in order): iteration k issues the 3 stores of flush(k-1), then the 3 pieces of chunk k+3
        if (c + 2 * c_nt < hi) {
            if (it <= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // DMA(1) DMA(2)  /  DMA(2) DMA(3)
            else if (it == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");      // DMA(3) st(0) DMA(4)
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // st(k-3) DMA(k+1) st(k-2) DMA(k+2)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // lgkmcnt: this wave's staging writes
#ifndef WREG_SPREAD_ST
        if (prev >= 0) flush(prev, it - 1);
#endif

        const int slot = it % NSLOT;
        const int nxt = c + (NSLOT - 1) * c_nt, nslot = (it + NSLOT - 1) % NSLOT;
        const bool more = nxt < hi;
        int rbs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rbs[j] = rb[j] + slot * SLOT;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[mt][t] = b[t];

        u32x4 a[2][2];
        WREG_RD(a[0][0], rbs[0], 0);
        if constexpr (BLK) WREG_RD(a[0][1], rbs[0], 256);
        else WREG_RD(a[0][1], rbs[0], 16 * ROWB);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < NKT) {
                switch (kt + 1) {                   // immediates must be literals
#define WREG_NEXT(KN)                                                                  \
    case KN:                                                                           \
        if constexpr (BLK) {                                                           \
            WREG_RD(a[(KN) & 1][0], rbs[0], (KN) * 2048);                              \
            WREG_RD(a[(KN) & 1][1], rbs[0], (KN) * 2048 + 256);                        \
        } else {                                                                       \
            WREG_RD(a[(KN) & 1][0], rbs[(KN) & 3], ((KN) >> 2) * 256);                 \
            WREG_RD(a[(KN) & 1][1], rbs[(KN) & 3], 16 * ROWB + ((KN) >> 2) * 256);     \
        }                                                                              \
        break;
                    WREG_NEXT(1) WREG_NEXT(2) WREG_NEXT(3) WREG_NEXT(4) WREG_NEXT(5) WREG_NEXT(6)
                    WREG_NEXT(7) WREG_NEXT(8) WREG_NEXT(9) WREG_NEXT(10) WREG_NEXT(11)
#undef WREG_NEXT
                }
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[cur][0]), "+v"(a[cur][1])::"memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(a[cur][1])::"memory");
            }
            const vec8 a0 = __builtin_bit_cast(vec8, a[cur][0]), a1 = __builtin_bit_cast(vec8, a[cur][1]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#ifdef WREG_ABL_NO_MFMA
                asm volatile("" : "+v"(acc[0][t]), "+v"(acc[1][t]) : "v"(w[t][kt]), "v"(a0), "v"(a1));
#else
                acc[0][t] = mfma16(w[t][kt], a0, acc[0][t]);
                acc[1][t] = mfma16(w[t][kt], a1, acc[1][t]);
#endif
            }
            if ((kt & 3) == 1) {                    // k-steps 1, 5, 9: one piece of the chunk three ahead
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_piece(nxt, nslot, kt >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef WREG_SPREAD_ST
            if ((kt & 3) == 3) {                    // k-steps 3, 7, 11: one 1 KiB store of the previous chunk
                __builtin_amdgcn_sched_barrier(0);
                if (prev >= 0) flush_piece(prev, it - 1, kt >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        put(it);
        prev = c;
        last_it = it;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // lgkmcnt: this wave's staging writes         // every wave's put of the last chunk (the compiler waits for its own LDS writes)
    flush(prev, last_it);
#ifdef WREG_CLOCK
    if (threadIdx.x == 0) {                         // overwrites the first 16 bytes of this workgroup's first output row: diagnostic only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t* o = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(C) + (int64_t)(lo + jcu) * CH * ldc * 2 + nt * BN * 2);
        o[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
}

template <typename T, bool BLK>
int launch_t(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M, int N,
             float cs, int sc, hipStream_t s) {
    static mst_lds_once lds_once;
    auto kern = gemm16_wreg_kernel<T, BLK>;
    mst_allow_lds((const void*)kern, LDS_BYTES, &lds_once);
    const int nchunks = (int)((M + CH - 1) / CH);
    kern<<<dim3(mst_persistent_grid()), dim3(512), LDS_BYTES, s>>>((const T*)A, lda, (const T*)W, ldw, bias, (T*)C, ldc, (int)M, N, cs,
                                                                  sc, nchunks);
    return mst_check_launch("gemm16_wreg");
}

}  // namespace

// K = 384 exactly (the register-resident weight slice), 384-column tiles, plain bias epilogue into the operand type, the scaled
// column range aligned to the tiles, at least 8,192 rows (profiles/r02v_qkv_wreg.txt, r03v)
bool gemm16_wreg_applicable(int64_t M, int N, int K, int dt, int cdt, int epi, int scale_cols, int64_t lda, int64_t ldc) {
    const int G = mst_persistent_grid() >> 3;
    const char* e = getenv("MST_GEMM_WREG_MIN_M");    // tests lower the threshold to reach the few-chunks-per-CU paths
    const int64_t min_m = e ? atoll(e) : 8192;        // measured crossover against the tiled kernels: faster from ~10 k rows (0.047 vs 0.080 ms at 43,840)
    return K == KD && N % BN == 0 && N / BN <= G && dt == cdt && epi == MST_EPI_BIAS && (scale_cols % BN == 0 || scale_cols >= N) &&
           M >= min_m && M < (1ll << 31) - CH && lda * 2 * CH < (1ll << 31) && ldc * 2 * CH < (1ll << 31) && ldc % 4 == 0;
}

int launch_gemm16_wreg(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                       int64_t M, int N, float col_scale, int scale_cols, hipStream_t s, int a_blocked) {
    if (a_blocked) {
        if (dt == MST_BF16) return launch_t<bf16_t, true>(A, lda, W, ldw, bias, C, ldc, M, N, col_scale, scale_cols, s);
        if (dt == MST_F16) return launch_t<f16_t, true>(A, lda, W, ldw, bias, C, ldc, M, N, col_scale, scale_cols, s);
    }
    if (dt == MST_BF16) return launch_t<bf16_t, false>(A, lda, W, ldw, bias, C, ldc, M, N, col_scale, scale_cols, s);
    if (dt == MST_F16) return launch_t<f16_t, false>(A, lda, W, ldw, bias, C, ldc, M, N, col_scale, scale_cols, s);
    mst_set_error("gemm16_wreg: bad operand dtype %d", dt);
    return MST_EINVAL;
}
