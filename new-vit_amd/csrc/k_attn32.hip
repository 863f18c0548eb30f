// Exact-fp32 flash attention for MST_F32 ("exact parity") mode: same structure as k_attn16.hip,
// on v_mfma_f32_32x32x2_f32 (one fp32 per lane per operand, k-ordered fmaf chain).
//   S^T[key][q] = K . Q^T : 32 k-steps over d = 64 per 32-key block, Q resident in registers.
//   O^T[d][q]  += V^T . P^T: the S^T accumulator register r already IS the B operand of a k-step
//       whose two k's are key (r&3)+8(r>>2) (lanes 0-31) and that key + 4 (lanes 32-63); the A
//       operand reads V[that key][d = lane&31] from LDS.  No cross-lane movement at all.
// LDS rows padded to 65 dwords: both fragment reads are bank-conflict-free.
// Reference arithmetic: attention.py:56-66.
#include "mst_common.h"

namespace {

constexpr int LDK = 65;
constexpr int TILE_F = 64 * LDK;  // floats per K or V tile

__global__ __launch_bounds__(256) void attn32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                     int N, int heads) {
    extern __shared__ float sm[];  // K[2][64][65] | V[2][64][65]
    float* const Ks = sm;
    float* const Vs = sm + 2 * TILE_F;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: the q-blocks of one (sequence, head) get consecutive tile ids on ONE XCD, so its
    // K/V (re-read by every q-block) stay in that XCD's L2 (plain (x,y,z) order deals them over all 8 XCDs:
    // rocprofv3 FETCH_SIZE showed 5.7x the algorithmic bytes).
    const int nqb = (N + 127) >> 7;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = tile % nqb, h = (tile / nqb) % heads, seq = tile / (nqb * heads);
    const int E = heads * 64, ld = 3 * E;
    const float* base = qkv + (int64_t)seq * N * ld;
    const int h2 = lane >> 5, ql = lane & 31;

    const int q = qb * 128 + wave * 32 + ql;
    const bool wave_active = (qb * 128 + wave * 32) < N;
    float bq[32];  // Q[q][2*kk + h2]
    {
        const int qc = q < N ? q : N - 1;
        const float* qp = base + (int64_t)qc * ld + h * 64 + h2;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) bq[kk] = qp[2 * kk];
    }

    const int sr = tid >> 2, sc = (tid & 3) * 16;
    float4 rk[4], rv[4];
    auto gload = [&](int t) {
        int key = t * 64 + sr;
        key = key < N ? key : N - 1;
        const float* kp = base + (int64_t)key * ld + E + h * 64 + sc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rk[u] = *reinterpret_cast<const float4*>(kp + 4 * u);
            rv[u] = *reinterpret_cast<const float4*>(kp + E + 4 * u);
        }
    };
    auto lstore = [&](int buf) {
        float* kd = Ks + buf * TILE_F + sr * LDK + sc;
        float* vd = Vs + buf * TILE_F + sr * LDK + sc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kd[4 * u] = rk[u].x; kd[4 * u + 1] = rk[u].y; kd[4 * u + 2] = rk[u].z; kd[4 * u + 3] = rk[u].w;
            vd[4 * u] = rv[u].x; vd[4 * u + 1] = rv[u].y; vd[4 * u + 2] = rv[u].z; vd[4 * u + 3] = rv[u].w;
        }
    };

    f32x16 oT[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[0][r] = oT[1][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (N + 63) >> 6;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        if (wave_active) {
            const float* Kb = Ks + buf * TILE_F;
            const float* Vb = Vs + buf * TILE_F;
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
                const float* kr = Kb + (kb * 32 + ql) * LDK + h2;
#pragma unroll
                for (int kk = 0; kk < 32; ++kk)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * kk], bq[kk], s[kb], 0, 0, 0);
            }
            if (t == nt - 1 && (N & 63)) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                        if (key >= N) s[kb][r] = -INFINITY;
                    }
            }
            float mx = s[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);
            m_run = m_new;
            float lsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = expf(s[kb][r] - m_new);
                    s[kb][r] = p;
                    lsum += p;
                }
            l_run = l_run * alpha + lsum;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    oT[0][r] *= alpha;
                    oT[1][r] *= alpha;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const float* vr = Vb + key * LDK + ql;
                    oT[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[kb][r], oT[0], 0, 0, 0);
                    oT[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[kb][r], oT[1], 0, 0, 0);
                }
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    if (wave_active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < N) {
            float* op = out + ((int64_t)seq * N + q) * E + h * 64 + 4 * h2;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 o;
                    o.x = oT[db][g * 4 + 0] * inv;
                    o.y = oT[db][g * 4 + 1] * inv;
                    o.z = oT[db][g * 4 + 2] * inv;
                    o.w = oT[db][g * 4 + 3] * inv;
                    *reinterpret_cast<float4*>(op + db * 32 + 8 * g) = o;
                }
        }
    }
}

}  // namespace

int launch_attn32(const float* qkv, int n_seq, int N, int heads, float* out, hipStream_t s) {
    MST_CHECK_ARG(n_seq > 0 && N > 0 && heads > 0, "attention32: bad sizes");
    const int64_t nwg = (int64_t)((N + 127) / 128) * heads * n_seq;
    MST_CHECK_ARG(nwg < (1ll << 31), "attention32: grid too large");
    const size_t sh = (size_t)4 * TILE_F * sizeof(float);  // 66,560 B
    static mst_lds_once lds_once;
    mst_allow_lds((const void*)attn32_kernel, (int)sh, &lds_once);
    attn32_kernel<<<dim3((unsigned)nwg), dim3(256), sh, s>>>(qkv, out, N, heads);
    return mst_check_launch("attention32");
}
