// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmst_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mst_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MST_WAVE 64
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <typename T> struct V8;
template <> struct V8<bf16_t> { typedef bf16x8 type; typedef bf16x4 half_type; };
template <> struct V8<f16_t> { typedef f16x8 type; typedef f16x4 half_type; };

// D(16x16 f32) += A(16x32) * B(32x16); lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
// D[row=(l>>4)*4+r][col=l&15].
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// D(32x32 f32) += A(32x16) * B(16x32); lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31];
// D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float v) {  // nn.GELU() exact form (mlp.py:22)
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// GELU for 16-bit outputs: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on erf, <= 5e-7 abs /
// 2e-4 rel on gelu: far below one fp16/bf16 ulp), one v_rcp + one v_exp instead of libm erff.
// The negative branch multiplies by the complementary term directly (no 1 - erf cancellation).
__device__ __forceinline__ float gelu_fast(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float pe = p * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);  // = erfc(z)
    const float h = 0.5f * v;
    return v >= 0.f ? h * (2.0f - pe) : h * pe;
}

// Transcendental-free GELU for 16-bit outputs inside MFMA-dense loops: erf(z) = z*P(z^2) on |z| <= 3.2 (degree-8
// Chebyshev fit, |erf err| <= 8e-5, |gelu err| <= 1.8e-4 absolute: below one fp16/bf16 ulp of O(1) activations),
// saturating to +-1 beyond.  9 FMAs + 5 simple ops, no v_exp / v_rcp (quarter-rate issue).
__device__ __forceinline__ float gelu_poly(float v) {
    const float z = fminf(fabsf(v) * 0.70710678118654752440f, 3.2f);
    const float u = z * z;
    float p = 3.263779068761133e-08f;
    p = fmaf(p, u, -1.673741012956179e-06f);
    p = fmaf(p, u, 3.751051790327989e-05f);
    p = fmaf(p, u, -0.000488409298395181f);
    p = fmaf(p, u, 0.0041668641449124355f);
    p = fmaf(p, u, -0.025040875590456917f);
    p = fmaf(p, u, 0.11119564373069263f);
    p = fmaf(p, u, -0.37554270907102755f);
    p = fmaf(p, u, 1.1283444184507676f);
    const float e = fminf(p * z, 1.0f);            // erf(|v|/sqrt2)
    const float h = 0.5f * v;
    return fmaf(fabsf(h), e, h);                   // 0.5 v (1 + sign(v) erf) = h + |h| e
}

// GELU for 16-bit outputs inside ISSUE-bound MFMA loops: v * sigmoid(v * P(v^2)), P fitted (minimax, scipy) to the exact
// erf form.  Beside dense MFMAs the SIMD's vector issue port is the scarce resource (an MFMA holds it 8 of its 16
// cycles, a plain VALU op 4, v_exp/v_rcp 8): this form costs 36 (bf16) / 44 (fp16) issue cycles per value against 60 for
// gelu_poly.  |error| vs nn.GELU(): 2.7e-4 with the linear P (bf16: 1/30 of a bf16 ulp at 1.0), 2.5e-5 with the
// quadratic P (fp16).  exp2 overflow for v < -10 gives rcp(inf) = 0 -> -0, the right limit.
template <typename T> __device__ __forceinline__ float gelu_sig(float v);
template <> __device__ __forceinline__ float gelu_sig<bf16_t>(float v) {
    const float w = fmaf(v * v, -0.06940179f * 1.4426950408889634f, -1.60031416f * 1.4426950408889634f);
    const float e = __builtin_amdgcn_exp2f(v * w);                 // exp(-v (a + b v^2))
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
template <> __device__ __forceinline__ float gelu_sig<f16_t>(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, -8.0f, 8.0f);       // the quadratic P turns over at |v| = 8.35
    const float u = c * c;
    float w = fmaf(u, 7.03033577e-04f * 1.4426950408889634f, -7.40112920e-02f * 1.4426950408889634f);
    w = fmaf(w, u, -1.59501577f * 1.4426950408889634f);
    const float e = __builtin_amdgcn_exp2f(c * w);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous run of tile ids (neighbouring tiles share operand panels in its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// host side ---------------------------------------------------------------------------------
void mst_set_error(const char* fmt, ...);
#define MST_CHECK_ARG(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            mst_set_error(__VA_ARGS__);         \
            return MST_EINVAL;                  \
        }                                       \
    } while (0)
int mst_check_launch(const char* what);
// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, and a process may drive
// several GPUs.  Thread-safe; `slot` is a static per-kernel token owned by the call site.
struct mst_lds_once { unsigned long long done_mask = 0; };
void mst_allow_lds(const void* kernel, int bytes, mst_lds_once* slot);
// Compute units of the current device (256 on an MI355X in SPX mode, 32 per logical GPU in CPX mode), cached per device and
// rounded down to a multiple of 8 (the persistent kernels deal tile ids to the 8 XCDs): the size of a persistent grid.
int mst_persistent_grid(void);

// kernel launchers shared between translation units (all asynchronous on `s`)
int launch_layernorm(const float* x, int64_t xs, const float* g, const float* b, void* out, int odt,
                     int64_t os, int64_t rows, int cols, float eps, hipStream_t s);
int launch_gemm16(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias,
                  void* C, int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                  float col_scale, int scale_cols, hipStream_t s);
int launch_patch_rows16(const void* vol, int idt, int n, int H, int W, const void* wp, int dt, const float* bias, const float* prefix,
                        int n_prefix, const float* pos_patch, float* x, void* xn, hipStream_t s);
bool gemm16_big_applicable(int64_t M, int N, int K);
int launch_gemm16_big(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias,
                      void* C, int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                      float col_scale, int scale_cols, hipStream_t s);
bool gemm16_wreg_applicable(int64_t M, int N, int K, int dt, int cdt, int epi, int scale_cols, int64_t lda, int64_t ldc);
// a_blocked: A in the 16-bit blocked layout of include/mst_hip.h (whole 32-row groups allocated; lda ignored)
int launch_gemm16_wreg(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                       int64_t M, int N, float col_scale, int scale_cols, hipStream_t s, int a_blocked = 0);
bool gemm16_mid_applicable(int64_t M, int N, int K, int dt, int cdt, int epi);
int launch_gemm16_mid(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                      int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma, float col_scale, int scale_cols,
                      hipStream_t s);
int launch_gemm32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                  float* C, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                  float col_scale, int scale_cols, hipStream_t s);
int launch_gemm16_splitk(const void* A, int dt, int64_t lda, const void* W, int64_t ldw, float* Cpart, int64_t ldc, int64_t M, int N, int K,
                         int splits, int64_t split_stride, hipStream_t s);
int launch_cvt16(const float* x, int64_t ldx, int64_t rows, int cols, float scale, void* out, int dt, int64_t ldo, int transpose,
                 int64_t rows_pad, hipStream_t s);
int launch_conv_dgrad32(const float* dz, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const float* Wt, int H, int W,
                        int Cin, float* dx, hipStream_t s);
int launch_conv_dgrad16(const void* dz, int dt, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const void* Wt, int H, int W,
                        int Cin, float* dx, hipStream_t s);
int launch_conv_wgrad32(const float* dz, const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                        int nsplit, int64_t rows_per_split, hipStream_t s);
int launch_conv_wgrad16(const void* dz, const void* x, int dt, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                        int nsplit, int64_t rows_per_split, hipStream_t s);
int launch_rope_rows(float* qkv, int64_t rows, int L, int heads, int hd, const float* freqs, float sign, hipStream_t s);
int launch_im2col_nhwc16(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, void* col, int dt, hipStream_t s);
int launch_maxpool_nhwc16(const void* x, int dt, int n, int H, int W, int C, void* y, hipStream_t s);
int launch_cvt32(const void* x, int dt, int64_t n, float* out, hipStream_t s);
int launch_conv_gemm16(const void* x, int dt, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const void* Wg, const float* bias,
                       void* out, int cdt, int Cout, int epi, hipStream_t s);
int launch_conv_gemm32(const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const float* Wg, int64_t ldw,
                       const float* bias, float* out, int64_t ldc, int Cout, int Kpad, int epi, const float* gamma, hipStream_t s);
bool gemm32_small_applicable(int64_t M, int N, int K);
int launch_gemm32_small(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                        int64_t M, int N, int K, int epi, const float* gamma, float col_scale, int scale_cols, hipStream_t s);
// scan: 1 = *amax = max(*amax, max|x|) first (one extra pass over x); 0 = *amax already covers x (its producer kept it)
int launch_quant8(const void* x, int dt, int64_t n, float* amax, void* out8, int scan, hipStream_t s);
// quantise under a calibrated scale the caller owns (read-only)
int launch_quant8_static(const void* x, int dt, int64_t n, const float* amax, void* out8, hipStream_t s);
// out_amax (nullable, 16-bit C, non-residual epilogues): *out_amax = max(*out_amax, max |C as stored|)
// c_amax (cdt == MST_F8E4M3 only): calibrated scale of the e4m3 output
int launch_gemm8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* bias, const float* a_amax,
                 float w_scale, void* C, int cdt, int64_t ldc, int64_t M, int N, int K, int epi, const float* gamma,
                 float col_scale, int scale_cols, float* out_amax, const float* c_amax, hipStream_t s);
int launch_layernorm_f8(const float* x, int64_t xs, const float* g, const float* b, void* out8, int64_t os, int64_t rows,
                        int cols, float eps, const float* amax, hipStream_t s);
int launch_amax_merge(float* out, const float* in, int n, hipStream_t s);
int launch_mlp16(float* x, void* xn_out, int dt, const void* wpack, const float* b1f, const float* b2, int64_t M, int E,
                 float eps, hipStream_t s);
// out-projection + residual + LayerNorm2 + MLP + residual (+ next normalise) in one launch (k_block16.hip); attn may alias xn_out
size_t block16_scratch_bytes(void);
int launch_block16(float* x, const void* attn, void* xn_out, int dt, const void* wproj, const float* bproj, const void* wpack,
                   const float* b1f, const float* b2, void* scratch, int64_t M, int E, float eps, hipStream_t s);
// the same in the single-role form (k_block16s.hip): weights as one stream in consumption order, no scratch
int launch_block16s(float* x, const void* attn, void* xn_out, int dt, const void* wseq, const float* b1f, const float* bproj,
                    const float* b2, int64_t M, int E, float eps, int layout, hipStream_t s);
// log2q: q arrives pre-multiplied by log2(e) as well (the encoder folds it into the QKV epilogue's fp32 q scaling, so no
// second 16-bit rounding of q); 0 = plain head_dim^-0.5 scaling, the public mst_attention* contract
// out_blocked: out in the 16-bit blocked layout of include/mst_hip.h (rows = seq * N + q; whole 32-row groups allocated)
int launch_attn16(const void* qkv, int dt, int n_seq, int N, int heads, void* out, int log2q, hipStream_t s, int out_blocked = 0);
int launch_attn32(const float* qkv, int n_seq, int N, int heads, float* out, hipStream_t s);
int launch_cls_attn(const void* qkv, int dt, int n_seq, int N, int heads, float* probs, void* out, int log2q, hipStream_t s);
int launch_cls_probs(const void* qkv, int dt, int n_seq, int N, int heads, int hd, float* probs, int log2q,
                     hipStream_t s);
int launch_probs_full(const void* qkv, int dt, int n_seq, int N, int heads, int hd, float* probs, int log2q,
                      hipStream_t s);
int launch_patch_embed(const void* vol, int idt, int n, int H, int W, const void* wp, int dt,
                       const float* bias, const float* prefix, int n_prefix, const float* pos_patch,
                       int E, float* x, hipStream_t s);
int launch_pos_interp(const float* pos, int M, int E, int gh, int gw, double offset, int antialias, float* out,
                      hipStream_t s);
int launch_slice_tokens(const float* emb, const float* cls, const float* pos, int B, int D, int E,
                        float* xs, hipStream_t s);
int launch_slice_attn(const float* qkv, int B, int L, int heads, int hd, const uint8_t* mask,
                      const float* rope, const float* liere, float* out, float* probs, hipStream_t s);
int launch_liere_rotation(const float* vars, int n_blocks, int n, int P, float* R, hipStream_t s);
int launch_rows_copy(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int rows,
                     int cols, hipStream_t s);
int launch_saliency_accumulate(const float* maps, const float* slice_attn, int D, int heads, int gh, int gw, int Np,
                               int flip_mask, int accumulate, float* low, float* slice_acc, hipStream_t s);
int launch_saliency_upsample(const float* low, int D, int gh, int gw, float scale, int Dout, int H, int W, float* out,
                             hipStream_t s);
int launch_bmm32_nn(const float* A, const float* B, float* C, int64_t batch, int M, int N, int K, hipStream_t s);
// training step (k_train.hip)
int launch_gemm_ex(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                   int64_t scm, int64_t scn, int nb1, int nb2, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1,
                   int64_t sc2, float alpha, float beta, hipStream_t s);
int launch_softmax_rows(float* S, const uint8_t* mask, int64_t rows, int L, int rows_per_b, hipStream_t s);
int launch_softmax_rows_bwd(const float* P, float* dP, int64_t rows, int L, float scale, hipStream_t s);
int launch_layernorm_bwd(const float* x, int64_t xs, const float* gamma, const float* dy, int64_t dys, const float* dres, int64_t drs,
                         float* dx, int64_t dxs, float* dgamma, float* dbeta, int64_t rows, int cols, float eps, hipStream_t s);
int launch_act_fwd(const float* h, float* y, int64_t n, int kind, hipStream_t s);
int launch_act_bwd(const float* h, float* dy, int64_t n, int kind, hipStream_t s);
int launch_colsum(const float* a, int64_t as, const float* b, int64_t bs, int64_t rows, int cols, float* out, hipStream_t s);
int launch_axpby_cols(const float* x, int64_t xs, const float* g, float alpha, float beta, float* y, int64_t ys, int64_t rows,
                      int cols, hipStream_t s);
int launch_im2col14(const void* vol, int dt, int n, int H, int W, float* col, hipStream_t s);
int launch_pos_interp_bwd(const float* dout, int M, int E, int gh, int gw, double offset, float* dpos, hipStream_t s);
// convolutional backbone (k_conv.hip)
int launch_im2col_nhwc(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* col,
                       hipStream_t s);
int launch_maxpool_nhwc(const float* x, int n, int H, int W, int C, float* y, hipStream_t s);
int launch_avgpool_nhwc(const float* x, int n, int HW, int C, float* y, hipStream_t s);
int launch_colsqdev(const float* z, const float* mean, int64_t rows, int C, float* out, hipStream_t s);
int launch_bn_finalize(int stage, const float* acc, int64_t rows, int C, float eps, float momentum, float* mean, float* rstd,
                       float* running_mean, float* running_var, hipStream_t s);
int launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* res,
                    int relu, int64_t rows, int C, float* y, hipStream_t s);
int launch_bn_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* dy, int64_t rows, int C,
                  float* dgamma, float* dbeta, float* dz, hipStream_t s);
int launch_col2im_nhwc(const float* dcol, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* dx,
                       hipStream_t s);
int launch_maxpool_bwd_nhwc(const float* x, const float* dy, int n, int H, int W, int C, float* dx, hipStream_t s);
int launch_avgpool_bwd_nhwc(const float* dy, int n, int HW, int C, float* dx, hipStream_t s);
int launch_gradcampp(const float* act, const float* out, int O, const float* W, int n, int HW, int C, float* cam, float* state,
                     hipStream_t s);
size_t znorm_state_bytes(void);
int launch_pad_axis(float* v, int n0, int n1, int n2, int axis, int lo, int hi, int a0, int a1, int b0, int b1, int use_const,
                    float cval, hipStream_t s);
int launch_copy_block(const float* src, int sn1, int sn2, int s0, int s1, int s2, float* dst, int dn1, int dn2, int d0, int d1,
                      int d2, int c0, int c1, int c2, hipStream_t s);
int launch_znorm(const float* x, int64_t n, float q_lo, float q_hi, float* y, void* state, hipStream_t s);
int launch_slices2rgb(const void* vol, int dt, int B, int D, int H, int W, void* out, hipStream_t s);
int launch_mean_slices(const float* x, int B, int D, int E, float* out, hipStream_t s);
int launch_readout(const float* cls_probs, const float* slice_probs, int B, int D, int heads, int N,
                   int R, int sheads, float* plane, float* slice_attn, float* maps, hipStream_t s);
