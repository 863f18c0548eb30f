// C ABI of libmst_hip.so (include/mst_hip.h): argument checking, dtype dispatch and the two
// orchestrators -- mst_vit_encode (per-slice DINOv2 ViT) and mst_slice_fusion (across-slice
// transformer + head).  Nothing here allocates or synchronises: every launch goes to the caller's
// stream, scratch comes from the caller's workspace, so a caller may capture a call into a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "mst_common.h"

static thread_local char g_err[512] = "";

void mst_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mst_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mst_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MST_ELAUNCH;
    }
    return MST_OK;
}

void mst_allow_lds(const void* kernel, int bytes, mst_lds_once* slot) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;   // unknown / far device: always set
    std::lock_guard<std::mutex> lk(mu);
    if (dev != 63 && (slot->done_mask >> dev) & 1ull) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    slot->done_mask |= 1ull << dev;
}

int mst_persistent_grid(void) {
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus[dev] = n / 8 * 8;
    }
    return cus[dev];
}

// ---- optional per-kernel event timing (bench only): state lives in a caller-owned mst_profiler -------------
struct mst_profiler {
    struct Rec { hipEvent_t a, b; int kind; };
    std::mutex mu;
    std::vector<Rec> used, idle;
};
namespace {
const char* const kKindNames[MST_K_COUNT] = {"patch_embed", "layernorm", "gemm_qkv", "attention",
                                             "gemm_proj", "gemm_fc1", "gemm_fc2", "cls_probs", "mlp_fused", "block_fused"};
struct ProfScope {
    mst_profiler* p;
    mst_profiler::Rec r{};
    hipStream_t s;
    ProfScope(mst_profiler* prof, int kind, hipStream_t st) : p(prof), s(st) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (!p->idle.empty()) { r = p->idle.back(); p->idle.pop_back(); }
            else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
        }
        r.kind = kind;
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!p) return;
        (void)hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> lk(p->mu);
        p->used.push_back(r);
    }
};
}  // namespace

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t dt_size(int dt) { return dt == MST_F32 ? 4 : 2; }

extern "C" {

int mst_version(void) { return 300; }
const char* mst_last_error(void) { return g_err; }

int mst_layernorm(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* out,
                  int out_dtype, int64_t out_stride, int64_t rows, int cols, float eps, mst_stream_t stream) {
    MST_CHECK_ARG(x && out && ((gamma != nullptr) == (beta != nullptr)), "layernorm: null pointer");
    return launch_layernorm(x, x_stride, gamma, beta, out, out_dtype, out_stride, rows, cols, eps, (hipStream_t)stream);
}

int mst_gemm(const void* A, int ab_dtype, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
             int c_dtype, int64_t ldc, int64_t M, int N, int K, int epilogue, const float* gamma,
             float col_scale, int scale_cols, mst_stream_t stream) {
    MST_CHECK_ARG(A && W && C, "gemm: null pointer");
    if (ab_dtype == MST_F32) {
        MST_CHECK_ARG(c_dtype == MST_F32, "gemm: f32 operands need f32 C");
        return launch_gemm32((const float*)A, lda, (const float*)W, ldw, bias, (float*)C, ldc, M, N, K, epilogue,
                             gamma, col_scale, scale_cols, (hipStream_t)stream);
    }
    return launch_gemm16(A, ab_dtype, lda, W, ldw, bias, C, c_dtype, ldc, M, N, K, epilogue, gamma, col_scale,
                         scale_cols, (hipStream_t)stream);
}

int mst_quantize_fp8(const void* x, int dtype, int64_t n, float* amax, void* out8, mst_stream_t stream) {
    MST_CHECK_ARG(x && amax && out8, "quantize_fp8: null pointer");
    return launch_quant8(x, dtype, n, amax, out8, 1, (hipStream_t)stream);
}

int mst_gemm_fp8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* bias, const float* a_amax,
                 float w_scale, void* C, int c_dtype, int64_t ldc, int64_t M, int N, int K, int epilogue,
                 const float* gamma, float col_scale, int scale_cols, const float* c_amax, mst_stream_t stream) {
    return launch_gemm8(A8, lda, W8, ldw, bias, a_amax, w_scale, C, c_dtype, ldc, M, N, K, epilogue, gamma, col_scale,
                        scale_cols, nullptr, c_dtype == MST_F8E4M3 ? c_amax : nullptr, (hipStream_t)stream);
}

int mst_layernorm_fp8(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* out8,
                      int64_t out_stride, int64_t rows, int cols, float eps, const float* amax, mst_stream_t stream) {
    return launch_layernorm_f8(x, x_stride, gamma, beta, out8, out_stride, rows, cols, eps, amax, (hipStream_t)stream);
}

int mst_attention(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim, void* out,
                  mst_stream_t stream) {
    MST_CHECK_ARG(qkv && out, "attention: null pointer");
    MST_CHECK_ARG(head_dim == 64, "attention: head_dim=%d unsupported (64)", head_dim);
    if (dtype == MST_F32) return launch_attn32((const float*)qkv, n_seq, N, heads, (float*)out, (hipStream_t)stream);
    return launch_attn16(qkv, dtype, n_seq, N, heads, out, 0, (hipStream_t)stream);
}

int mst_attention_cls_probs(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim, float* probs,
                            mst_stream_t stream) {
    MST_CHECK_ARG(qkv && probs, "cls_probs: null pointer");
    return launch_cls_probs(qkv, dtype, n_seq, N, heads, head_dim, probs, 0, (hipStream_t)stream);
}

int mst_attention_probs_full(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim, float* probs,
                             mst_stream_t stream) {
    MST_CHECK_ARG(qkv && probs, "probs_full: null pointer");
    return launch_probs_full(qkv, dtype, n_seq, N, heads, head_dim, probs, 0, (hipStream_t)stream);
}

int mst_pos_embed_interp(const float* pos_patch, int M, int E, int gh, int gw, double offset, int antialias, float* out,
                         mst_stream_t stream) {
    MST_CHECK_ARG(pos_patch && out, "pos_embed_interp: null pointer");
    return launch_pos_interp(pos_patch, M, E, gh, gw, offset, antialias, out, (hipStream_t)stream);
}

int mst_patch_embed(const void* vol, int in_dtype, int n, int H, int W, const void* wp, int dtype, const float* bias,
                    const float* prefix, int n_prefix, const float* pos_patch, int E, float* x, mst_stream_t stream) {
    MST_CHECK_ARG(vol && wp && bias && prefix && pos_patch && x, "patch_embed: null pointer");
    return launch_patch_embed(vol, in_dtype, n, H, W, wp, dtype, bias, prefix, n_prefix, pos_patch, E, x,
                              (hipStream_t)stream);
}

int mst_mlp_fused(float* x, void* xn_out, int dtype, const void* wpack, const float* b1f, const float* b2f,
                  int64_t M, int E, float eps, mst_stream_t stream) {
    MST_CHECK_ARG(x && wpack && b1f && b2f, "mlp_fused: null pointer");
    return launch_mlp16(x, xn_out, dtype, wpack, b1f, b2f, M, E, eps, (hipStream_t)stream);
}

// ---- training step: per-op entry points (the backward is orchestrated by mst/train.py) ---------------------------------
int mst_gemm_ex(const float* A, const float* B, float* C, int M, int N, int K, const int64_t* st, int nb1, int nb2, float alpha,
                float beta, mst_stream_t stream) {
    MST_CHECK_ARG(st, "gemm_ex: null strides");
    return launch_gemm_ex(A, B, C, M, N, K, st[0], st[1], st[2], st[3], st[4], st[5], nb1, nb2, st[6], st[7], st[8], st[9], st[10],
                          st[11], alpha, beta, (hipStream_t)stream);
}
int mst_cvt16(const float* x, int64_t ldx, int64_t rows, int cols, float scale, void* out, int out_dtype, int64_t ldo, int transpose,
              int64_t rows_pad, mst_stream_t stream) {
    return launch_cvt16(x, ldx, rows, cols, scale, out, out_dtype, ldo, transpose, rows_pad, (hipStream_t)stream);
}
int mst_gemm16_splitk(const void* A, int ab_dtype, int64_t lda, const void* W, int64_t ldw, float* Cpart, int64_t ldc, int64_t M, int N, int K,
                      int splits, int64_t split_stride, mst_stream_t stream) {
    return launch_gemm16_splitk(A, ab_dtype, lda, W, ldw, Cpart, ldc, M, N, K, splits, split_stride, (hipStream_t)stream);
}
int mst_softmax_rows(float* S, const uint8_t* mask, int64_t rows, int L, int rows_per_batch, mst_stream_t stream) {
    MST_CHECK_ARG(S && rows > 0 && L > 0 && rows_per_batch > 0, "softmax_rows: bad arguments");
    return launch_softmax_rows(S, mask, rows, L, rows_per_batch, (hipStream_t)stream);
}
int mst_softmax_rows_bwd(const float* P, float* dP, int64_t rows, int L, float scale, mst_stream_t stream) {
    MST_CHECK_ARG(P && dP && rows > 0 && L > 0, "softmax_rows_bwd: bad arguments");
    return launch_softmax_rows_bwd(P, dP, rows, L, scale, (hipStream_t)stream);
}
int mst_layernorm_bwd(const float* x, int64_t x_stride, const float* gamma, const float* dy, int64_t dy_stride, const float* dres,
                      int64_t dres_stride, float* dx, int64_t dx_stride, float* dgamma, float* dbeta, int64_t rows, int cols,
                      float eps, mst_stream_t stream) {
    MST_CHECK_ARG(x && dy && rows > 0, "layernorm_bwd: bad arguments");
    return launch_layernorm_bwd(x, x_stride, gamma, dy, dy_stride, dres, dres_stride, dx, dx_stride, dgamma, dbeta, rows, cols, eps,
                                (hipStream_t)stream);
}
int mst_act_fwd(const float* h, float* y, int64_t n, int kind, mst_stream_t stream) {
    MST_CHECK_ARG(h && y && n > 0 && (kind == 0 || kind == 1), "act_fwd: bad arguments");
    return launch_act_fwd(h, y, n, kind, (hipStream_t)stream);
}
int mst_act_bwd(const float* h, float* dy, int64_t n, int kind, mst_stream_t stream) {
    MST_CHECK_ARG(h && dy && n > 0 && (kind == 0 || kind == 1), "act_bwd: bad arguments");
    return launch_act_bwd(h, dy, n, kind, (hipStream_t)stream);
}
int mst_colsum(const float* a, int64_t a_stride, const float* b, int64_t b_stride, int64_t rows, int cols, float* out,
               mst_stream_t stream) {
    MST_CHECK_ARG(a && out && rows > 0 && cols > 0, "colsum: bad arguments");
    return launch_colsum(a, a_stride, b, b_stride, rows, cols, out, (hipStream_t)stream);
}
int mst_axpby_cols(const float* x, int64_t x_stride, const float* g, float alpha, float beta, float* y, int64_t y_stride, int64_t rows,
                   int cols, mst_stream_t stream) {
    MST_CHECK_ARG(x && y && rows > 0 && cols > 0, "axpby_cols: bad arguments");
    return launch_axpby_cols(x, x_stride, g, alpha, beta, y, y_stride, rows, cols, (hipStream_t)stream);
}
int mst_im2col14(const void* vol, int dtype, int n, int H, int W, float* col, mst_stream_t stream) {
    MST_CHECK_ARG(vol && col && n > 0 && H > 0 && W > 0 && H % 14 == 0 && W % 14 == 0, "im2col14: bad arguments");
    return launch_im2col14(vol, dtype, n, H, W, col, (hipStream_t)stream);
}
int mst_pos_embed_interp_bwd(const float* dout, int M, int E, int gh, int gw, double offset, float* dpos, mst_stream_t stream) {
    MST_CHECK_ARG(dout && dpos && M > 0 && E > 0 && gh > 0 && gw > 0, "pos_embed_interp_bwd: bad arguments");
    return launch_pos_interp_bwd(dout, M, E, gh, gw, offset, dpos, (hipStream_t)stream);
}

// ---- convolutional backbone of the ResNet models ------------------------------------------------------------
int mst_im2col_nhwc(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* col,
                    mst_stream_t stream) {
    MST_CHECK_ARG(x && col && n > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "im2col_nhwc: bad arguments");
    return launch_im2col_nhwc(x, n, H, W, C, kh, kw, stride, pad, Kpad, col, (hipStream_t)stream);
}
int mst_conv_gemm(const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const float* Wg, const float* bias,
                  float* out, int Cout, int Kpad, int epilogue, const float* gamma, mst_stream_t stream) {
    return launch_conv_gemm32(x, n, H, W, Cin, kh, kw, stride, pad, Wg, Kpad, bias, out, Cout, Cout, Kpad, epilogue, gamma, (hipStream_t)stream);
}
int mst_conv_gemm16(const void* x, int dtype, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const void* Wg, const float* bias,
                    void* out, int out_dtype, int Cout, int epilogue, mst_stream_t stream) {
    return launch_conv_gemm16(x, dtype, n, H, W, Cin, kh, kw, stride, pad, Wg, bias, out, out_dtype, Cout, epilogue, (hipStream_t)stream);
}
int mst_conv_dgrad(const void* dz, int dtype, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const void* Wt, int H, int W,
                   int Cin, float* dx, mst_stream_t stream) {
    if (dtype == MST_F32)
        return launch_conv_dgrad32((const float*)dz, n, Ho, Wo, Cout, kh, kw, stride, pad, (const float*)Wt, H, W, Cin, dx, (hipStream_t)stream);
    return launch_conv_dgrad16(dz, dtype, n, Ho, Wo, Cout, kh, kw, stride, pad, Wt, H, W, Cin, dx, (hipStream_t)stream);
}
int mst_conv_wgrad(const float* dz, const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                   int nsplit, int64_t rows_per_split, mst_stream_t stream) {
    return launch_conv_wgrad32(dz, x, n, H, W, Cin, kh, kw, stride, pad, Cout, part, nsplit, rows_per_split, (hipStream_t)stream);
}
int mst_conv_wgrad16(const void* dz, const void* x, int dtype, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                     int nsplit, int64_t rows_per_split, mst_stream_t stream) {
    return launch_conv_wgrad16(dz, x, dtype, n, H, W, Cin, kh, kw, stride, pad, Cout, part, nsplit, rows_per_split, (hipStream_t)stream);
}
int mst_rope_rows(float* qkv, int64_t rows, int L, int heads, int head_dim, const float* freqs, float sign, mst_stream_t stream) {
    return launch_rope_rows(qkv, rows, L, heads, head_dim, freqs, sign, (hipStream_t)stream);
}
int mst_im2col_nhwc16(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, void* col, int out_dtype,
                      mst_stream_t stream) {
    MST_CHECK_ARG(x && col && n > 0, "im2col16: bad arguments");
    return launch_im2col_nhwc16(x, n, H, W, C, kh, kw, stride, pad, Kpad, col, out_dtype, (hipStream_t)stream);
}
int mst_maxpool_nhwc16(const void* x, int dtype, int n, int H, int W, int C, void* y, mst_stream_t stream) {
    MST_CHECK_ARG(x && y && n > 0 && H > 0 && W > 0 && C > 0, "maxpool16: bad arguments");
    return launch_maxpool_nhwc16(x, dtype, n, H, W, C, y, (hipStream_t)stream);
}
int mst_cvt32(const void* x, int dtype, int64_t n, float* out, mst_stream_t stream) { return launch_cvt32(x, dtype, n, out, (hipStream_t)stream); }
int mst_maxpool_nhwc(const float* x, int n, int H, int W, int C, float* y, mst_stream_t stream) {
    MST_CHECK_ARG(x && y && n > 0 && H > 0 && W > 0 && C > 0, "maxpool_nhwc: bad arguments");
    return launch_maxpool_nhwc(x, n, H, W, C, y, (hipStream_t)stream);
}
int mst_avgpool_nhwc(const float* x, int n, int HW, int C, float* y, mst_stream_t stream) {
    MST_CHECK_ARG(x && y && n > 0 && HW > 0 && C > 0, "avgpool_nhwc: bad arguments");
    return launch_avgpool_nhwc(x, n, HW, C, y, (hipStream_t)stream);
}

int mst_batchnorm_train(const float* z, int64_t rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                        const float* residual, int relu, float* y, float* mean, float* rstd, float* running_mean,
                        float* running_var, float* scratch, mst_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MST_CHECK_ARG(z && gamma && beta && y && mean && rstd && scratch && rows > 0 && C > 0, "batchnorm_train: bad arguments");
    if (hipMemsetAsync(scratch, 0, sizeof(float) * C, s) != hipSuccess) { mst_set_error("batchnorm_train: memset failed"); return MST_ELAUNCH; }
    int rc = launch_colsum(z, C, nullptr, 0, rows, C, scratch, s);
    if (rc) return rc;
    if ((rc = launch_bn_finalize(0, scratch, rows, C, eps, momentum, mean, rstd, nullptr, nullptr, s))) return rc;
    if (hipMemsetAsync(scratch, 0, sizeof(float) * C, s) != hipSuccess) { mst_set_error("batchnorm_train: memset failed"); return MST_ELAUNCH; }
    if ((rc = launch_colsqdev(z, mean, rows, C, scratch, s))) return rc;
    if ((rc = launch_bn_finalize(1, scratch, rows, C, eps, momentum, mean, rstd, running_mean, running_var, s))) return rc;
    return launch_bn_apply(z, mean, rstd, gamma, beta, residual, relu, rows, C, y, s);
}
int mst_batchnorm_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* dy, int64_t rows, int C,
                      float* dgamma, float* dbeta, float* dz, mst_stream_t stream) {
    MST_CHECK_ARG(z && mean && rstd && gamma && dy && dgamma && dbeta && dz && rows > 0 && C > 0, "batchnorm_bwd: bad arguments");
    return launch_bn_bwd(z, mean, rstd, gamma, dy, rows, C, dgamma, dbeta, dz, (hipStream_t)stream);
}
int mst_col2im_nhwc(const float* dcol, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* dx,
                    mst_stream_t stream) {
    MST_CHECK_ARG(dcol && dx && n > 0 && H > 0 && W > 0 && C > 0, "col2im_nhwc: bad arguments");
    return launch_col2im_nhwc(dcol, n, H, W, C, kh, kw, stride, pad, Kpad, dx, (hipStream_t)stream);
}
int mst_maxpool_bwd_nhwc(const float* x, const float* dy, int n, int H, int W, int C, float* dx, mst_stream_t stream) {
    MST_CHECK_ARG(x && dy && dx && n > 0, "maxpool_bwd_nhwc: bad arguments");
    return launch_maxpool_bwd_nhwc(x, dy, n, H, W, C, dx, (hipStream_t)stream);
}
int mst_avgpool_bwd_nhwc(const float* dy, int n, int HW, int C, float* dx, mst_stream_t stream) {
    MST_CHECK_ARG(dy && dx && n > 0 && HW > 0 && C > 0, "avgpool_bwd_nhwc: bad arguments");
    return launch_avgpool_bwd_nhwc(dy, n, HW, C, dx, (hipStream_t)stream);
}

int mst_gradcampp(const float* act, const float* out, int O, const float* W, int n, int HW, int C, float* cam, float* state,
                  mst_stream_t stream) {
    MST_CHECK_ARG(act && out && cam && state && n > 0 && HW > 0 && C > 0 && C <= 8192 && O > 0 && (W || O == C), "gradcampp: bad arguments");
    return launch_gradcampp(act, out, O, W, n, HW, C, cam, state, (hipStream_t)stream);
}

// ---- input pipeline ------------------------------------------------------------------------------------
int mst_crop_or_pad(const float* src, int s0, int s1, int s2, float* dst, int t0, int t1, int t2, int pad_minimum,
                    float pad_value, void* ws, size_t ws_bytes, mst_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MST_CHECK_ARG(src && dst && s0 > 0 && s1 > 0 && s2 > 0 && t0 > 0 && t1 > 0 && t2 > 0, "crop_or_pad: bad arguments");
    const int sn[3] = {s0, s1, s2}, tn[3] = {t0, t1, t2};
    int pad_lo[3], pad_hi[3], crop_lo[3], pn[3];
    bool any_pad = false;
    for (int a = 0; a < 3; ++a) {                       // augmentations_3d.py:164-172: ini = ceil(n / 2), fin = n - ini
        const int d = tn[a] - sn[a];
        const int p = d > 0 ? d : 0, c = d < 0 ? -d : 0;
        pad_lo[a] = (p + 1) / 2; pad_hi[a] = p - pad_lo[a];
        crop_lo[a] = (c + 1) / 2;
        pn[a] = sn[a] + p;
        any_pad = any_pad || p > 0;
    }
    if (!any_pad)                                        // crop only
        return launch_copy_block(src, s1, s2, crop_lo[0], crop_lo[1], crop_lo[2], dst, t1, t2, 0, 0, 0, t0, t1, t2, s);
    const bool direct = pn[0] == t0 && pn[1] == t1 && pn[2] == t2;    // pad only: build the padded array in dst itself
    const size_t need = direct ? 0 : sizeof(float) * (size_t)pn[0] * pn[1] * pn[2];
    MST_CHECK_ARG(direct || (ws && ws_bytes >= need), "crop_or_pad: workspace %zu < %zu bytes", ws_bytes, need);
    float* pad = direct ? dst : (float*)ws;
    int rc = launch_copy_block(src, s1, s2, 0, 0, 0, pad, pn[1], pn[2], pad_lo[0], pad_lo[1], pad_lo[2], s0, s1, s2, s);
    if (rc) return rc;
    // numpy.pad: one axis after the other; axis a sees the pads of the axes before it and only the valid part of the later ones
    const int lo0 = pad_lo[0], hi0 = pad_lo[0] + s0, lo1 = pad_lo[1], hi1 = pad_lo[1] + s1, lo2 = pad_lo[2], hi2 = pad_lo[2] + s2;
    if (pn[0] > s0 && (rc = launch_pad_axis(pad, pn[0], pn[1], pn[2], 0, lo0, hi0, lo1, hi1, lo2, hi2, !pad_minimum, pad_value, s))) return rc;
    if (pn[1] > s1 && (rc = launch_pad_axis(pad, pn[0], pn[1], pn[2], 1, lo1, hi1, 0, pn[0], lo2, hi2, !pad_minimum, pad_value, s))) return rc;
    if (pn[2] > s2 && (rc = launch_pad_axis(pad, pn[0], pn[1], pn[2], 2, lo2, hi2, 0, pn[0], 0, pn[1], !pad_minimum, pad_value, s))) return rc;
    if (direct) return MST_OK;
    return launch_copy_block(pad, pn[1], pn[2], crop_lo[0], crop_lo[1], crop_lo[2], dst, t1, t2, 0, 0, 0, t0, t1, t2, s);
}

size_t mst_znorm_state_bytes(void) { return znorm_state_bytes(); }

int mst_znorm(const float* x, int64_t n, float q_lo, float q_hi, float* y, void* state, mst_stream_t stream) {
    MST_CHECK_ARG(x && y && state && n > 0 && q_lo >= 0.f && q_lo <= q_hi && q_hi <= 1.f, "znorm: bad arguments");
    return launch_znorm(x, n, q_lo, q_hi, y, state, (hipStream_t)stream);
}

int mst_slices2rgb(const void* vol, int dtype, int B, int D, int H, int W, void* out, mst_stream_t stream) {
    MST_CHECK_ARG(vol && out && B > 0 && D > 0 && H > 0 && W > 0, "slices2rgb: bad arguments");
    MST_CHECK_ARG(dtype == MST_F32 || dtype == MST_F16 || dtype == MST_BF16, "slices2rgb: bad dtype %d", dtype);
    return launch_slices2rgb(vol, dtype, B, D, H, W, out, (hipStream_t)stream);
}

size_t mst_block_fused_scratch_bytes(void) { return block16_scratch_bytes(); }

int mst_block_fused(float* x, const void* attn_out, void* xn_out, int dtype, const void* proj_pack, const float* proj_bf,
                    const void* wpack, const float* b1f, const float* b2f, void* scratch, size_t scratch_bytes, int64_t M,
                    int E, float eps, mst_stream_t stream) {
    MST_CHECK_ARG(x && attn_out && proj_pack && proj_bf && wpack && b1f && b2f && scratch, "block_fused: null pointer");
    MST_CHECK_ARG(scratch_bytes >= block16_scratch_bytes(), "block_fused: scratch %zu < %zu bytes", scratch_bytes, block16_scratch_bytes());
    return launch_block16(x, attn_out, xn_out, dtype, proj_pack, proj_bf, wpack, b1f, b2f, scratch, M, E, eps, (hipStream_t)stream);
}

int mst_block_fused_s(float* x, const void* attn_out, void* xn_out, int dtype, const void* block_seq, const float* b1f,
                      const float* proj_bf, const float* b2f, int64_t M, int E, float eps, int layout, mst_stream_t stream) {
    MST_CHECK_ARG(x && attn_out && block_seq && b1f && proj_bf && b2f, "block_fused_s: null pointer");
    MST_CHECK_ARG(layout >= 0 && layout < 8, "block_fused_s: layout=%d", layout);
    return launch_block16s(x, attn_out, xn_out, dtype, block_seq, b1f, proj_bf, b2f, M, E, eps, layout, (hipStream_t)stream);
}

// ---- per-slice encoder ----------------------------------------------------------------------
// workspace carve (chunk of C slices, Mc = C*N rows):
//   x   fp32 [Mc, E]      residual stream
//   xn  T    [Mc, E]      LayerNorm output; reused as the attention output
//   big T    [Mc, 4E]     qkv [Mc, 3E] during attention, then the MLP hidden [Mc, 4E]
//   fp8_linear only:  a8 u8 [Mc, 4E] (the quantised input of the current GEMM) | amax f32 [depth][4]
static void vit_carve(const mst_vit_weights* w, int N, int chunk, size_t* off_xn, size_t* off_big, size_t* total,
                      size_t* off_a8 = nullptr, size_t* off_amax = nullptr, size_t* off_blk = nullptr) {
    // rows rounded up to whole 32-row groups: the blocked / image layouts between two single-role block launches address groups
    const size_t Mc = ((size_t)chunk * N + 31) / 32 * 32, E = (size_t)w->embed_dim, ts = dt_size(w->compute_dtype);
    size_t o = 0;
    o += align_up(Mc * E * 4, 256);
    *off_xn = o;
    o += align_up(Mc * E * ts, 256);
    *off_big = o;
    o += align_up(Mc * 4 * E * ts, 256);
    if (w->fp8_linear) {
        if (off_a8) *off_a8 = o;
        o += align_up(Mc * 4 * E, 256);
        if (off_amax) *off_amax = o;
        o += align_up((size_t)w->depth * 4 * sizeof(float), 256);
    }
    if (off_blk) *off_blk = o;
    o += align_up(block16_scratch_bytes(), 256);        // lane-private LayerNorm2 hand-off of the fused block kernel (96 KiB per CU)
    *total = o;
}

size_t mst_vit_workspace_bytes(const mst_vit_weights* w, int H, int W, int chunk_slices) {
    if (!w || H <= 0 || W <= 0 || chunk_slices <= 0) return 0;
    const int N = 1 + w->num_registers + (H / 14) * (W / 14);
    size_t a, b, t;
    vit_carve(w, N, chunk_slices, &a, &b, &t);
    return t;
}

int mst_vit_encode(const mst_vit_weights* w, const void* vol, int in_dtype, int n_slices, int H, int W,
                   float* cls_out, float* cls_probs, float* full_probs, int n_layers_probs, int chunk_slices,
                   void* ws, size_t ws_bytes, mst_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MST_CHECK_ARG(w && vol && cls_out && ws, "vit_encode: null pointer");
    mst_profiler* const prof = w->profiler;
    MST_CHECK_ARG(w->layers && w->depth > 0, "vit_encode: no layers");
    MST_CHECK_ARG(H > 0 && W > 0 && H % 14 == 0 && W % 14 == 0, "vit_encode: H=%d W=%d must be multiples of 14", H, W);
    MST_CHECK_ARG(H / 14 == w->grid_h && W / 14 == w->grid_w, "vit_encode: pos_patch prepared for grid %dx%d, input is %dx%d",
                  w->grid_h, w->grid_w, H / 14, W / 14);
    const int E = w->embed_dim, heads = w->num_heads, dt = w->compute_dtype;
    MST_CHECK_ARG(heads > 0 && E == heads * 64, "vit_encode: embed_dim=%d must be 64*num_heads (%d)", E, heads);
    MST_CHECK_ARG(dt == MST_F16 || dt == MST_BF16 || dt == MST_F32, "vit_encode: bad compute dtype %d", dt);
    MST_CHECK_ARG(n_slices > 0 && chunk_slices > 0, "vit_encode: n_slices=%d chunk=%d", n_slices, chunk_slices);
    MST_CHECK_ARG(n_layers_probs >= 0 && n_layers_probs <= w->depth, "vit_encode: n_layers_probs=%d", n_layers_probs);
    const int R = w->num_registers, Np = w->grid_h * w->grid_w, N = 1 + R + Np;
    if (chunk_slices > n_slices) chunk_slices = n_slices;
    size_t off_xn, off_big, total, off_a8 = 0, off_amax = 0, off_blk = 0;
    vit_carve(w, N, chunk_slices, &off_xn, &off_big, &total, &off_a8, &off_amax, &off_blk);
    if (ws_bytes < total) {
        mst_set_error("vit_encode: workspace %zu < %zu bytes", ws_bytes, total);
        return MST_EWORKSPACE;
    }
    float* x = (float*)ws;
    void* xn = (char*)ws + off_xn;
    void* big = (char*)ws + off_big;
    const bool fp8 = w->fp8_linear != 0;
    const bool fp8_static = fp8 && w->fp8_amax != nullptr;   // calibrated scales: producers write e4m3, nothing is scanned
    void* a8 = fp8 ? (char*)ws + off_a8 : nullptr;
    float* amax = fp8 ? (float*)((char*)ws + off_amax) : nullptr;
    if (fp8) {
        MST_CHECK_ARG(dt == MST_F16 || dt == MST_BF16, "vit_encode: fp8_linear needs a 16-bit compute dtype");
        MST_CHECK_ARG(E % 128 == 0, "vit_encode: fp8_linear needs embed_dim %% 128 == 0 (got %d)", E);
        for (int l = 0; l < w->depth; ++l)
            MST_CHECK_ARG(w->layers[l].qkv_w8 && w->layers[l].proj_w8 && w->layers[l].fc1_w8 && w->layers[l].fc2_w8,
                          "vit_encode: fp8_linear set but layer %d has no e4m3 weights", l);
    }
    const size_t in_sz = dt_size(in_dtype);
    // head_dim^-0.5, head_dim = 64 (attention.py:48); the 16-bit attention kernels work in the log2 domain, so log2(e) rides
    // on the same fp32 multiply of the QKV epilogue (one rounding of q, not two)
    const int log2q = dt != MST_F32;
    const float qscale = log2q ? 0.125f * 1.4426950408889634f : 0.125f;

#define RUN(call)              \
    do {                       \
        int rc_ = (call);      \
        if (rc_) return rc_;   \
    } while (0)
#define RUNK(kind, call)             \
    do {                             \
        int rc_;                     \
        {                            \
            ProfScope ps_(prof, kind, s); \
            rc_ = (call);            \
        }                            \
        if (rc_) return rc_;         \
    } while (0)

    for (int s0 = 0; s0 < n_slices; s0 += chunk_slices) {
        const int c = (n_slices - s0 < chunk_slices) ? n_slices - s0 : chunk_slices;
        const int64_t Mc = (int64_t)c * N;
        const char* v = (const char*)vol + (size_t)s0 * H * W * in_sz;
        // fused-LayerNorm pipeline (16-bit, E = 384): norm1 folded into QKV, norm2 + MLP in one kernel.  The fused MLP is a
        // persistent kernel of 128-token tiles: below ~one tile per CU it leaves CUs idle (c1 shape, 4k tokens: 90 us per
        // launch on 33 CUs against ~45 us for LN + fc1 + fc2 as three well-filled launches), so small calls take the unfused
        // path.  The choice depends on the WHOLE call (n_slices x N), never on the chunking.
        static const int64_t fused_min_tokens = getenv("MST_FUSED_MIN_TOKENS") ? atoll(getenv("MST_FUSED_MIN_TOKENS")) : 12288;   // measured crossover (tools/bench_crossover.py): 8k tokens unfused 1.57 vs 1.92 ms, 16k fused 2.34 vs 2.62
        const bool prune = w->prune_last_block && !fp8 && heads * 64 == E && c <= 65535;
        bool fused = !fp8 && (dt != MST_F32) && E == 384 && (int64_t)n_slices * N >= fused_min_tokens;
        for (int l = 0; l < w->depth && fused; ++l)
            fused = w->layers[l].mlp_pack && w->layers[l].fc1_bf && w->layers[l].fc2_bf && w->layers[l].qkv_wf && w->layers[l].qkv_bf;
        // out-projection folded into the same launch (k_block16.hip) when its packed image is there (MST_NO_PROJ_FOLD=1: A/B switch)
        static const bool no_fold = getenv("MST_NO_PROJ_FOLD") && atoi(getenv("MST_NO_PROJ_FOLD"));
        bool folded = fused && !no_fold;
        for (int l = 0; l < w->depth && folded; ++l) folded = w->layers[l].proj_pack && w->layers[l].proj_bf;
        // round 3: the single-role block kernel when its weight stream is there (MST_BLOCK_ROLES=1: the producer/consumer form, A/B switch)
        static const bool roles = getenv("MST_BLOCK_ROLES") && atoi(getenv("MST_BLOCK_ROLES"));
        bool single_role = folded && !roles;
        for (int l = 0; l < w->depth && single_role; ++l) single_role = w->layers[l].block_seq != nullptr;
        // between two single-role block launches the rows stay in the order the registers hold them (include/mst_hip.h, mst_layout_flags):
        // x as the fp32 image from block 0's output to the last block's input, the 16-bit rows (attention output, normalised rows)
        // blocked from block 0's attention on.  Needs the QKV kernel that reads blocked rows (weights-in-registers form) and every
        // block on the fused path (MST_BLOCK_ROWMAJOR=1: row-major everywhere, A/B switch).
        static const bool rowmajor = getenv("MST_BLOCK_ROWMAJOR") && atoi(getenv("MST_BLOCK_ROWMAJOR"));
        const bool blocked = single_role && !rowmajor && !prune && (dt == MST_F16 || dt == MST_BF16) &&
                             gemm16_wreg_applicable(Mc, 3 * E, E, dt, dt, MST_EPI_BIAS, E, E, 3 * E);
        // tokens: in the fused pipeline one kernel writes the residual stream AND block 0's plain-normalised rows (k_patch_rows.hip;
        // MST_PATCH_ROWS=0: the tiled patch kernel + a LayerNorm launch, the A/B baseline)
        static const bool patch_rows = !(getenv("MST_PATCH_ROWS") && atoi(getenv("MST_PATCH_ROWS")) == 0);
        if (fused && patch_rows) {
            RUNK(MST_K_PATCH_EMBED, launch_patch_rows16(v, in_dtype, c, H, W, w->patch_w, dt, w->patch_b, w->prefix, 1 + R, w->pos_patch, x, xn, s));
        } else {
            RUNK(MST_K_PATCH_EMBED, launch_patch_embed(v, in_dtype, c, H, W, w->patch_w, dt, w->patch_b, w->prefix, 1 + R, w->pos_patch, E, x, s));
            if (fused) RUNK(MST_K_LAYERNORM, launch_layernorm(x, E, nullptr, nullptr, xn, dt, E, Mc, E, 1e-6f, s));
        }
        if (fp8 && !fp8_static && hipMemsetAsync(amax, 0, (size_t)w->depth * 4 * sizeof(float), s) != hipSuccess) {
            mst_set_error("vit_encode: hipMemsetAsync(amax) failed");
            return MST_ELAUNCH;
        }
        for (int l = 0; l < w->depth; ++l) {
            const mst_vit_layer* L = &w->layers[l];
            // x += ls1(proj(attn(qkv(norm1 x))))                       block.py:90-91,112
            if (fp8_static) {
                const float* am = w->fp8_amax + l * 4;
                RUNK(MST_K_LAYERNORM, launch_layernorm_f8(x, E, L->ln1_w, L->ln1_b, a8, E, Mc, E, 1e-6f, am + 0, s));
                RUNK(MST_K_GEMM_QKV, launch_gemm8(a8, E, L->qkv_w8, E, L->qkv_b, am + 0, L->w8_scale[0], big, dt, 3 * E, Mc, 3 * E, E, MST_EPI_BIAS, nullptr, qscale, E, nullptr, nullptr, s));
            } else if (fp8) {
                // every linear layer: quantise its input with a fresh per-tensor scale (this chunk's max|x|), e4m3 GEMM.  The
                // [M,E] inputs are scanned (53 us; a running maximum kept by the one-wave-per-row LayerNorm kernel cost 125 us:
                // 350 k waves polling one address serialise in one L2 channel); the 4x larger hidden activation gets its maximum
                // from the fc1 epilogue (one conditional atomic per workgroup)
                float* am = amax + l * 4;
                RUNK(MST_K_LAYERNORM, launch_layernorm(x, E, L->ln1_w, L->ln1_b, xn, dt, E, Mc, E, 1e-6f, s));
                RUN(launch_quant8(xn, dt, Mc * E, am + 0, a8, 1, s));
                RUNK(MST_K_GEMM_QKV, launch_gemm8(a8, E, L->qkv_w8, E, L->qkv_b, am + 0, L->w8_scale[0], big, dt, 3 * E, Mc, 3 * E, E, MST_EPI_BIAS, nullptr, qscale, E, nullptr, nullptr, s));
            } else if (fused && blocked && l > 0) {
                RUNK(MST_K_GEMM_QKV, launch_gemm16_wreg(xn, dt, E, L->qkv_wf, E, L->qkv_bf, big, 3 * E, Mc, 3 * E, qscale, E, s, 1));
            } else if (fused) {
                RUNK(MST_K_GEMM_QKV, mst_gemm(xn, dt, E, L->qkv_wf, E, L->qkv_bf, big, dt, 3 * E, Mc, 3 * E, E, MST_EPI_BIAS, nullptr, qscale, E, s));
            } else {
                RUNK(MST_K_LAYERNORM, launch_layernorm(x, E, L->ln1_w, L->ln1_b, xn, dt, E, Mc, E, 1e-6f, s));
                RUNK(MST_K_GEMM_QKV, mst_gemm(xn, dt, E, L->qkv_w, E, L->qkv_b, big, dt, 3 * E, Mc, 3 * E, E, MST_EPI_BIAS, nullptr, qscale, E, s));
            }
            const int li = l - (w->depth - n_layers_probs);
            if (full_probs && li >= 0)
                RUN(launch_probs_full(big, dt, c, N, heads, 64, full_probs + ((int64_t)li * n_slices + s0) * heads * N * N, log2q, s));
            if (prune && l == w->depth - 1) {
                // nothing behind this block reads a patch token (final norm + head take the CLS rows): attention, out-projection
                // and MLP for the c CLS rows only, on the unfused kernels.  Scratch: xn[0, cE) attention rows, xn[cE, 2cE)
                // LayerNorm2 rows, big (free once K and V are consumed) the hidden rows.
                const size_t esz = dt == MST_F32 ? 4 : 2;
                char* const a_cls = (char*)xn;
                char* const n_cls = (char*)xn + (size_t)c * E * esz;
                float* const pr = (cls_probs && li >= 0) ? cls_probs + ((int64_t)li * n_slices + s0) * heads * N : nullptr;
                RUNK(MST_K_ATTENTION, launch_cls_attn(big, dt, c, N, heads, pr, a_cls, log2q, s));
                RUNK(MST_K_GEMM_PROJ, mst_gemm(a_cls, dt, E, L->proj_w, E, L->proj_b, x, MST_F32, (int64_t)N * E, c, E, E, MST_EPI_RESIDUAL, L->ls1, 1.f, 0, s));
                RUNK(MST_K_LAYERNORM, launch_layernorm(x, (int64_t)N * E, L->ln2_w, L->ln2_b, n_cls, dt, E, c, E, 1e-6f, s));
                RUNK(MST_K_GEMM_FC1, mst_gemm(n_cls, dt, E, L->fc1_w, E, L->fc1_b, big, dt, 4 * E, c, 4 * E, E, MST_EPI_BIAS_GELU, nullptr, 1.f, 0, s));
                RUNK(MST_K_GEMM_FC2, mst_gemm(big, dt, 4 * E, L->fc2_w, 4 * E, L->fc2_b, x, MST_F32, (int64_t)N * E, c, E, 4 * E, MST_EPI_RESIDUAL, L->ls2, 1.f, 0, s));
                continue;
            }
            if (cls_probs && li >= 0)
                RUNK(MST_K_CLS_PROBS, launch_cls_probs(big, dt, c, N, heads, 64, cls_probs + ((int64_t)li * n_slices + s0) * heads * N, log2q, s));
            if (dt == MST_F32) RUNK(MST_K_ATTENTION, launch_attn32((const float*)big, c, N, heads, (float*)xn, s));
            else RUNK(MST_K_ATTENTION, launch_attn16(big, dt, c, N, heads, xn, 1, s, blocked ? 1 : 0));
            if (fp8_static) {
                // the e4m3 hidden activation lives in `big` (bytes); fc1 reads a8 and writes big, fc2 reads big
                const float* am = w->fp8_amax + l * 4;
                RUN(launch_quant8_static(xn, dt, Mc * E, am + 1, a8, s));
                RUNK(MST_K_GEMM_PROJ, launch_gemm8(a8, E, L->proj_w8, E, L->proj_b, am + 1, L->w8_scale[1], x, MST_F32, E, Mc, E, E, MST_EPI_RESIDUAL, L->ls1, 1.f, 0, nullptr, nullptr, s));
                RUNK(MST_K_LAYERNORM, launch_layernorm_f8(x, E, L->ln2_w, L->ln2_b, a8, E, Mc, E, 1e-6f, am + 2, s));
                RUNK(MST_K_GEMM_FC1, launch_gemm8(a8, E, L->fc1_w8, E, L->fc1_b, am + 2, L->w8_scale[2], big, MST_F8E4M3, 4 * E, Mc, 4 * E, E, MST_EPI_BIAS_GELU, nullptr, 1.f, 0, nullptr, am + 3, s));
                RUNK(MST_K_GEMM_FC2, launch_gemm8(big, 4 * E, L->fc2_w8, 4 * E, L->fc2_b, am + 3, L->w8_scale[3], x, MST_F32, E, Mc, E, 4 * E, MST_EPI_RESIDUAL, L->ls2, 1.f, 0, nullptr, nullptr, s));
                continue;
            }
            if (fp8) {
                float* am = amax + l * 4;
                RUN(launch_quant8(xn, dt, Mc * E, am + 1, a8, 1, s));
                RUNK(MST_K_GEMM_PROJ, launch_gemm8(a8, E, L->proj_w8, E, L->proj_b, am + 1, L->w8_scale[1], x, MST_F32, E, Mc, E, E, MST_EPI_RESIDUAL, L->ls1, 1.f, 0, nullptr, nullptr, s));
                RUNK(MST_K_LAYERNORM, launch_layernorm(x, E, L->ln2_w, L->ln2_b, xn, dt, E, Mc, E, 1e-6f, s));
                RUN(launch_quant8(xn, dt, Mc * E, am + 2, a8, 1, s));
                RUNK(MST_K_GEMM_FC1, launch_gemm8(a8, E, L->fc1_w8, E, L->fc1_b, am + 2, L->w8_scale[2], big, dt, 4 * E, Mc, 4 * E, E, MST_EPI_BIAS_GELU, nullptr, 1.f, 0, am + 3, nullptr, s));
                RUN(launch_quant8(big, dt, Mc * 4 * E, am + 3, a8, 0, s));
                RUNK(MST_K_GEMM_FC2, launch_gemm8(a8, 4 * E, L->fc2_w8, 4 * E, L->fc2_b, am + 3, L->w8_scale[3], x, MST_F32, E, Mc, E, 4 * E, MST_EPI_RESIDUAL, L->ls2, 1.f, 0, nullptr, nullptr, s));
                continue;
            }
            if (folded && single_role) {
                // the same in the single-role form (k_block16s.hip): rows stay in the registers of the wave that owns them
                RUNK(MST_K_BLOCK_FUSED, launch_block16s(x, xn, (l + 1 < w->depth) ? xn : nullptr, dt, L->block_seq, L->fc1_bf, L->proj_bf, L->fc2_bf, Mc, E, 1e-6f,
                                                       blocked ? (MST_LAYOUT_ACT_BLOCKED | (l > 0 ? MST_LAYOUT_X_IN_IMAGE : 0) | (l + 1 < w->depth ? MST_LAYOUT_X_OUT_IMAGE : 0)) : 0, s));
                continue;
            }
            if (folded) {
                // x += ls1(proj(attn)); x += ls2(fc2(gelu(fc1(norm2 x)))); xn = normalise(x): one launch, x read and written once
                RUNK(MST_K_BLOCK_FUSED, launch_block16(x, xn, (l + 1 < w->depth) ? xn : nullptr, dt, L->proj_pack, L->proj_bf, L->mlp_pack,
                                                     L->fc1_bf, L->fc2_bf, (char*)ws + off_blk, Mc, E, 1e-6f, s));
                continue;
            }
            RUNK(MST_K_GEMM_PROJ, mst_gemm(xn, dt, E, L->proj_w, E, L->proj_b, x, MST_F32, E, Mc, E, E, MST_EPI_RESIDUAL, L->ls1, 1.f, 0, s));
            // x += ls2(fc2(gelu(fc1(norm2 x))))                        block.py:93-94,113
            if (fused) {
                RUNK(MST_K_MLP_FUSED, launch_mlp16(x, (l + 1 < w->depth) ? xn : nullptr, dt, L->mlp_pack, L->fc1_bf, L->fc2_bf, Mc, E, 1e-6f, s));
            } else {
                RUNK(MST_K_LAYERNORM, launch_layernorm(x, E, L->ln2_w, L->ln2_b, xn, dt, E, Mc, E, 1e-6f, s));
                RUNK(MST_K_GEMM_FC1, mst_gemm(xn, dt, E, L->fc1_w, E, L->fc1_b, big, dt, 4 * E, Mc, 4 * E, E, MST_EPI_BIAS_GELU, nullptr, 1.f, 0, s));
                RUNK(MST_K_GEMM_FC2, mst_gemm(big, dt, 4 * E, L->fc2_w, 4 * E, L->fc2_b, x, MST_F32, E, Mc, E, 4 * E, MST_EPI_RESIDUAL, L->ls2, 1.f, 0, s));
            }
        }
        if (fp8 && !fp8_static && w->fp8_amax_out) RUN(launch_amax_merge(w->fp8_amax_out, amax, w->depth * 4, s));
        // final norm, CLS rows only (vision_transformer.py:263-265,329)
        RUN(launch_layernorm(x, (int64_t)N * E, w->norm_w, w->norm_b, cls_out + (int64_t)s0 * E, MST_F32, E, c, E, 1e-6f, s));
    }
    return MST_OK;
}

// ---- across-slice transformer + head ----------------------------------------------------------
// workspace carve (fp32): eb [B*D, emb] (bottleneck out) | xs [B*L, emb] | y [B*L, emb] |
//                         qkv [B*L, 3emb] | ao [B*L, emb] | feat [B, emb]
size_t mst_fusion_workspace_bytes(const mst_fusion_weights* w, int B, int D) {
    if (!w || B <= 0 || D <= 0) return 0;
    const size_t L = (size_t)D + 1, e = (size_t)w->emb;
    size_t o = 0;
    o += align_up((size_t)B * D * e * 4, 256);
    o += 4 * align_up((size_t)B * L * e * 4, 256);
    o += align_up((size_t)B * L * 3 * e * 4, 256);
    o += align_up((size_t)B * e * 4, 256);
    return o;
}

int mst_slice_fusion(const mst_fusion_weights* w, const float* emb, int B, int D, const uint8_t* key_padding_mask,
                     float* features, float* logits, float* slice_probs, void* ws, size_t ws_bytes,
                     mst_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MST_CHECK_ARG(w && emb && features && ws, "slice_fusion: null pointer");
    MST_CHECK_ARG(B > 0 && D > 0, "slice_fusion: B=%d D=%d", B, D);
    const int e = w->emb, L = D + 1;
    const size_t need = mst_fusion_workspace_bytes(w, B, D);
    if (ws_bytes < need) {
        mst_set_error("slice_fusion: workspace %zu < %zu bytes", ws_bytes, need);
        return MST_EWORKSPACE;
    }
    char* p = (char*)ws;
    float* eb = (float*)p;  p += align_up((size_t)B * D * e * 4, 256);
    float* xs = (float*)p;  p += align_up((size_t)B * L * e * 4, 256);
    float* y = (float*)p;   p += align_up((size_t)B * L * e * 4, 256);
    float* ao = (float*)p;  p += align_up((size_t)B * L * e * 4, 256);
    float* y2 = (float*)p;  p += align_up((size_t)B * L * e * 4, 256);
    float* qkv = (float*)p; p += align_up((size_t)B * L * 3 * e * 4, 256);
    float* feat = (float*)p;

    const float* src = emb;
    if (w->bottleneck_w) {  // dino.py:134-135
        MST_CHECK_ARG(w->emb_in % 16 == 0, "slice_fusion: emb_in=%d must be a multiple of 16", w->emb_in);
        RUN(launch_gemm32(emb, w->emb_in, w->bottleneck_w, w->emb_in, w->bottleneck_b, eb, e, (int64_t)B * D, e,
                          w->emb_in, MST_EPI_BIAS, nullptr, 1.f, 0, s));
        src = eb;
    } else {
        MST_CHECK_ARG(w->emb_in == e, "slice_fusion: emb_in=%d != emb=%d without a bottleneck", w->emb_in, e);
    }

    int F = e;  // feature width
    if (w->fusion_type == MST_FUSION_TRANSFORMER) {
        MST_CHECK_ARG(w->cls_token && w->in_proj_w && w->out_proj_w && w->lin1_w && w->lin2_w && w->ln1_w && w->ln2_w && w->norm_w,
                      "slice_fusion: missing transformer weights");
        MST_CHECK_ARG(w->num_heads > 0 && e % w->num_heads == 0, "slice_fusion: emb=%d not divisible by heads=%d", e, w->num_heads);
        MST_CHECK_ARG(e % 16 == 0, "slice_fusion: emb=%d must be a multiple of 16", e);
        MST_CHECK_ARG(!w->slice_pos_emb || D <= 256, "slice_fusion: slice_pos_emb holds 256 positions, D=%d", D);
        const int hd = e / w->num_heads;
        const int64_t ML = (int64_t)B * L;
        RUN(launch_slice_tokens(src, w->cls_token, w->slice_pos_emb, B, D, e, xs, s));  // dino.py:140-145
        // x = x + out_proj(attn(in_proj(norm1 x)))                 transformer_blocks.py:567,576-582
        RUN(launch_layernorm(xs, e, w->ln1_w, w->ln1_b, y, MST_F32, e, ML, e, 1e-5f, s));
        RUN(launch_gemm32(y, e, w->in_proj_w, e, w->in_proj_b, qkv, 3 * e, ML, 3 * e, e, MST_EPI_BIAS, nullptr, 1.f, 0, s));
        if (w->liere_rot) {
            // the reference's LieRE path views [B, 33, heads, hd] and then [B*heads, L, hd] of a permuted tensor:
            // both views fail (RuntimeError) unless D == 32 and B == 1 (rotary_embedding_torch.py:349;
            // transformer_blocks.py:263)
            MST_CHECK_ARG(!w->rope_freqs, "slice_fusion: rope_freqs and liere_rot are exclusive");
            MST_CHECK_ARG(L == 33, "slice_fusion: LieRE is built for 33 tokens (axes_length, transformer_blocks.py:355): "
                          "shape '[%d, 33, %d, %d]' is invalid for %d slices", B, w->num_heads, hd, D);
            MST_CHECK_ARG(B == 1, "slice_fusion: LieRE: view size is not compatible with the rotated tensor's "
                          "size and stride for batch %d > 1 (transformer_blocks.py:263)", B);
        }
        RUN(launch_slice_attn(qkv, B, L, w->num_heads, hd, key_padding_mask, w->rope_freqs, w->liere_rot, ao, slice_probs, s));
        RUN(launch_gemm32(ao, e, w->out_proj_w, e, w->out_proj_b, xs, e, ML, e, e, MST_EPI_RESIDUAL, nullptr, 1.f, 0, s));
        // x = x + linear2(relu(linear1(norm2 x)))                  transformer_blocks.py:568,585-587
        RUN(launch_layernorm(xs, e, w->ln2_w, w->ln2_b, y, MST_F32, e, ML, e, 1e-5f, s));
        RUN(launch_gemm32(y, e, w->lin1_w, e, w->lin1_b, y2, e, ML, e, e, MST_EPI_BIAS_RELU, nullptr, 1.f, 0, s));
        RUN(launch_gemm32(y2, e, w->lin2_w, e, w->lin2_b, xs, e, ML, e, e, MST_EPI_RESIDUAL, nullptr, 1.f, 0, s));
        // final LayerNorm, row 0 of every volume                   dino.py:95,153
        RUN(launch_layernorm(xs, (int64_t)L * e, w->norm_w, w->norm_b, feat, MST_F32, e, B, e, 1e-5f, s));
    } else if (w->fusion_type == MST_FUSION_LINEAR) {  // dino.py:154-155: 'b d e -> b (d e)'
        F = D * e;
        feat = (float*)src;
    } else if (w->fusion_type == MST_FUSION_AVERAGE) {  // dino.py:156-157
        RUN(launch_mean_slices(src, B, D, e, feat, s));
    } else {
        mst_set_error("slice_fusion: bad fusion type %d", w->fusion_type);
        return MST_EINVAL;
    }
    RUN(launch_rows_copy(feat, F, features, F, B, F, s));
    if (logits) {  // dino.py:166
        MST_CHECK_ARG(w->head_w && w->out_ch > 0, "slice_fusion: logits requested without a head");
        MST_CHECK_ARG(F % 16 == 0, "slice_fusion: feature width %d must be a multiple of 16", F);
        MST_CHECK_ARG(w->head_in <= 0 || F == w->head_in, "slice_fusion: mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)",
                      B, F, w->head_in, w->out_ch);
        RUN(launch_gemm32(feat, F, w->head_w, F, w->head_b, logits, w->out_ch, B, w->out_ch, F, MST_EPI_BIAS, nullptr, 1.f, 0, s));
    }
    return MST_OK;
}

int mst_attention_readout(const float* cls_probs_last, const float* slice_probs, int B, int D, int heads, int N,
                          int num_registers, int sheads, float* plane, float* slice_attn, float* maps,
                          mst_stream_t stream) {
    MST_CHECK_ARG(B > 0 && D > 0 && heads > 0 && N > 1 + num_registers, "readout: bad sizes");
    return launch_readout(cls_probs_last, slice_probs, B, D, heads, N, num_registers, sheads, plane, slice_attn, maps,
                          (hipStream_t)stream);
}

int mst_saliency_accumulate(const float* maps, const float* slice_attn, int D, int heads, int gh, int gw, int Np,
                            int flip_mask, int accumulate, float* lowres, float* slice_acc, mst_stream_t stream) {
    MST_CHECK_ARG(maps && lowres && (slice_attn || !slice_acc), "saliency_accumulate: null pointer");
    MST_CHECK_ARG(D > 0 && heads > 0 && gh > 0 && gw > 0 && gh * gw <= Np, "saliency_accumulate: grid %d x %d does not fit %d map columns", gh, gw, Np);
    MST_CHECK_ARG((flip_mask & ~7) == 0, "saliency_accumulate: flip_mask %d", flip_mask);
    return launch_saliency_accumulate(maps, slice_attn, D, heads, gh, gw, Np, flip_mask, accumulate, lowres, slice_acc,
                                      (hipStream_t)stream);
}

int mst_saliency_upsample(const float* lowres, int D, int gh, int gw, float scale, int Dout, int H, int W, float* out,
                          mst_stream_t stream) {
    MST_CHECK_ARG(lowres && out, "saliency_upsample: null pointer");
    MST_CHECK_ARG(D > 0 && gh > 0 && gw > 0 && Dout > 0 && H > 0 && W > 0, "saliency_upsample: bad sizes");
    return launch_saliency_upsample(lowres, D, gh, gw, scale, Dout, H, W, out, (hipStream_t)stream);
}

int mst_liere_rotation(const float* vars, int n_blocks, int block, int axes_length, float* R, mst_stream_t stream) {
    MST_CHECK_ARG(vars && R, "liere_rotation: null pointer");
    return launch_liere_rotation(vars, n_blocks, block, axes_length, R, (hipStream_t)stream);
}

int mst_attention_rollout(const float* const* maps, int n_layers, int64_t batch, int N, float* out, float* tmp,
                          mst_stream_t stream) {
    MST_CHECK_ARG(maps && out && n_layers >= 1 && batch > 0 && N > 0, "rollout: bad arguments");
    MST_CHECK_ARG(n_layers <= 2 || tmp, "rollout: tmp is required for more than two maps");
    hipStream_t s = (hipStream_t)stream;
    for (int l = 0; l < n_layers; ++l)
        MST_CHECK_ARG(maps[l] && maps[l] != out && maps[l] != tmp, "rollout: map %d is null or aliases out/tmp", l);
    if (n_layers == 1) {
        if (hipMemcpyAsync(out, maps[0], sizeof(float) * (size_t)batch * N * N, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            mst_set_error("rollout: copy failed");
            return MST_ELAUNCH;
        }
        return MST_OK;
    }
    // n_layers - 1 products, alternating tmp/out so that the last one lands in `out`
    const float* cur = maps[n_layers - 1];
    const int links = n_layers - 1;
    for (int i = 0; i < links; ++i) {
        float* dst = ((links - 1 - i) & 1) ? tmp : out;
        int rc = launch_bmm32_nn(maps[n_layers - 2 - i], cur, dst, batch, N, N, N, s);
        if (rc != MST_OK) return rc;
        cur = dst;
    }
    return MST_OK;
}

mst_profiler* mst_profiler_create(void) { return new (std::nothrow) mst_profiler(); }

void mst_profiler_destroy(mst_profiler* p) {
    if (!p) return;
    for (auto* v : {&p->used, &p->idle})
        for (const auto& r : *v) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    delete p;
}

int mst_profiler_collect(mst_profiler* p, double* ms_total, int64_t* launches) {
    MST_CHECK_ARG(p && ms_total && launches, "profiler_collect: null pointer");
    std::lock_guard<std::mutex> lk(p->mu);
    for (int k = 0; k < MST_K_COUNT; ++k) { ms_total[k] = 0.0; launches[k] = 0; }
    for (const auto& r : p->used) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
            mst_set_error("profiler_collect: event query failed");
            return MST_ELAUNCH;
        }
        ms_total[r.kind] += ms;
        launches[r.kind] += 1;
        p->idle.push_back(r);
    }
    p->used.clear();
    return MST_OK;
}

const char* mst_kernel_kind_name(int kind) { return (kind >= 0 && kind < MST_K_COUNT) ? kKindNames[kind] : "?"; }

}  // extern "C"
