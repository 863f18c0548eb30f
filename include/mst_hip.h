/*
 * mst_hip.h -- C ABI of libmst_hip.so: the MI355X (gfx950) kernels behind the Medical Slice
 * Transformer hot path  DinoV2ClassifierSlice.forward  (+ its save_attn read-outs).
 *
 * The reference (gabrielfnayres/new-vit) has no FFI: the path sits behind a Python class
 * (mst/models/dino.py:32-275).  This header is the boundary the build introduces underneath that
 * class: plain device pointers, sizes and a hipStream_t; no torch types; no
 * allocation inside (the caller passes a workspace); every call is asynchronous on `stream` and
 * re-entrant for distinct streams; status codes instead of exceptions (0 = ok, message via
 * mst_last_error()).  Each entry point cites the reference lines whose arithmetic it replaces
 * (paths relative to the reference root).  The reference-side binding is in INTEGRATION.md.
 *
 * Layouts: all matrices row-major.  "compute dtype" T is MST_F16 or MST_BF16 (MFMA operands,
 * fp32 accumulate) or MST_F32 (exact fp32 MFMA).  The residual stream, LayerNorm statistics,
 * softmax and every bias/affine parameter are fp32 in all modes.
 */
#ifndef MST_HIP_H
#define MST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mst_stream_t; /* hipStream_t */

enum mst_dtype { MST_F32 = 0, MST_F16 = 1, MST_BF16 = 2, MST_F8E4M3 = 3 /* OCP e4m3 bytes: fp8 entry points only */ };

enum mst_status {
    MST_OK = 0,
    MST_EINVAL = 1,     /* bad argument / unsupported shape */
    MST_EWORKSPACE = 2, /* workspace too small */
    MST_ELAUNCH = 3     /* HIP launch error */
};

enum mst_epilogue {
    MST_EPI_BIAS = 0,      /* C = A W^T + b                               (any linear)            */
    MST_EPI_BIAS_GELU = 1, /* C = gelu_erf(A W^T + b)                     mlp.py:34-36            */
    MST_EPI_BIAS_RELU = 2, /* C = relu(A W^T + b)                         transformer_blocks.py:586 */
    MST_EPI_RESIDUAL = 3,  /* C(f32) += gamma * (A W^T + b)               block.py:90-94,112-113  */
    MST_EPI_RESIDUAL_RELU = 4 /* C(f32) = relu(C + gamma * (A W^T + b))    resnet BasicBlock / Bottleneck exit; fp32 operands only */
};

enum mst_fusion_type { MST_FUSION_TRANSFORMER = 0, MST_FUSION_LINEAR = 1, MST_FUSION_AVERAGE = 2 };

/* Library ---------------------------------------------------------------------------------- */
int mst_version(void);            /* 100 = round 1; 200 = round 2: mst_vit_layer.proj_pack/proj_bf, mst_fusion_weights.head_in,
                                   * mst_vit_weights.profiler / prune_last_block appended; 300 = round 3: mst_vit_layer.block_seq
                                   * inserted, mst_block_fused_s; the ctypes mirror checks it */
const char* mst_last_error(void); /* thread-local message of the last failing call */

/* Per-op entry points (unit parity; also what mst_vit_encode / mst_slice_fusion launch) ----- */

/* nn.LayerNorm over the last dim: block.py:63,75; vision_transformer.py:165,263 (eps 1e-6);
 * transformer_blocks.py:499-500, dino.py:95 (eps 1e-5).  x fp32 [rows, cols] with row stride
 * x_stride (elements); out dtype f32/f16/bf16 with row stride out_stride.  cols even, <= 2048.
 * gamma == beta == NULL: normalise only (the affine is folded into the consumer's weights). */
int mst_layernorm(const float* x, int64_t x_stride, const float* gamma, const float* beta,
                  void* out, int out_dtype, int64_t out_stride, int64_t rows, int cols, float eps,
                  mst_stream_t stream);

/* nn.Linear / F.linear with fused epilogue: attention.py:58,67; mlp.py:35,38;
 * transformer_blocks.py:166,283,586; dino.py:135,166.
 * C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]).  A and W share one dtype.
 *   16-bit A/W: MFMA 16x16x32 path; needs K % 64 == 0, N % 128 == 0, 16-byte aligned rows.
 *   f32   A/W: exact fp32 MFMA path; needs K % 16 == 0; any M, N.
 * c_dtype: f32 or (16-bit A only) the A dtype; MST_EPI_RESIDUAL needs c_dtype f32 (C is read and
 * written, gamma = LayerScale vector or NULL: layer_scale.py:26-27).
 * Columns n < scale_cols are multiplied by col_scale after the bias (q * head_dim^-0.5 of
 * attention.py:60 folded into the QKV projection); pass scale_cols = 0 for none. */
int mst_gemm(const void* A, int ab_dtype, int64_t lda, const void* W, int64_t ldw, const float* bias,
             void* C, int c_dtype, int64_t ldc, int64_t M, int N, int K, int epilogue,
             const float* gamma, float col_scale, int scale_cols, mst_stream_t stream);

/* FP8 linear layers (BASELINE.json configs[4]): OCP e4m3 operands, one scale per tensor from its
 * absolute maximum, fp32 accumulate.  The reference has no fp8 code; this is F.linear (attention.py:58,67; mlp.py:35,38)
 * with both operands rounded to e4m3:  y = (sa*sw) * (q(x/sa) . q(W/sw)^T) + b,  sa = max|x|/448,  sw = max|W|/448.
 *
 * mst_quantize_fp8: x (bf16/fp16, n elements, n % 8 == 0) -> out8[n] e4m3 bytes = rne(x * 448/amax), where *amax (device
 *   fp32) = max(*amax on entry, max|x|): pass 0 for a fresh per-tensor scale, or a calibrated floor.  Stream-ordered, no
 *   host synchronisation.
 * mst_gemm_fp8: C[M,N] = epi((*a_amax/448 * w_scale) * A8[M,K] . W8[N,K]^T + bias); A8, W8 e4m3 bytes with K-contiguous
 *   rows; K % 128 == 0, N % 128 == 0, lda/ldw % 16 == 0; c_dtype f32 / bf16 / fp16; epilogues, gamma, col_scale and
 *   scale_cols as mst_gemm.  c_dtype MST_F8E4M3 (non-residual epilogues): C is written as e4m3 bytes rne(clamp(v * 448 /
 *   *c_amax)) under the caller's scale c_amax (device fp32; calibrated: nothing is scanned); c_amax is ignored otherwise.
 * mst_layernorm_fp8: mst_layernorm writing e4m3 bytes under a calibrated scale: out8 = rne(clamp(LN(x) * 448 / *amax)). */
int mst_quantize_fp8(const void* x, int dtype, int64_t n, float* amax, void* out8, mst_stream_t stream);
int mst_gemm_fp8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* bias, const float* a_amax,
                 float w_scale, void* C, int c_dtype, int64_t ldc, int64_t M, int N, int K, int epilogue,
                 const float* gamma, float col_scale, int scale_cols, const float* c_amax, mst_stream_t stream);
int mst_layernorm_fp8(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* out8,
                      int64_t out_stride, int64_t rows, int cols, float eps, const float* amax, mst_stream_t stream);

/* softmax(q k^T) v per (sequence, head), q pre-scaled: attention.py:56-66 (== xformers
 * memory_efficient_attention, attention.py:84).  qkv [n_seq*N, 3*heads*head_dim] packed as the
 * reference's fused projection lays it out (q | k | v, head-major inside each); out
 * [n_seq*N, heads*head_dim].  16-bit dtypes: head_dim == 64 (flash-style, never materialises
 * [N,N]); f32: head_dim == 64. */
int mst_attention(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim, void* out,
                  mst_stream_t stream);

/* Row 0 (CLS query) of the softmax, all heads: what dino.py:226-243 stores and
 * dino.py:190-192 consumes.  probs fp32 [n_seq, heads, N]. */
int mst_attention_cls_probs(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim,
                            float* probs, mst_stream_t stream);

/* The full softmax matrix (dino.py:232-241 `attention_maps` entry): probs fp32
 * [n_seq, heads, N, N].  API parity for get_attention_cls (dino.py:204-212); O(N^2) memory. */
int mst_attention_probs_full(const void* qkv, int dtype, int n_seq, int N, int heads, int head_dim,
                             float* probs, mst_stream_t stream);

/* Bicubic resampling of the patch position grid: vision_transformer.py:179-211 (F.interpolate bicubic).
 * offset = interpolate_offset (l.194-202): != 0 -> scale_factor = (g + offset) / M, the vendored default 0.1; 0 -> the
 * output size is given.  antialias = interpolate_antialias (l.206): torch's anti-aliased bicubic (Keys A = -0.5, support
 * widened by the down-sampling factor); the hub's register models use offset 0, antialias 1.
 * pos_patch fp32 [M*M, E] -> out fp32 [gh*gw, E]. */
int mst_pos_embed_interp(const float* pos_patch, int M, int E, int gh, int gw, double offset, int antialias,
                         float* out, mst_stream_t stream);

/* Gray->RGB + Conv2d(3,E,14,14) + flatten + CLS/register rows + position add:
 * dino.py:125-127; patch_embed.py:68-81; vision_transformer.py:213-232.
 * vol [n, H, W] (in_dtype f32/f16/bf16), wp = channel-summed kernel [E][224] in compute dtype
 * (k = ky*16+kx, kx 14,15 zero), prefix fp32 [1+R, E] (row 0 = cls_token + pos[0], rows 1..R =
 * register tokens), pos_patch fp32 [Np, E].  Writes tokens x fp32 [n, 1+R+Np, E]. */
int mst_patch_embed(const void* vol, int in_dtype, int n, int H, int W, const void* wp, int dtype,
                    const float* bias, const float* prefix, int n_prefix, const float* pos_patch,
                    int E, float* x, mst_stream_t stream);

/* Convolutional backbone of the ResNet models (reference mst/models/resnet.py:44-50,127-243 on torchvision's
 * resnet34, which is not part of the reference tree: the published architecture is restated).  Activations are NHWC fp32, so a
 * convolution (+ folded BatchNorm + ReLU / residual) is mst_im2col_nhwc followed by mst_gemm with the matching epilogue.
 * mst_im2col_nhwc: x [n,H,W,C] -> col [n*Ho*Wo, Kpad], col[(n,oy,ox)][(ky,kx,c)] = x[n][oy*stride-pad+ky][ox*stride-pad+kx][c],
 *   zero outside the image and for k >= kh*kw*C.  mst_maxpool_nhwc: 3x3, stride 2, padding 1 (the resnet stem).
 * mst_avgpool_nhwc: x [n, HW, C] -> y [n, C], the adaptive average pool to 1x1. */
int mst_im2col_nhwc(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* col,
                    mst_stream_t stream);
/* mst_conv_gemm: the same convolution (+ folded BatchNorm bias, ReLU / residual epilogues of mst_gemm) as an IMPLICIT GEMM: the A operand
 * is gathered from x [n,H,W,Cin] on the fly, no [rows, kh*kw*Cin] matrix is materialised (9x the activation for a 3x3 layer).  Cin % 16
 * == 0 (every layer behind the stem); Wg [Cout, Kpad] and the (ky, kx, c) order as above; out [n*Ho*Wo, Cout] fp32 (read and written
 * by MST_EPI_RESIDUAL / _RESIDUAL_RELU); epilogue MST_EPI_BIAS / MST_EPI_BIAS_RELU / MST_EPI_RESIDUAL / MST_EPI_RESIDUAL_RELU. */
int mst_conv_gemm(const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const float* Wg, const float* bias,
                  float* out, int Cout, int Kpad, int epilogue, const float* gamma, mst_stream_t stream);
/* mst_conv_gemm16: the implicit-GEMM convolution on 16-bit MFMA operands (fp32 accumulation): x [n,H,W,Cin] and Wg [Cout, kh*kw*Cin] bf16 / f16
 * (Cin % 64 == 0, Cout % 4 == 0), out [n*Ho*Wo, Cout] of out_dtype: MST_EPI_BIAS / MST_EPI_BIAS_RELU -> the operand type or f32;
 * MST_EPI_RESIDUAL_RELU -> out = relu(out + conv + bias) in place on a 16-bit out (the residual unit's exit).  The 16-bit inference path of
 * the ResNet backbone (ResNet(..., compute_dtype='bf16')).  mst_cvt32: out[i] = float(x[i]), n % 4 == 0. */
int mst_conv_gemm16(const void* x, int dtype, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, const void* Wg, const float* bias,
                    void* out, int out_dtype, int Cout, int epilogue, mst_stream_t stream);
int mst_cvt32(const void* x, int dtype, int64_t n, float* out, mst_stream_t stream);
/* The stem of the 16-bit backbone: mst_im2col_nhwc with the rows rounded to out_dtype (bf16 / f16) on the way out, and mst_maxpool_nhwc on
 * 16-bit activations (3 x 3, stride 2, padding 1). */
int mst_im2col_nhwc16(const float* x, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, void* col, int out_dtype,
                      mst_stream_t stream);
int mst_maxpool_nhwc16(const void* x, int dtype, int n, int H, int W, int C, void* y, mst_stream_t stream);
/* mst_conv_dgrad: d input of a convolution AS a convolution (what torch.autograd's conv backward gives the reference): the same implicit GEMMs
 * with stride 1, padding k - 1 - pad and the gradient rows dilated by the forward stride (1 or 2), instead of dZ . W into a [rows, kh*kw*Cin]
 * matrix and a scatter with atomics (mst_col2im_nhwc).  dz [n,Ho,Wo,Cout] and Wt [Cin, kh*kw*Cout] of `dtype` (f32: Cout % 16 == 0; bf16 / f16:
 * Cout % 64 == 0), Wt[c][(ky',kx',co)] = W[co][c][k-1-ky'][k-1-kx']; dx fp32 [n*H*W, Cin], overwritten. */
/* mst_conv_wgrad: d weight of a convolution as an implicit GEMM: part[z][co][(ky,kx,c)] = sum over the output pixels [z rps, (z + 1) rps) of
 * dz[r][co] * x[pixel(r) + (ky,kx)][c] -- nsplit partial products fp32 [nsplit, Cout, kh*kw*Cin] (the caller sums them: mst_colsum), B operand
 * gathered from x [n,H,W,Cin] (no im2col matrix).  dz [n*Ho*Wo, Cout] fp32; Cin % 64 == 0, Cout % 4 == 0, nsplit * rows_per_split >= rows. */
int mst_conv_wgrad(const float* dz, const float* x, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                   int nsplit, int64_t rows_per_split, mst_stream_t stream);
/* mst_conv_wgrad16: the same partial products from 16-bit operands (dz and x bf16 / f16; Cin % 64 == 0, Cout % 64 == 0, rows_per_split % 64 == 0):
 * pixel-major tiles in LDS, fragments through the hardware-transposing LDS read. */
int mst_conv_wgrad16(const void* dz, const void* x, int dtype, int n, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Cout, float* part,
                     int nsplit, int64_t rows_per_split, mst_stream_t stream);
int mst_conv_dgrad(const void* dz, int dtype, int n, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, const void* Wt, int H, int W,
                   int Cin, float* dx, mst_stream_t stream);
int mst_maxpool_nhwc(const float* x, int n, int H, int W, int C, float* y, mst_stream_t stream);
int mst_avgpool_nhwc(const float* x, int n, int HW, int C, float* y, mst_stream_t stream);
/* Training step of the backbone (BASELINE configs[3]; what torch.autograd + nn.BatchNorm2d(train) do for the reference):
 * mst_batchnorm_train: z [rows, C] (raw convolution output) -> y = gamma (z - mean) rstd + beta (+ residual) (ReLU if relu) with the
 *   BATCH statistics (biased variance); mean / rstd [C] are kept for the backward; running_mean / running_var (nullable) are
 *   updated as nn.BatchNorm2d does (momentum, unbiased variance); scratch: C floats.
 * mst_batchnorm_bwd: dgamma, dbeta (+=, zero them first) and dz = gamma rstd (dy - dbeta/rows - xhat dgamma/rows).
 * mst_col2im_nhwc: adjoint of mst_im2col_nhwc, dx += (fp32 atomics; zero dx first): dX of a convolution = col2im(dZ . W).
 * mst_maxpool_bwd_nhwc (gradient to the first maximum of each window), mst_avgpool_bwd_nhwc. */
int mst_batchnorm_train(const float* z, int64_t rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                        const float* residual, int relu, float* y, float* mean, float* rstd, float* running_mean,
                        float* running_var, float* scratch, mst_stream_t stream);
int mst_batchnorm_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* dy, int64_t rows, int C,
                      float* dgamma, float* dbeta, float* dz, mst_stream_t stream);
int mst_col2im_nhwc(const float* dcol, int n, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* dx,
                    mst_stream_t stream);
int mst_maxpool_bwd_nhwc(const float* x, const float* dy, int n, int H, int W, int C, float* dx, mst_stream_t stream);
int mst_avgpool_bwd_nhwc(const float* dy, int n, int HW, int C, float* dx, mst_stream_t stream);
/* Grad-CAM++ map of the backbone's last ReLU output (reference mst/models/resnet.py:93-118 compute_attention_maps /
 * compute_grad_cam_weights, the map get_attention_maps returns): act [n, HW, C] fp32 NHWC; out [n, O] = the model output whose
 * per-image maximum is the loss (resnet.py:66-68); W [O, C] = the fc weight, or NULL when the output IS the pooled features
 * (O == C).  d loss / d pooled features = W[argmax] (or the one-hot), and behind the global average pool every position of a
 * channel has that gradient / HW -> cam [n, HW] = relu(sum_c w_c act_c), shifted by the global minimum and divided by the global
 * maximum as the reference does.  state: 2 floats of scratch. */
int mst_gradcampp(const float* act, const float* out, int O, const float* W, int n, int HW, int C, float* cam, float* state,
                  mst_stream_t stream);

/* Input pipeline in front of the model (reference mst/data/datasets/augmentations/augmentations_3d.py on
 * torchio 0.19.9, which is not part of the reference tree: restated from its published algorithm, numpy.pad semantics included).
 * mst_crop_or_pad: CropOrPad (l.144-195) with the deterministic centre (random_center=False: ini = ceil(n/2), fin = n - ini per
 *   axis).  src fp32 [s0,s1,s2] -> dst fp32 [t0,t1,t2]; padding first (pad_minimum = 1: numpy.pad mode 'minimum', axis after
 *   axis, what the datasets use; 0: the constant pad_value), then cropping.  ws: (s0+p0)(s1+p1)(s2+p2) floats when an axis is
 *   padded while another is cropped, else unused.
 * mst_znorm: ZNormalization (l.40-86) of ONE channel (per_channel=True, per_slice=False) with the datasets' masking method
 *   (x > x.min()) & (x < x.max()): cut-offs = torch.quantile(masked values, {q_lo, q_hi}) (linear), clamp, then
 *   y = (clamp(x) - mean) / std with mean / unbiased std of the masked clamped values.  Everything stays on the device (exact
 *   order statistics by radix select); `state` (mst_znorm_state_bytes()) receives, among others, a zero_std flag the host mirror
 *   turns into the reference's RuntimeError. */
int mst_crop_or_pad(const float* src, int s0, int s1, int s2, float* dst, int t0, int t1, int t2, int pad_minimum,
                    float pad_value, void* ws, size_t ws_bytes, mst_stream_t stream);
size_t mst_znorm_state_bytes(void);
int mst_znorm(const float* x, int64_t n, float q_lo, float q_hi, float* y, void* state, mst_stream_t stream);

/* slices2rgb (mst/models/dino.py:10-27; dead code in the reference: its call at dino.py:129 is commented out): three
 * consecutive gray slices become the channels of one image.  vol [B,1,D,H,W] (dtype) -> out [B*Dp/3, 3, H, W] with
 * Dp = D rounded up to a multiple of 3, the extra slices being the volume's own first Dp-D slices (needs Dp-D <= D). */
int mst_slices2rgb(const void* vol, int dtype, int B, int D, int H, int W, void* out, mst_stream_t stream);

/* Fused MLP half of a ViT block (block.py:93-94,113; mlp.py:34-40), E = 384, 16-bit operands:
 *   x[M,E] (fp32, in place) += ls2 * (fc2(gelu(fc1(normalise(x)))) + b2);  xn_out (nullable, dtype) = normalise(x_new)
 * with LayerScale ls2 folded by the caller into W2's rows and into b2f = ls2 * b2 (ls2 = 1 when absent).
 * normalise = LayerNorm without affine (eps).  The hidden activations stay on chip.
 * wpack: 48 chunks x 49152 B, chunk c = LDS image of W1f rows [32c,32c+32) and W2 columns [32c,32c+32):
 *   W1 part  [ks 0..11][h 0..31][slot 0..3][8]  = W1f[32c+h][32ks + 8(slot^f(h)) ..+7],   f(r) = (-(r>>2))&3
 *   W2 part  [R 0..383][slot 0..3][8]           = ls2[n(R)] * W2[n(R)][32c + phys(slot^f(R), 0..7)]
 *     n(R)      = 32((R>>4)>>1) + 8((R&15)>>2) + 4((R>>4)&1) + (R&3)
 *     phys(c,j) = j<4 ? 4c+j : 16+4c+(j-4)
 *   W1f = fc1_w * ln2_w (columns);  b1f (fp32, 1536+32 padded) = fc1_b + fc1_w . ln2_b;  b2 = fc2 bias.
 * (new-vit_amd/mst/models/dino.py::_pack_mlp builds it.) */
int mst_mlp_fused(float* x, void* xn_out, int dtype, const void* wpack, const float* b1f, const float* b2f,
                  int64_t M, int E, float eps, mst_stream_t stream);

/* Everything of a ViT block after the attention kernel in ONE launch (E = 384, 16-bit operands): the out-projection
 * (attention.py:67-68) + residual (block.py:90-91,112), then the fused MLP half exactly as mst_mlp_fused:
 *   x[M,E] (fp32, in place) += ls1 * (proj(attn_out) + b_proj);  x += ls2 * (fc2(gelu(fc1(normalise(x)))) + b2);
 *   xn_out (nullable, dtype; MAY alias attn_out) = normalise(x_new)
 * x is read once and written once.  attn_out [M,E] dtype.  proj_pack: 12 chunks x 24576 B, chunk j = LDS image
 *   [R 0..383][slot 0..3][8] = ls1[n(R)] * proj_w[n(R)][32j + 8(slot^f(R)) ..+7]   (n, f as in mst_mlp_fused),
 * proj_bf fp32 [E] = ls1 * proj_b (ls1 = 1 when absent); wpack / b1f / b2f as mst_mlp_fused.  scratch: at least
 * mst_block_fused_scratch_bytes() (96 KiB per compute unit: the LayerNorm2 hand-off between the kernel's wave roles). */
size_t mst_block_fused_scratch_bytes(void);
int mst_block_fused(float* x, const void* attn_out, void* xn_out, int dtype, const void* proj_pack, const float* proj_bf,
                    const void* wpack, const float* b1f, const float* b2f, void* scratch, size_t scratch_bytes, int64_t M,
                    int E, float eps, mst_stream_t stream);

/* The same block tail (attention.py:67-68; block.py:90-94,112-113; mlp.py:34-40) as mst_block_fused, in the round-3 SINGLE-ROLE
 * form: one wave per SIMD owns 32 token rows end to end (y^T accumulators, the LayerNorm2 rows and the hidden activations never
 * leave its registers), so no scratch and no activation hand-off exist.  Same arithmetic contract and aliasing rules.
 * block_seq: the block's weights as ONE stream of 108 elements x 24576 B in the order the kernel consumes them --
 *   12 out-projection chunks, W1(0), then W1(c+1), W2(c) for c = 0..46, then W2(47) -- each element 24 MFMA A-fragments
 *   [fragment f = 2t+p][lane l][8], m = l & 31, h = l >> 5, k8(e) = 16p + 8(e>>2) + 4h + (e&3):
 *     out-projection chunk j : ls1[32t+m] * proj_w[32t+m][32j + 16p + 8h + e]
 *     W1 chunk c (f = k-step): ln2_w[.] * fc1_w[32c+m][32t + k8(e)]
 *     W2 chunk c             : ls2[32t+m] * fc2_w[32t+m][32c + k8(e)]
 * b1f fp32 [1536] = fc1_b + fc1_w . ln2_b; proj_bf fp32 [E] = ls1 * proj_b; b2f fp32 [E] = ls2 * fc2_b.
 * layout: 0 = every operand row-major, or a combination of mst_layout_flags.  The kernel's lanes own ROWS, so a row-major operand
 * costs 32 scattered 32-byte runs per memory instruction; between two blocks of one encoder the operands stay in the order the
 * registers hold them (buffers then cover whole groups of 32 rows: round M up to a multiple of 32 when allocating):
 *   16-bit "blocked" [M/32 groups][24 pieces][64 slots][8]: element (row r, column c) of group g at piece c/16, slot (r%32) + 32*((c/8)%2),
 *     position c%8 (a piece = one contiguous KiB = 32 rows x 16 consecutive columns; natural column order);
 *   fp32 "image"     [M/32 groups][48 pieces][64 slots][4]: element (r, c) at piece c/8, slot (r%32) + 32*((c/4)%2), position c%4. */
enum mst_layout_flags {
    MST_LAYOUT_X_IN_IMAGE = 1,   /* x is read in the fp32 image layout  */
    MST_LAYOUT_X_OUT_IMAGE = 2,  /* x is written in the fp32 image layout (in place also when the x flags differ: a 32-row group
                                  * occupies the same 48 KiB in both layouts and is read whole before it is written) */
    MST_LAYOUT_ACT_BLOCKED = 4   /* attn_out is read and xn_out written in the 16-bit blocked layout */
};
int mst_block_fused_s(float* x, const void* attn_out, void* xn_out, int dtype, const void* block_seq, const float* b1f,
                      const float* proj_bf, const float* b2f, int64_t M, int E, float eps, int layout, mst_stream_t stream);

/* Training step: what torch.autograd does for the reference (base_model.py:148-181, main_train.py:110-126),
 * as per-op entry points; mst/train.py orchestrates them behind a torch.autograd.Function.  All fp32, exact fp32 MFMA.
 * mst_gemm_ex: C[b] = alpha * A[b] . B[b] + beta * C[b], A [M,K], B [K,N], C [M,N] with element strides
 *   strides[12] = {A_m, A_k, B_k, B_n, C_m, C_n, A_b1, A_b2, B_b1, B_b2, C_b1, C_b2}, batch index b = b1 * nb2 + b2
 *   (dX = dY.W, dW = dY^T.X of nn.Linear; S = q.k^T, O = P.v and the four backward products of attention.py:56-66 on the
 *   packed q|k|v rows).  nb1 * nb2 <= 65535.
 * mst_softmax_rows: S [rows, L] -> softmax over L in place; mask (nullable) uint8 [rows / rows_per_batch, L], 1 = key
 *   ignored (src_key_padding_mask, transformer_blocks.py:244-252).  mst_softmax_rows_bwd: dP <- scale * P o (dP - rowsum(dP o P)).
 * mst_layernorm_bwd: dx[r] = (dres ? dres[r] : 0) + LayerNorm'(x[r]; gamma) . dy[r]; dgamma[c] += sum_r dy xhat, dbeta[c] += sum_r dy
 *   (dx, dres, dgamma, dbeta, gamma nullable; row strides in elements; cols <= 2048; accumulation by fp32 atomics).
 * mst_act_fwd / mst_act_bwd: kind 0 GELU (erf form, mlp.py:22), 1 ReLU; bwd: dy <- dy * act'(h) in place.
 * mst_colsum: out[c] += sum_r a[r][c] * (b ? b[r][c] : 1)   (bias and LayerScale gradients).
 * mst_axpby_cols: y[r][c] = alpha * x[r][c] * (g ? g[c] : 1) + beta * y[r][c].
 * mst_im2col14: vol [n,H,W] -> col fp32 [n*Np, 196], the 14 x 14 patches as rows (patch_embed.py:68-81; d W = dX^T . col).
 * mst_pos_embed_interp_bwd: dpos [M*M, E] += adjoint of mst_pos_embed_interp (antialias = 0) applied to dout [gh*gw, E]. */
/* Mixed-precision training step (the reference trains under precision='16-mixed': scripts/main_train.py:110-123): activations, weights and
 * gradients stay fp32 in memory, the products of nn.Linear run on 16-bit MFMA operands with fp32 accumulation.
 * mst_cvt16: out[r][c] = T(scale * x[r][c]) for r < rows, c < cols (transpose = 0; cols, ldx, ldo multiples of 4), or the TRANSPOSED image
 *   out[c][r], r < rows_pad, zero-filled for r >= rows (transpose = 1): both operands of d weight = dY^T . X contiguous along the token index.
 * mst_gemm16_splitk: Cpart[z] (fp32 [M, N], split_stride elements apart) = A[:, z Kc:(z+1) Kc] . W[:, z Kc:(z+1) Kc]^T, Kc = K / splits a
 *   multiple of 64, N of 128; the caller sums the partial products (mst_colsum over [splits, M * N]). */
int mst_cvt16(const float* x, int64_t ldx, int64_t rows, int cols, float scale, void* out, int out_dtype, int64_t ldo, int transpose,
              int64_t rows_pad, mst_stream_t stream);
int mst_gemm16_splitk(const void* A, int ab_dtype, int64_t lda, const void* W, int64_t ldw, float* Cpart, int64_t ldc, int64_t M, int N, int K,
                      int splits, int64_t split_stride, mst_stream_t stream);
/* mst_rope_rows: RoPE of the across-slice attention (rotary_embedding_torch.py:38-62,159-173) in place on packed q | k | v rows [rows, 3 * heads *
 * head_dim]: the pairs (2p, 2p+1) of q and k rotated by sign * (row % L) * freqs[p]; sign +1 = the training forward, -1 = its adjoint on the
 * gradient rows (the frequencies carry no gradient: learned_freq = False). */
int mst_rope_rows(float* qkv, int64_t rows, int L, int heads, int head_dim, const float* freqs, float sign, mst_stream_t stream);
int mst_gemm_ex(const float* A, const float* B, float* C, int M, int N, int K, const int64_t* strides, int nb1, int nb2,
                float alpha, float beta, mst_stream_t stream);
int mst_softmax_rows(float* S, const uint8_t* mask, int64_t rows, int L, int rows_per_batch, mst_stream_t stream);
int mst_softmax_rows_bwd(const float* P, float* dP, int64_t rows, int L, float scale, mst_stream_t stream);
int mst_layernorm_bwd(const float* x, int64_t x_stride, const float* gamma, const float* dy, int64_t dy_stride, const float* dres,
                      int64_t dres_stride, float* dx, int64_t dx_stride, float* dgamma, float* dbeta, int64_t rows, int cols,
                      float eps, mst_stream_t stream);
int mst_act_fwd(const float* h, float* y, int64_t n, int kind, mst_stream_t stream);
int mst_act_bwd(const float* h, float* dy, int64_t n, int kind, mst_stream_t stream);
int mst_colsum(const float* a, int64_t a_stride, const float* b, int64_t b_stride, int64_t rows, int cols, float* out,
               mst_stream_t stream);
int mst_axpby_cols(const float* x, int64_t x_stride, const float* g, float alpha, float beta, float* y, int64_t y_stride,
                   int64_t rows, int cols, mst_stream_t stream);
int mst_im2col14(const void* vol, int dtype, int n, int H, int W, float* col, mst_stream_t stream);
int mst_pos_embed_interp_bwd(const float* dout, int M, int E, int gh, int gw, double offset, float* dpos, mst_stream_t stream);

/* Optional per-kernel timing of the launches inside mst_vit_encode (bench / profiling only).  A caller-owned object:
 * while mst_vit_weights.profiler points at one, every launch of that call is bracketed by hipEventRecord on the call's own
 * stream; mst_profiler_collect waits for the recorded events, returns the accumulated milliseconds and launch counts per
 * kernel kind since the last collect, and resets.  No process-global state: calls without a profiler record nothing. */
typedef struct mst_profiler mst_profiler;
enum mst_kernel_kind {
    MST_K_PATCH_EMBED = 0, MST_K_LAYERNORM = 1, MST_K_GEMM_QKV = 2, MST_K_ATTENTION = 3,
    MST_K_GEMM_PROJ = 4, MST_K_GEMM_FC1 = 5, MST_K_GEMM_FC2 = 6, MST_K_CLS_PROBS = 7, MST_K_MLP_FUSED = 8,
    MST_K_BLOCK_FUSED = 9,        /* out-projection + MLP in one launch (mst_block_fused) */
    MST_K_COUNT = 10
};
mst_profiler* mst_profiler_create(void);
void mst_profiler_destroy(mst_profiler* p);
int mst_profiler_collect(mst_profiler* p, double* ms_total, int64_t* launches); /* arrays of MST_K_COUNT */
const char* mst_kernel_kind_name(int kind);

/* Whole per-slice encoder ------------------------------------------------------------------ */
typedef struct mst_vit_layer {
    const float* ln1_w; const float* ln1_b;       /* block.py:63  */
    const void* qkv_w;  const float* qkv_b;       /* attention.py:50  [3E,E] compute dtype */
    const void* proj_w; const float* proj_b;      /* attention.py:52  [E,E] */
    const float* ls1;                             /* layer_scale.py:25 or NULL */
    const float* ln2_w; const float* ln2_b;       /* block.py:75  */
    const void* fc1_w;  const float* fc1_b;       /* mlp.py:28  [4E,E] */
    const void* fc2_w;  const float* fc2_b;       /* mlp.py:30  [E,4E] */
    const float* ls2;
    /* Optional fused-LayerNorm form (16-bit modes, E = 384).  When mlp_pack != NULL for every layer the
     * encoder runs  normalise -> QKV(qkv_wf, qkv_bf) -> attention -> proj -> mst_mlp_fused  per block:
     *   qkv_wf = qkv_w * ln1_w (columns), qkv_bf = qkv_b + qkv_w . ln1_b            (norm1 folded)
     *   mlp_pack / fc1_bf / fc2_bf: see mst_mlp_fused                        (norm2 and ls2 folded) */
    const void* qkv_wf; const float* qkv_bf;
    const void* mlp_pack; const float* fc1_bf; const float* fc2_bf;
    /* with proj_pack / proj_bf (see mst_block_fused) also present for every layer the out-projection joins that launch */
    const void* proj_pack; const float* proj_bf;
    /* block_seq (see mst_block_fused_s) with fc1_bf / proj_bf / fc2_bf: the single-role launch is taken instead (ABI 300) */
    const void* block_seq;
    /* Optional FP8 form (mst_vit_weights.fp8_linear): the four block weights as e4m3 bytes, same [out,in] layout, with their
     * per-tensor scales w8_scale[] = max|W|/448 in the order qkv, proj, fc1, fc2 (see mst_gemm_fp8) */
    const void* qkv_w8; const void* proj_w8; const void* fc1_w8; const void* fc2_w8;
    float w8_scale[4];
} mst_vit_layer;

typedef struct mst_vit_weights {
    int embed_dim, depth, num_heads, num_registers;
    int compute_dtype;            /* MST_F16 / MST_BF16 / MST_F32 */
    int grid_h, grid_w;           /* patch grid pos_patch was prepared for */
    const void* patch_w;          /* [E][224] compute dtype (f32 mode: f32) */
    const float* patch_b;         /* [E] */
    const float* prefix;          /* [1+R, E] */
    const float* pos_patch;       /* [grid_h*grid_w, E] */
    const mst_vit_layer* layers;  /* [depth], host memory */
    const float* norm_w; const float* norm_b; /* vision_transformer.py:165 */
    int fp8_linear;               /* 1: the blocks' four linear layers run as mst_gemm_fp8 with dynamic per-tensor activation
                                   * scales (one per GEMM call, i.e. per chunk of slices); needs a 16-bit compute_dtype, which
                                   * stays the type of LayerNorm outputs, q/k/v and the attention kernel */
    const float* fp8_amax;        /* NULL: dynamic scales.  Else device fp32 [depth][4] = calibrated max|x| of the inputs of
                                   * qkv, proj, fc1, fc2 per block (static scales; larger values saturate at +-448 quanta): LayerNorm
                                   * and the GELU epilogue then write e4m3 directly and nothing is scanned */
    float* fp8_amax_out;          /* dynamic mode, nullable: device fp32 [depth][4], out[i] = max(out[i], this call's scales):
                                   * zero it, run representative inputs, and pass it back as fp8_amax (calibration) */
    mst_profiler* profiler;       /* nullable: time this call's launches (see mst_profiler_create) */
    int prune_last_block;         /* 0 (default): every block computes every token, as the reference does.  1: the LAST block
                                   * computes only what is read behind it -- K and V of every token (and the CLS probabilities /
                                   * full maps when asked for), then attention, out-projection and MLP of the CLS rows alone.  The
                                   * reference discards the last block's patch-token outputs (vision_transformer.py:324-329 returns
                                   * x_norm_clstoken only), so the results are the same; ignored in fp8 mode. */
} mst_vit_weights;

/* DinoVisionTransformer.forward on n_slices gray slices (vision_transformer.py:254-270,324-329;
 * block.py:89-114) -> normalised CLS embeddings cls_out fp32 [n_slices, E].
 * cls_probs (nullable) fp32 [n_layers_probs, n_slices, heads, N]: CLS-row softmax of the LAST
 * n_layers_probs blocks (1 is all the reference consumes: dino.py:190; depth reproduces the
 * whole `attention_maps` list, CLS rows only).
 * full_probs (nullable) fp32 [n_layers_probs, n_slices, heads, N, N]: the complete softmax of the
 * same blocks -- API parity for get_attention_cls (dino.py:204-212); O(N^2) memory, off by default.
 * chunk_slices: slices processed per pass (activations of one pass stay resident in the 256 MB
 * Infinity Cache); ws must hold mst_vit_workspace_bytes(...). */
size_t mst_vit_workspace_bytes(const mst_vit_weights* w, int H, int W, int chunk_slices);
int mst_vit_encode(const mst_vit_weights* w, const void* vol, int in_dtype, int n_slices, int H, int W,
                   float* cls_out, float* cls_probs, float* full_probs, int n_layers_probs,
                   int chunk_slices, void* ws, size_t ws_bytes, mst_stream_t stream);

/* Across-slice transformer + head ---------------------------------------------------------- */
typedef struct mst_fusion_weights {
    int emb_in;      /* E of the encoder */
    int emb;         /* E after the optional bottleneck (dino.py:75-78) */
    int out_ch;      /* 0 = no head (enable_linear=False, dino.py:103) */
    int fusion_type; /* mst_fusion_type (dino.py:144-157) */
    int num_heads;   /* 12 (dino.py:87) */
    const float* bottleneck_w; const float* bottleneck_b; /* [emb, emb_in] or NULL */
    const float* slice_pos_emb;                           /* [256, emb] or NULL (dino.py:82) */
    const float* cls_token;                               /* [emb] (dino.py:97) */
    const float* ln1_w; const float* ln1_b;               /* transformer_blocks.py:499 */
    const float* in_proj_w; const float* in_proj_b;       /* [3emb, emb] */
    const float* out_proj_w; const float* out_proj_b;
    const float* ln2_w; const float* ln2_b;
    const float* lin1_w; const float* lin1_b;             /* dim_feedforward = emb (dino.py:88) */
    const float* lin2_w; const float* lin2_b;
    const float* norm_w; const float* norm_b;             /* dino.py:95 */
    const float* rope_freqs;                              /* [head_dim/2] or NULL (RoPE) */
    const float* head_w; const float* head_b;             /* [out_ch, emb*] (dino.py:103) */
    const float* liere_rot;                               /* [head_dim, head_dim] from mst_liere_rotation, or NULL (LieRE) */
    int head_in;     /* columns of head_w: emb, or 32*emb for 'linear' fusion (dino.py:245).  The feature width of the call
                      * (emb*D for 'linear') must equal it, as nn.Linear would insist; 0 = unchecked */
} mst_fusion_weights;

/* dino.py:134-166 after the encoder: bottleneck, slice position embedding, CLS concat,
 * nn.TransformerEncoder(1 layer, norm_first) + final LayerNorm (transformer_blocks.py:565-587,
 * 29-318), row 0, linear head.  emb fp32 [B*D, emb_in]; key_padding_mask uint8 [B, D]
 * (1 = padded slice; NULL = none) -- the CLS column is prepended inside (dino.py:147-150).
 * features fp32 [B, F] (F = emb; emb*D for 'linear'), logits fp32 [B, out_ch] (nullable),
 * slice_probs fp32 [B, heads, 1+D, 1+D] (nullable; transformer_blocks.py:266-295). */
size_t mst_fusion_workspace_bytes(const mst_fusion_weights* w, int B, int D);
int mst_slice_fusion(const mst_fusion_weights* w, const float* emb, int B, int D,
                     const uint8_t* key_padding_mask, float* features, float* logits,
                     float* slice_probs, void* ws, size_t ws_bytes, mst_stream_t stream);

/* Saliency read-outs: dino.py:173-202.  cls_probs_last fp32 [n, heads, N] (N = 1+R+Np),
 * slice_probs fp32 [B, sheads, 1+D, 1+D], n = B*D.  Outputs (each nullable): plane fp32
 * [n, heads, Np], slice_attn fp32 [n], maps fp32 [n, heads, Np]. */
int mst_attention_readout(const float* cls_probs_last, const float* slice_probs, int B, int D,
                          int heads, int N, int num_registers, int sheads, float* plane,
                          float* slice_attn, float* maps, mst_stream_t stream);

/* Saliency volume of `--get_attention` (scripts/main_predict.py:72-105, 147-165: what _pred_trans / run_pred do after
 * the forward of ONE volume).  mst_saliency_accumulate: lowres[D, gh, gw] (+)= head-mean of maps fp32 [D, heads, Np]
 * (the get_attention_maps() result; columns >= gh*gw ignored, l.94-96) with the axes named in flip_mask (bit 0 depth,
 * bit 1 height, bit 2 width) mirrored back -- the test-time-augmentation sum of l.147-153; slice_acc[D] (+)= slice_attn[D]
 * likewise (nullable).  accumulate = 0 overwrites.  mst_saliency_upsample: out fp32 [Dout, H, W] = scale * trilinear
 * (align_corners=False, F.interpolate at l.163) of lowres. */
int mst_saliency_accumulate(const float* maps, const float* slice_attn, int D, int heads, int gh, int gw, int Np,
                            int flip_mask, int accumulate, float* lowres, float* slice_acc, mst_stream_t stream);
int mst_saliency_upsample(const float* lowres, int D, int gh, int gw, float scale, int Dout, int H, int W, float* out,
                          mst_stream_t stream);

/* LieRE rotation of the slice transformer (AttentionLiereRotator, rotary_embedding_torch.py:319-372;
 * transformer_blocks.py:350-357): vars fp32 [n_blocks, block(block-1)/2, axes_length] (the module's ParameterList
 * stacked; spacial_dims = 1) -> R fp32 [n_blocks*block, n_blocks*block], block-diagonal, R_blk = exp(A_blk) with the
 * skew generator A_blk[i][j] = sum_p p * vars[blk][tril(i,j)][p].  block <= 16.  Weight preparation: call once per
 * weight version and pass R as mst_fusion_weights.liere_rot. */
int mst_liere_rotation(const float* vars, int n_blocks, int block, int axes_length, float* R, mst_stream_t stream);

/* Attention rollout (dino.py:204-212, get_attention_cls): R = maps[L-1]; for l = L-2 .. 0: R = maps[l] @ R,
 * every map fp32 [batch, N, N] row-major (batch = n*heads), exact fp32 MFMA.  `maps` is a HOST array of
 * n_layers device pointers.  The product lands in `out`; `tmp` (same size as out) is scratch and may be NULL
 * when n_layers <= 2.  Neither may alias a map.  n_layers == 1 copies the map. */
int mst_attention_rollout(const float* const* maps, int n_layers, int64_t batch, int N, float* out,
                          float* tmp, mst_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MST_HIP_H */
