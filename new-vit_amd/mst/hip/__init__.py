"""ctypes binding of libmst_hip.so (C ABI: include/mst_hip.h).

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``), the current HIP
stream and dtype tags.  There is NO fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised (the product path never routes through torch ops or the CPU oracle).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Sequence, Optional, Tuple

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MST_HIP_LIB", _HERE / "libmst_hip.so"))

F32, F16, BF16, F8E4M3 = 0, 1, 2, 3
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_RESIDUAL, EPI_RESIDUAL_RELU = 0, 1, 2, 3, 4
FUSION_TRANSFORMER, FUSION_LINEAR, FUSION_AVERAGE = 0, 1, 2

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
TORCH_DT = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}
DT_NAMES = {"fp32": F32, "f32": F32, "float32": F32, "fp16": F16, "f16": F16, "float16": F16,
            "bf16": BF16, "bfloat16": BF16,
            # e4m3 linear layers on a bf16 carrier (LayerNorm outputs, q/k/v, attention): mst_vit_weights.fp8_linear
            "fp8": BF16, "f8": BF16, "fp8_e4m3": BF16}
FP8_NAMES = ("fp8", "f8", "fp8_e4m3")

_vp, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t


class VitLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
                                   "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2",
                                   "qkv_wf", "qkv_bf", "mlp_pack", "fc1_bf", "fc2_bf", "proj_pack", "proj_bf", "block_seq",
                                   "qkv_w8", "proj_w8", "fc1_w8", "fc2_w8")] + [("w8_scale", C.c_float * 4)]


class VitWeights(C.Structure):
    _fields_ = [("embed_dim", _i), ("depth", _i), ("num_heads", _i), ("num_registers", _i),
                ("compute_dtype", _i), ("grid_h", _i), ("grid_w", _i),
                ("patch_w", _vp), ("patch_b", _vp), ("prefix", _vp), ("pos_patch", _vp),
                ("layers", C.POINTER(VitLayer)), ("norm_w", _vp), ("norm_b", _vp), ("fp8_linear", _i),
                ("fp8_amax", _vp), ("fp8_amax_out", _vp), ("profiler", _vp), ("prune_last_block", _i)]


class FusionWeights(C.Structure):
    _fields_ = [("emb_in", _i), ("emb", _i), ("out_ch", _i), ("fusion_type", _i), ("num_heads", _i)] + \
               [(n, _vp) for n in ("bottleneck_w", "bottleneck_b", "slice_pos_emb", "cls_token",
                                   "ln1_w", "ln1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
                                   "ln2_w", "ln2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                                   "norm_w", "norm_b", "rope_freqs", "head_w", "head_b", "liere_rot")] + [("head_in", _i)]


# symbol -> (restype, argtypes); tests check every symbol of include/mst_hip.h is exported
SIGNATURES = {
    "mst_version": (_i, []),
    "mst_last_error": (C.c_char_p, []),
    "mst_layernorm": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i64, _i64, _i, _f, _vp]),
    "mst_gemm": (_i, [_vp, _i, _i64, _vp, _i64, _vp, _vp, _i, _i64, _i64, _i, _i, _i, _vp, _f, _i, _vp]),
    "mst_quantize_fp8": (_i, [_vp, _i, _i64, _vp, _vp, _vp]),
    "mst_gemm_fp8": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _f, _vp, _i, _i64, _i64, _i, _i, _i, _vp, _f, _i, _vp, _vp]),
    "mst_layernorm_fp8": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i, _f, _vp, _vp]),
    "mst_attention": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_attention_cls_probs": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_attention_probs_full": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_pos_embed_interp": (_i, [_vp, _i, _i, _i, _i, _d, _i, _vp, _vp]),
    "mst_mlp_fused": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "mst_block_fused_scratch_bytes": (_sz, []),
    "mst_block_fused": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _i, _f, _vp]),
    "mst_block_fused_s": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "mst_gemm_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(_i64), _i, _i, _f, _f, _vp]),
    "mst_softmax_rows": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "mst_softmax_rows_bwd": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "mst_layernorm_bwd": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _f, _vp]),
    "mst_act_fwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mst_act_bwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mst_colsum": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp, _vp]),
    "mst_axpby_cols": (_i, [_vp, _i64, _vp, _f, _f, _vp, _i64, _i64, _i, _vp]),
    "mst_im2col14": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mst_pos_embed_interp_bwd": (_i, [_vp, _i, _i, _i, _i, _d, _vp, _vp]),
    "mst_im2col_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_cvt16": (_i, [_vp, _i64, _i64, _i, _f, _vp, _i, _i64, _i, _i64, _vp]),
    "mst_gemm16_splitk": (_i, [_vp, _i, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _i64, _vp]),
    "mst_conv_gemm": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mst_conv_gemm16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mst_rope_rows": (_i, [_vp, _i64, _i, _i, _i, _vp, _f, _vp]),
    "mst_cvt32": (_i, [_vp, _i, _i64, _vp, _vp]),
    "mst_im2col_nhwc16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mst_maxpool_nhwc16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_conv_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i64, _vp]),
    "mst_conv_wgrad16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i64, _vp]),
    "mst_conv_dgrad": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "mst_maxpool_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mst_avgpool_nhwc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mst_batchnorm_train": (_i, [_vp, _i64, _i, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mst_batchnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "mst_col2im_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_maxpool_bwd_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mst_avgpool_bwd_nhwc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mst_gradcampp": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mst_crop_or_pad": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "mst_znorm_state_bytes": (_sz, []),
    "mst_znorm": (_i, [_vp, _i64, _f, _f, _vp, _vp, _vp]),
    "mst_slices2rgb": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mst_patch_embed": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "mst_vit_workspace_bytes": (_sz, [C.POINTER(VitWeights), _i, _i, _i]),
    "mst_vit_encode": (_i, [C.POINTER(VitWeights), _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "mst_fusion_workspace_bytes": (_sz, [C.POINTER(FusionWeights), _i, _i]),
    "mst_slice_fusion": (_i, [C.POINTER(FusionWeights), _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mst_attention_readout": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mst_saliency_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mst_saliency_upsample": (_i, [_vp, _i, _i, _i, _f, _i, _i, _i, _vp, _vp]),
    "mst_liere_rotation": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mst_attention_rollout": (_i, [C.POINTER(_vp), _i, _i64, _i, _vp, _vp, _vp]),
    "mst_profiler_create": (_vp, []),
    "mst_profiler_destroy": (None, [_vp]),
    "mst_profiler_collect": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mst_kernel_kind_name": (C.c_char_p, [_i]),
}
K_COUNT = 10
ABI_VERSION = 300        # mst_version(): round 3 (mst_vit_layer.block_seq, mst_block_fused_s)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library once; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"libmst_hip.so not found at {LIB_PATH}: build it with `python new-vit_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback for the MST hot path.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mst_version() != ABI_VERSION:             # the struct mirrors below follow include/mst_hip.h of exactly this version
            raise RuntimeError(f"{LIB_PATH} reports ABI version {lib.mst_version()}, this binding was written for {ABI_VERSION}: "
                               "rebuild with `python new-vit_amd/build.py --force`")
        _lib = lib
    return _lib


def last_error() -> str:
    return (load().mst_last_error() or b"").decode()


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")


def dt_of(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor must live on a HIP device (got {t.device}); the MST kernels have no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"{what}: tensor must be contiguous")


# ---- per-op wrappers (unit parity tests call these; the models call the orchestrators) --------
def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
              out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    _dev(x, "layernorm")
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _check(load().mst_layernorm(ptr(x), cols, ptr(weight), ptr(bias), ptr(out), _DT[out_dtype], cols, rows, cols,
                                eps, stream_of(x)), "mst_layernorm")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, epilogue: int = EPI_BIAS,
         out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
         gamma: Optional[torch.Tensor] = None, col_scale: float = 1.0, scale_cols: int = 0) -> torch.Tensor:
    _dev(a, "gemm")
    _dev(w, "gemm")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
    _check(load().mst_gemm(ptr(a), dt_of(a), K, ptr(w), K, ptr(bias), ptr(out), dt_of(out), N, M, N, K, epilogue,
                           ptr(gamma), col_scale, scale_cols, stream_of(a)), "mst_gemm")
    return out


F8_MAX = 448.0      # largest finite OCP e4m3


def quantize_weight_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, float]:
    """Per-tensor e4m3 form of a weight matrix: (bytes [out,in] uint8, scale = max|W|/448), W ~ scale * e4m3(bytes).  Done
    once per weight version with torch's round-to-nearest-even cast (host-side preparation, like the other packers)."""
    w = w.detach().to(torch.float32)
    amax = float(w.abs().max())
    scale = amax / F8_MAX if amax > 0 else 1.0
    q = (w / scale).clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale


def quantize_fp8(x: torch.Tensor, amax: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mst_quantize_fp8: (e4m3 bytes with x's shape, amax fp32 [1] on the device)."""
    _dev(x, "quantize_fp8")
    if amax is None:
        amax = torch.zeros(1, dtype=torch.float32, device=x.device)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _check(load().mst_quantize_fp8(ptr(x), dt_of(x), x.numel(), ptr(amax), ptr(out), stream_of(x)), "mst_quantize_fp8")
    return out, amax


def gemm_fp8(a8: torch.Tensor, a_amax: torch.Tensor, w8: torch.Tensor, w_scale: float, bias: Optional[torch.Tensor], *,
             epilogue: int = EPI_BIAS, out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32,
             gamma: Optional[torch.Tensor] = None, col_scale: float = 1.0, scale_cols: int = 0,
             c_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c_amax given: the output is e4m3 bytes (uint8) under that calibrated scale."""
    _dev(a8, "gemm_fp8")
    _dev(w8, "gemm_fp8")
    M, K = a8.shape
    N = w8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.uint8 if c_amax is not None else out_dtype, device=a8.device)
    cdt = F8E4M3 if c_amax is not None else dt_of(out)
    _check(load().mst_gemm_fp8(ptr(a8), K, ptr(w8), K, ptr(bias), ptr(a_amax), w_scale, ptr(out), cdt, N, M, N, K,
                               epilogue, ptr(gamma), col_scale, scale_cols, ptr(c_amax), stream_of(a8)), "mst_gemm_fp8")
    return out


def layernorm_fp8(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                  amax: torch.Tensor) -> torch.Tensor:
    """mst_layernorm_fp8: LayerNorm over the last dim of fp32 x [rows, cols] -> e4m3 bytes under the calibrated scale."""
    _dev(x, "layernorm_fp8")
    rows, cols = x.shape
    out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    _check(load().mst_layernorm_fp8(ptr(x), cols, ptr(gamma), ptr(beta), ptr(out), cols, rows, cols, eps, ptr(amax),
                                    stream_of(x)), "mst_layernorm_fp8")
    return out


def attention(qkv: torch.Tensor, n_seq: int, N: int, heads: int, head_dim: int = 64) -> torch.Tensor:
    _dev(qkv, "attention")
    out = torch.empty((n_seq * N, heads * head_dim), dtype=qkv.dtype, device=qkv.device)
    _check(load().mst_attention(ptr(qkv), dt_of(qkv), n_seq, N, heads, head_dim, ptr(out), stream_of(qkv)), "mst_attention")
    return out


def attention_cls_probs(qkv: torch.Tensor, n_seq: int, N: int, heads: int, head_dim: int = 64) -> torch.Tensor:
    _dev(qkv, "attention_cls_probs")
    out = torch.empty((n_seq, heads, N), dtype=torch.float32, device=qkv.device)
    _check(load().mst_attention_cls_probs(ptr(qkv), dt_of(qkv), n_seq, N, heads, head_dim, ptr(out), stream_of(qkv)),
           "mst_attention_cls_probs")
    return out


def attention_probs_full(qkv: torch.Tensor, n_seq: int, N: int, heads: int, head_dim: int = 64) -> torch.Tensor:
    _dev(qkv, "attention_probs_full")
    out = torch.empty((n_seq, heads, N, N), dtype=torch.float32, device=qkv.device)
    _check(load().mst_attention_probs_full(ptr(qkv), dt_of(qkv), n_seq, N, heads, head_dim, ptr(out), stream_of(qkv)),
           "mst_attention_probs_full")
    return out


def pos_embed_interp(pos_patch: torch.Tensor, M: int, gh: int, gw: int, offset: float = 0.1,
                     antialias: bool = False) -> torch.Tensor:
    _dev(pos_patch, "pos_embed_interp")
    E = pos_patch.shape[-1]
    out = torch.empty((gh * gw, E), dtype=torch.float32, device=pos_patch.device)
    _check(load().mst_pos_embed_interp(ptr(pos_patch), M, E, gh, gw, offset, 1 if antialias else 0, ptr(out), stream_of(pos_patch)),
           "mst_pos_embed_interp")
    return out


def patch_embed(vol: torch.Tensor, wp: torch.Tensor, bias: torch.Tensor, prefix: torch.Tensor,
                pos_patch: torch.Tensor) -> torch.Tensor:
    _dev(vol, "patch_embed")
    n, H, W = vol.shape
    E = wp.shape[0]
    n_prefix = prefix.shape[0]
    N = n_prefix + (H // 14) * (W // 14)
    x = torch.empty((n, N, E), dtype=torch.float32, device=vol.device)
    _check(load().mst_patch_embed(ptr(vol), dt_of(vol), n, H, W, ptr(wp), dt_of(wp), ptr(bias), ptr(prefix), n_prefix,
                                  ptr(pos_patch), E, ptr(x), stream_of(vol)), "mst_patch_embed")
    return x


def slices2rgb(vol: torch.Tensor) -> torch.Tensor:
    """mst_slices2rgb: [B,1,D,H,W] -> [B*ceil(D/3), 3, H, W] (reference dino.py:10-27)."""
    _dev(vol, "slices2rgb")
    B, C, D, H, W = vol.shape
    assert C == 1, "More than one channel"              # dino.py:14
    Dp = (D + 2) // 3 * 3
    out = torch.empty((B * Dp // 3, 3, H, W), dtype=vol.dtype, device=vol.device)
    _check(load().mst_slices2rgb(ptr(vol), dt_of(vol), B, D, H, W, ptr(out), stream_of(vol)), "mst_slices2rgb")
    return out


def mlp_fused(x: torch.Tensor, wpack: torch.Tensor, b1f: torch.Tensor, b2f: torch.Tensor,
              xn_out: Optional[torch.Tensor], dtype: torch.dtype, eps: float = 1e-6):
    """x [M,384] fp32 updated in place; xn_out (optional, `dtype`) receives normalise(x_new).
    wpack / b1f / b2f from pack_mlp (LayerNorm affine and LayerScale folded)."""
    _dev(x, "mlp_fused")
    M, E = x.shape
    _check(load().mst_mlp_fused(ptr(x), ptr(xn_out), _DT[dtype], ptr(wpack), ptr(b1f), ptr(b2f), M, E, eps,
                                stream_of(x)), "mst_mlp_fused")


def block_fused(x: torch.Tensor, attn_out: torch.Tensor, proj_pack: torch.Tensor, proj_bf: torch.Tensor, wpack: torch.Tensor,
                b1f: torch.Tensor, b2f: torch.Tensor, xn_out: Optional[torch.Tensor], eps: float = 1e-6,
                scratch: Optional[torch.Tensor] = None):
    """mst_block_fused: x [M,384] fp32 in place += ls1*proj(attn_out) then += the fused MLP; xn_out (may be attn_out itself)
    receives normalise(x_new).  proj_pack / proj_bf from pack_proj, wpack / b1f / b2f from pack_mlp."""
    _dev(x, "block_fused")
    _dev(attn_out, "block_fused")
    M, E = x.shape
    if scratch is None:
        scratch = torch.empty(int(load().mst_block_fused_scratch_bytes()), dtype=torch.uint8, device=x.device)
    _check(load().mst_block_fused(ptr(x), ptr(attn_out), ptr(xn_out), dt_of(attn_out), ptr(proj_pack), ptr(proj_bf), ptr(wpack),
                                  ptr(b1f), ptr(b2f), ptr(scratch), scratch.numel(), M, E, eps, stream_of(x)), "mst_block_fused")


def block_fused_s(x: torch.Tensor, attn_out: torch.Tensor, block_seq: torch.Tensor, b1f: torch.Tensor, proj_bf: torch.Tensor,
                  b2f: torch.Tensor, xn_out: Optional[torch.Tensor], eps: float = 1e-6, layout: int = 0):
    """mst_block_fused_s (single-role form of mst_block_fused): x [M,384] fp32 in place += ls1*proj(attn_out) then += the fused
    MLP; xn_out (may be attn_out itself) receives normalise(x_new).  block_seq / b1f / proj_bf / b2f from pack_block_seq."""
    _dev(x, "block_fused_s")
    _dev(attn_out, "block_fused_s")
    M, E = x.shape
    _check(load().mst_block_fused_s(ptr(x), ptr(attn_out), ptr(xn_out), dt_of(attn_out), ptr(block_seq), ptr(b1f), ptr(proj_bf),
                                    ptr(b2f), M, E, eps, layout, stream_of(x)), "mst_block_fused_s")


LAYOUT_X_IN_IMAGE, LAYOUT_X_OUT_IMAGE, LAYOUT_ACT_BLOCKED = 1, 2, 4


def to_blocked16(a: torch.Tensor) -> torch.Tensor:
    """Row-major [M, C] (M % 32 == 0, C % 16 == 0) -> the 16-bit "blocked" layout of include/mst_hip.h (test / tooling helper)."""
    M, Cn = a.shape
    return a.reshape(M // 32, 32, Cn // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(M, Cn)


def from_blocked16(a: torch.Tensor) -> torch.Tensor:
    M, Cn = a.shape
    return a.reshape(M // 32, Cn // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).contiguous().reshape(M, Cn)


def to_image32(a: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 [M, C] (M % 32 == 0, C % 8 == 0) -> the fp32 "image" layout of include/mst_hip.h."""
    M, Cn = a.shape
    return a.reshape(M // 32, 32, Cn // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(M, Cn)


def from_image32(a: torch.Tensor) -> torch.Tensor:
    M, Cn = a.shape
    return a.reshape(M // 32, Cn // 8, 2, 32, 4).permute(0, 3, 1, 2, 4).contiguous().reshape(M, Cn)


def pack_block_seq(proj_w, proj_b, ls1, fc1_w, fc1_b, fc2_w, fc2_b, ln_w, ln_b, ls2, dtype: torch.dtype):
    """Host-side packing of one block's out-projection + MLP weights for mst_block_fused_s (layout: include/mst_hip.h): one
    stream of 108 x 24 KiB elements in the order the kernel consumes them, every element 24 MFMA A-fragments in lane order.
    LayerNorm2's affine is folded into fc1, LayerScale into proj / fc2 (layer_scale.py:26-27).  Inputs fp32 (any device):
    proj_w [384,384], proj_b [384], fc1_w [1536,384], fc1_b [1536], fc2_w [384,1536], fc2_b [384], ln_w / ln_b [384], ls1 /
    ls2 [384] or None.  Returns (block_seq [108, 12288] dtype, b1f [1536] fp32, proj_bf [384] fp32, b2f [384] fp32)."""
    dev = proj_w.device
    wp = proj_w.float()
    pbf = proj_b.float().clone()
    if ls1 is not None:
        wp = wp * ls1.float()[:, None]
        pbf = pbf * ls1.float()
    w1 = fc1_w.float() * ln_w.float()[None, :]
    b1f = fc1_b.float() + (fc1_w.float() * ln_b.float()[None, :]).sum(dim=1)      # (weight preparation without a vendor BLAS call)
    w2 = fc2_w.float()
    b2f = fc2_b.float().clone()
    if ls2 is not None:
        w2 = w2 * ls2.float()[:, None]
        b2f = b2f * ls2.float()
    ar = lambda n: torch.arange(n, device=dev)
    lane = ar(64)
    m, h = (lane & 31)[None, None, :, None], (lane >> 5)[None, None, :, None]          # [t, p, lane, e]
    t, p, e = ar(12)[:, None, None, None], ar(2)[None, :, None, None], ar(8)[None, None, None, :]
    k8 = 16 * p + 8 * (e >> 2) + 4 * h + (e & 3)                                       # accumulator-tile k order within 32
    rows_t = (32 * t + m).expand(12, 2, 64, 8)
    # out-projection chunk j: Wp[32t+m][32j + 16p + 8h + e]
    kp = (16 * p + 8 * h + e).expand(12, 2, 64, 8)
    proj = torch.stack([wp[rows_t, 32 * j + kp] for j in range(12)])                   # [12, t, p, lane, e]
    # W1 chunk c, fragment = k-step 2t+p: W1[32c+m][32t + k8]
    cols1 = (32 * t + k8).expand(12, 2, 64, 8)
    rows1 = m.expand(12, 2, 64, 8)
    w1c = torch.stack([w1[32 * c + rows1, cols1] for c in range(48)])                  # [48, t, p, lane, e]
    # W2 chunk c, fragment 2t+p: W2[32t+m][32c + k8]
    k2 = k8.expand(12, 2, 64, 8)
    w2c = torch.stack([w2[rows_t, 32 * c + k2] for c in range(48)])
    proj, w1c, w2c = proj.reshape(12, -1), w1c.reshape(48, -1), w2c.reshape(48, -1)
    seq = [proj, w1c[0:1]]
    inter = torch.stack([w1c[1:48], w2c[0:47]], dim=1).reshape(94, -1)                 # W1(c+1), W2(c) for c = 0..46
    seq += [inter, w2c[47:48]]
    block_seq = torch.cat(seq, dim=0).to(dtype).contiguous()
    assert block_seq.shape == (108, 12288)
    return block_seq, b1f.contiguous(), pbf.contiguous(), b2f.contiguous()


def _w2_row_order(dev):
    R = torch.arange(384, device=dev)
    t, i = R >> 4, R & 15
    nphys = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3)          # accumulator row R holds output column n(R)
    cidx = torch.arange(4, device=dev)[None, :] ^ ((-(R >> 2)) & 3)[:, None]   # [R, slot] -> 16-byte chunk (bank swizzle)
    return R, nphys, cidx


def pack_proj(proj_w, proj_b, ls1, dtype: torch.dtype):
    """Host-side packing of the attention out-projection for mst_block_fused (layout: include/mst_hip.h).  proj_w [384,384],
    proj_b [384], ls1 [384] or None (LayerScale folded: layer_scale.py:26-27).  Returns (proj_pack [12, 12288] dtype,
    proj_bf [384] fp32)."""
    dev = proj_w.device
    w = proj_w.float()
    b = proj_b.float().clone()
    if ls1 is not None:
        w = w * ls1.float()[:, None]
        b = b * ls1.float()
    R, nphys, cidx = _w2_row_order(dev)
    wc = w[nphys].view(384, 12, 4, 8).permute(1, 0, 2, 3)                   # [chunk, R, c, j]: natural k order inside a chunk
    img = wc[:, R[:, None], cidx, :]                                        # [chunk, R, slot, 8]
    return img.reshape(12, -1).to(dtype).contiguous(), b.contiguous()


def pack_mlp(fc1_w, fc1_b, fc2_w, fc2_b, ln_w, ln_b, ls2, dtype: torch.dtype):
    """Host-side packing of one MLP for mst_mlp_fused (layout: include/mst_hip.h).  Inputs fp32 tensors on any
    device: fc1_w [1536,384], fc1_b [1536], fc2_w [384,1536], fc2_b [384], ln_w/ln_b [384], ls2 [384] or None.
    Returns (wpack [48,24576] dtype, b1f [1568] fp32, b2f [384] fp32)."""
    dev = fc1_w.device
    w1f = fc1_w.float() * ln_w.float()[None, :]
    b1f = fc1_b.float() + (fc1_w.float() * ln_b.float()[None, :]).sum(dim=1)      # (weight preparation without a vendor BLAS call)
    b2f = fc2_b.float().clone()
    if ls2 is not None:                                   # LayerScale folded into fc2 (layer_scale.py:26-27)
        fc2_w = fc2_w.float() * ls2.float()[:, None]
        b2f = b2f * ls2.float()
    ar = lambda n: torch.arange(n, device=dev)
    # W1 image: [chunk][ks][h][slot][8]
    w1c = w1f.view(48, 32, 12, 4, 8).permute(0, 2, 1, 3, 4)                    # [chunk, ks, h, c, j]
    fh = (-(ar(32) >> 2)) & 3
    cidx = ar(4)[None, :] ^ fh[:, None]                                         # [h, slot] -> c
    w1img = w1c[:, :, ar(32)[:, None], cidx, :]                                 # [chunk, ks, h, slot, 8]
    # W2 image: [chunk][R][slot][8]
    p = ar(32); c = p >> 3; j = p & 7
    phys = torch.where(j < 4, 4 * c + j, 16 + 4 * c + j - 4)
    R = ar(384); t = R >> 4; i = R & 15
    nphys = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3)
    w2p = fc2_w.float().view(384, 48, 32)[:, :, phys][nphys]                    # [R, chunk, p]
    w2c = w2p.permute(1, 0, 2).reshape(48, 384, 4, 8)                           # [chunk, R, c, j]
    fR = (-(R >> 2)) & 3
    cidxR = ar(4)[None, :] ^ fR[:, None]
    w2img = w2c[:, R[:, None], cidxR, :]                                        # [chunk, R, slot, 8]
    wpack = torch.cat([w1img.reshape(48, -1), w2img.reshape(48, -1)], dim=1).to(dtype).contiguous()
    b1p = torch.zeros(1536 + 32, device=dev, dtype=torch.float32)
    b1p[:1536] = b1f
    return wpack, b1p, b2f.contiguous()


def vit_workspace_bytes(w: VitWeights, H: int, W: int, chunk: int) -> int:
    return int(load().mst_vit_workspace_bytes(C.byref(w), H, W, chunk))


def vit_encode(w: VitWeights, vol: torch.Tensor, cls_out: torch.Tensor, cls_probs: Optional[torch.Tensor],
               n_layers_probs: int, chunk: int, ws: torch.Tensor, full_probs: Optional[torch.Tensor] = None):
    _dev(vol, "vit_encode")
    n, H, W = vol.shape
    _check(load().mst_vit_encode(C.byref(w), ptr(vol), dt_of(vol), n, H, W, ptr(cls_out), ptr(cls_probs),
                                 ptr(full_probs), n_layers_probs, chunk, ptr(ws), ws.numel() * ws.element_size(),
                                 stream_of(vol)), "mst_vit_encode")


def fusion_workspace_bytes(w: FusionWeights, B: int, D: int) -> int:
    return int(load().mst_fusion_workspace_bytes(C.byref(w), B, D))


def slice_fusion(w: FusionWeights, emb: torch.Tensor, B: int, D: int, mask: Optional[torch.Tensor],
                 features: torch.Tensor, logits: Optional[torch.Tensor], slice_probs: Optional[torch.Tensor],
                 ws: torch.Tensor):
    _dev(emb, "slice_fusion")
    _check(load().mst_slice_fusion(C.byref(w), ptr(emb), B, D, ptr(mask), ptr(features), ptr(logits), ptr(slice_probs),
                                   ptr(ws), ws.numel() * ws.element_size(), stream_of(emb)), "mst_slice_fusion")


def attention_readout(cls_probs_last: Optional[torch.Tensor], slice_probs: Optional[torch.Tensor], B: int, D: int,
                      heads: int, N: int, num_registers: int, sheads: int, plane: Optional[torch.Tensor],
                      slice_attn: Optional[torch.Tensor], maps: Optional[torch.Tensor]):
    t = cls_probs_last if cls_probs_last is not None else slice_probs
    _check(load().mst_attention_readout(ptr(cls_probs_last), ptr(slice_probs), B, D, heads, N, num_registers, sheads,
                                        ptr(plane), ptr(slice_attn), ptr(maps), stream_of(t)), "mst_attention_readout")


def saliency_accumulate(maps: torch.Tensor, slice_attn: Optional[torch.Tensor], gh: int, gw: int, flip_mask: int,
                        lowres: torch.Tensor, slice_acc: Optional[torch.Tensor], accumulate: bool):
    """lowres [D,gh,gw] (+)= head-mean of maps [D,heads,Np], TTA flips (bit 0 depth, 1 height, 2 width) mirrored back."""
    _dev(maps, "saliency_accumulate")
    D, heads, Np = maps.shape
    _check(load().mst_saliency_accumulate(ptr(maps), ptr(slice_attn), D, heads, gh, gw, Np, flip_mask, 1 if accumulate else 0,
                                          ptr(lowres), ptr(slice_acc), stream_of(maps)), "mst_saliency_accumulate")


def saliency_upsample(lowres: torch.Tensor, size, scale: float = 1.0) -> torch.Tensor:
    """scale * F.interpolate(lowres[None, None], size, mode='trilinear') -> [Dout, H, W] fp32."""
    _dev(lowres, "saliency_upsample")
    D, gh, gw = lowres.shape
    Dout, H, W = (int(v) for v in size)
    out = torch.empty((Dout, H, W), dtype=torch.float32, device=lowres.device)
    _check(load().mst_saliency_upsample(ptr(lowres), D, gh, gw, float(scale), Dout, H, W, ptr(out), stream_of(lowres)),
           "mst_saliency_upsample")
    return out


def liere_rotation(vars_: Sequence[torch.Tensor]) -> torch.Tensor:
    """AttentionLiereRotator's rotation R [hd, hd] from its ParameterList (each [n(n-1)/2, axes_length, 1])."""
    v = torch.stack([t.detach().float().reshape(t.shape[0], t.shape[1]) for t in vars_]).contiguous()
    _dev(v, "liere_rotation")
    nb, m, P = v.shape
    n = int(round((1 + (1 + 8 * m) ** 0.5) / 2))
    if n * (n - 1) // 2 != m:
        raise ValueError(f"liere_rotation: {m} parameters per block is not n(n-1)/2")
    R = torch.empty((nb * n, nb * n), dtype=torch.float32, device=v.device)
    _check(load().mst_liere_rotation(ptr(v), nb, n, P, ptr(R), stream_of(v)), "mst_liere_rotation")
    return R


def attention_rollout(maps: Sequence[torch.Tensor]) -> torch.Tensor:
    """get_attention_cls (dino.py:204-212): maps[0] @ maps[1] @ ... @ maps[-1] on full fp32 [..., N, N] maps."""
    m0 = maps[0]
    _dev(m0, "attention_rollout")
    N = m0.shape[-1]
    for m in maps:
        if m.dtype != torch.float32 or tuple(m.shape) != tuple(m0.shape) or m.shape[-2] != N or not m.is_contiguous():
            raise ValueError("attention_rollout: maps must be contiguous fp32 [..., N, N] tensors of one shape")
    batch = m0.numel() // (N * N)
    out = torch.empty_like(m0)
    tmp = torch.empty_like(m0) if len(maps) > 2 else None
    arr = (_vp * len(maps))(*[ptr(m) for m in maps])
    _check(load().mst_attention_rollout(arr, len(maps), batch, N, ptr(out), ptr(tmp), stream_of(m0)),
           "mst_attention_rollout")
    return out


# ---- training-step ops (fp32; orchestrated by mst/train.py) ---------------------------------------------------------
def gemm_ex(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, sa, sb, sc, nb=(1, 1),
            ba=(0, 0), bb=(0, 0), bc=(0, 0), alpha: float = 1.0, beta: float = 0.0, offs=(0, 0, 0)):
    """C[b] = alpha * A[b] . B[b] + beta * C[b]; sa = (A_m, A_k), sb = (B_k, B_n), sc = (C_m, C_n) element strides, nb = (nb1, nb2),
    ba / bb / bc = per-level batch strides, offs = element offsets of A, B, C into their tensors (views without copies)."""
    st = (_i64 * 12)(sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], ba[0], ba[1], bb[0], bb[1], bc[0], bc[1])
    _check(load().mst_gemm_ex(A.data_ptr() + 4 * offs[0], B.data_ptr() + 4 * offs[1], C.data_ptr() + 4 * offs[2], M, N, K, st,
                              nb[0], nb[1], alpha, beta, stream_of(C)), "mst_gemm_ex")
    return C


def cvt16(x: torch.Tensor, dtype: torch.dtype, *, transpose: bool = False, rows_pad: Optional[int] = None, scale: float = 1.0) -> torch.Tensor:
    """mst_cvt16: the 16-bit image of an fp32 matrix [rows, cols] -- the same layout, or transposed [cols, rows_pad] with zero columns for
    rows .. rows_pad (operands of the mixed-precision training GEMMs)."""
    _dev(x, "cvt16")
    rows, cols = x.shape
    if transpose:
        rp = rows_pad or rows
        out = torch.empty((cols, rp), dtype=dtype, device=x.device)
        _check(load().mst_cvt16(ptr(x), cols, rows, cols, scale, ptr(out), dt_of(out), rp, 1, rp, stream_of(x)), "mst_cvt16")
        return out
    out = torch.empty((rows, cols), dtype=dtype, device=x.device)
    _check(load().mst_cvt16(ptr(x), cols, rows, cols, scale, ptr(out), dt_of(out), cols, 0, rows, stream_of(x)), "mst_cvt16")
    return out


def gemm16_splitk(a: torch.Tensor, w: torch.Tensor, splits: int) -> torch.Tensor:
    """mst_gemm16_splitk: partial products [splits, M, N] fp32 of a [M, K] . w [N, K]^T over `splits` equal K ranges (16-bit operands)."""
    _dev(a, "gemm16_splitk")
    _dev(w, "gemm16_splitk")
    M, K = a.shape
    N = w.shape[0]
    part = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
    _check(load().mst_gemm16_splitk(ptr(a), dt_of(a), K, ptr(w), K, ptr(part), N, M, N, K, splits, M * N, stream_of(a)), "mst_gemm16_splitk")
    return part


def rope_rows(qkv: torch.Tensor, L: int, heads: int, hd: int, freqs: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """mst_rope_rows: rotate the q and k pairs of packed rows [rows, 3*heads*hd] in place (sign -1: the adjoint, for gradient rows)."""
    _dev(qkv, "rope_rows")
    _check(load().mst_rope_rows(ptr(qkv), qkv.shape[0], L, heads, hd, ptr(freqs), sign, stream_of(qkv)), "mst_rope_rows")
    return qkv


def softmax_rows(S: torch.Tensor, mask: Optional[torch.Tensor], rows_per_batch: int):
    L = S.shape[-1]
    _check(load().mst_softmax_rows(ptr(S), ptr(mask), S.numel() // L, L, rows_per_batch, stream_of(S)), "mst_softmax_rows")
    return S


def softmax_rows_bwd(P: torch.Tensor, dP: torch.Tensor, scale: float = 1.0):
    L = P.shape[-1]
    _check(load().mst_softmax_rows_bwd(ptr(P), ptr(dP), P.numel() // L, L, scale, stream_of(P)), "mst_softmax_rows_bwd")
    return dP


def layernorm_rows(x: torch.Tensor, x_stride: int, rows: int, cols: int, weight, bias, eps: float) -> torch.Tensor:
    """LayerNorm of `rows` rows of `cols` fp32 values starting at x's first element, row stride x_stride -> [rows, cols] fp32."""
    out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    _check(load().mst_layernorm(ptr(x), x_stride, ptr(weight), ptr(bias), ptr(out), F32, cols, rows, cols, eps, stream_of(x)),
           "mst_layernorm")
    return out


def layernorm_bwd(x, x_stride, gamma, dy, dy_stride, dres, dres_stride, dx, dx_stride, dgamma, dbeta, rows, cols, eps):
    _check(load().mst_layernorm_bwd(ptr(x), x_stride, ptr(gamma), ptr(dy), dy_stride, ptr(dres), dres_stride, ptr(dx), dx_stride,
                                    ptr(dgamma), ptr(dbeta), rows, cols, eps, stream_of(x)), "mst_layernorm_bwd")


def act_fwd(h: torch.Tensor, kind: int) -> torch.Tensor:
    y = torch.empty_like(h)
    _check(load().mst_act_fwd(ptr(h), ptr(y), h.numel(), kind, stream_of(h)), "mst_act_fwd")
    return y


def act_bwd(h: torch.Tensor, dy: torch.Tensor, kind: int) -> torch.Tensor:
    _check(load().mst_act_bwd(ptr(h), ptr(dy), h.numel(), kind, stream_of(h)), "mst_act_bwd")
    return dy


def colsum(a: torch.Tensor, out: torch.Tensor, b: Optional[torch.Tensor] = None):
    rows, cols = a.shape
    _check(load().mst_colsum(ptr(a), cols, ptr(b), cols, rows, cols, ptr(out), stream_of(a)), "mst_colsum")
    return out


def axpby_cols(x: torch.Tensor, y: torch.Tensor, g: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 1.0,
               rows: Optional[int] = None, cols: Optional[int] = None, x_stride: Optional[int] = None, y_stride: Optional[int] = None):
    """y = alpha * x * g + beta * y over [rows, cols] (defaults: x's own 2-D shape, contiguous)."""
    cols = cols if cols is not None else x.shape[-1]
    rows = rows if rows is not None else x.numel() // cols
    _check(load().mst_axpby_cols(ptr(x), x_stride if x_stride is not None else cols, ptr(g), alpha, beta, ptr(y),
                                 y_stride if y_stride is not None else cols, rows, cols, stream_of(x)), "mst_axpby_cols")
    return y


def im2col14(vol: torch.Tensor) -> torch.Tensor:
    n, H, W = vol.shape
    col = torch.empty((n * (H // 14) * (W // 14), 196), dtype=torch.float32, device=vol.device)
    _check(load().mst_im2col14(ptr(vol), dt_of(vol), n, H, W, ptr(col), stream_of(vol)), "mst_im2col14")
    return col


def pos_embed_interp_bwd(dout: torch.Tensor, M: int, gh: int, gw: int, offset: float, dpos: torch.Tensor):
    E = dout.shape[-1]
    _check(load().mst_pos_embed_interp_bwd(ptr(dout), M, E, gh, gw, offset, ptr(dpos), stream_of(dout)), "mst_pos_embed_interp_bwd")
    return dpos


# ---- convolutional backbone ops (NHWC fp32) --------------------------------------------------------------------------
def im2col_nhwc(x: torch.Tensor, kh: int, kw: int, stride: int, pad: int, kpad: Optional[int] = None,
                out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """x [n,H,W,C] fp32 -> col [n*Ho*Wo, Kpad] (K = kh*kw*C in (ky,kx,c) order, zero-padded to Kpad), fp32 or rounded to bf16 / fp16."""
    _dev(x, "im2col_nhwc")
    n, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = kh * kw * Cc
    kpad = kpad or (K + 15) // 16 * 16
    col = torch.empty((n * Ho * Wo, kpad), dtype=out_dtype, device=x.device)
    if out_dtype == torch.float32:
        _check(load().mst_im2col_nhwc(ptr(x), n, H, W, Cc, kh, kw, stride, pad, kpad, ptr(col), stream_of(x)), "mst_im2col_nhwc")
    else:
        _check(load().mst_im2col_nhwc16(ptr(x), n, H, W, Cc, kh, kw, stride, pad, kpad, ptr(col), dt_of(col), stream_of(x)), "mst_im2col_nhwc16")
    return col


def conv_gemm(x: torch.Tensor, wg: torch.Tensor, bias: Optional[torch.Tensor], kh: int, kw: int, stride: int, pad: int, *,
              epilogue: int = EPI_BIAS, out: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Convolution of x [n,H,W,Cin] (NHWC fp32, Cin % 16 == 0) with the GEMM-form weight wg [Cout, Kpad] ((ky, kx, c) order) as an
    implicit GEMM (mst_conv_gemm: no im2col matrix) -> [n*Ho*Wo, Cout].  out: the residual operand of EPI_RESIDUAL / EPI_RESIDUAL_RELU (updated in place)."""
    _dev(x, "conv_gemm")
    n, H, W, Cin = x.shape
    Cout, kpad = wg.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((n * Ho * Wo, Cout), dtype=torch.float32, device=x.device)
    _check(load().mst_conv_gemm(ptr(x), n, H, W, Cin, kh, kw, stride, pad, ptr(wg), ptr(bias), ptr(out), Cout, kpad, epilogue, ptr(gamma),
                                stream_of(x)), "mst_conv_gemm")
    return out


def conv_gemm16(x: torch.Tensor, wg: torch.Tensor, bias: Optional[torch.Tensor], kh: int, kw: int, stride: int, pad: int, *,
                epilogue: int = EPI_BIAS, out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """mst_conv_gemm16: convolution of x [n,H,W,Cin] (NHWC bf16 / fp16, Cin % 64 == 0) with wg [Cout, kh*kw*Cin] of the same type as an
    implicit GEMM on 16-bit MFMA operands -> [n*Ho*Wo, Cout] (the operand type, or fp32).  out: the 16-bit residual operand of
    EPI_RESIDUAL_RELU (updated in place)."""
    _dev(x, "conv_gemm16")
    _dev(wg, "conv_gemm16")
    n, H, W, Cin = x.shape
    Cout = wg.shape[0]
    if wg.shape[1] != kh * kw * Cin:
        raise ValueError(f"conv_gemm16: weight [{Cout}, {wg.shape[1]}] does not match kh*kw*Cin = {kh * kw * Cin}")
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((n * Ho * Wo, Cout), dtype=out_dtype or x.dtype, device=x.device)
    _check(load().mst_conv_gemm16(ptr(x), dt_of(x), n, H, W, Cin, kh, kw, stride, pad, ptr(wg), ptr(bias), ptr(out), dt_of(out), Cout, epilogue,
                                  stream_of(x)), "mst_conv_gemm16")
    return out


def conv_dgrad(dz: torch.Tensor, wt: torch.Tensor, k: int, stride: int, pad: int, H: int, W: int) -> torch.Tensor:
    """mst_conv_dgrad: gradient of a convolution's input from dz [n,Ho,Wo,Cout] (fp32 / bf16 / fp16) and the flipped, transposed weight
    wt [Cin, k*k*Cout] (conv_dgrad_weight) -> dx [n,H,W,Cin] fp32."""
    _dev(dz, "conv_dgrad")
    _dev(wt, "conv_dgrad")
    n, Ho, Wo, Cout = dz.shape
    Cin = wt.shape[0]
    if wt.shape[1] != k * k * Cout or wt.dtype != dz.dtype:
        raise ValueError(f"conv_dgrad: weight {tuple(wt.shape)} {wt.dtype} does not match k*k*Cout = {k * k * Cout} of {dz.dtype}")
    dx = torch.empty((n, H, W, Cin), dtype=torch.float32, device=dz.device)
    _check(load().mst_conv_dgrad(ptr(dz), dt_of(dz), n, Ho, Wo, Cout, k, k, stride, pad, ptr(wt), H, W, Cin, ptr(dx), stream_of(dz)), "mst_conv_dgrad")
    return dx


def conv_wgrad(dz: torch.Tensor, x: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    """mst_conv_wgrad(16) + mst_colsum: gradient of a convolution's weight from dz [n*Ho*Wo, Cout] and the input x [n,H,W,Cin] (both fp32, or
    both bf16 / fp16; Cin % 64 == 0) -> fp32 [Cout, k*k*Cin] in (ky, kx, c) order.  The pixels are split into enough partial products to fill
    the chip."""
    _dev(dz, "conv_wgrad")
    _dev(x, "conv_wgrad")
    n, H, W, Cin = x.shape
    rows, Cout = dz.shape
    K = k * k * Cin
    lo = dz.dtype != torch.float32
    if x.dtype != dz.dtype:
        raise ValueError(f"conv_wgrad: dz is {dz.dtype}, x is {x.dtype}")
    tile, step = (128, 64) if lo else (64, 16)
    tiles = ((K + tile - 1) // tile) * ((Cout + tile - 1) // tile)
    nsplit = max(1, min(1024, 2048 // tiles, (rows + 255) // 256))
    rps = (-(-rows // nsplit) + step - 1) // step * step
    nsplit = -(-rows // rps)
    part = torch.empty((nsplit, Cout * K), dtype=torch.float32, device=dz.device)
    if lo:
        _check(load().mst_conv_wgrad16(ptr(dz), ptr(x), dt_of(dz), n, H, W, Cin, k, k, stride, pad, Cout, ptr(part), nsplit, rps, stream_of(dz)),
               "mst_conv_wgrad16")
    else:
        _check(load().mst_conv_wgrad(ptr(dz), ptr(x), n, H, W, Cin, k, k, stride, pad, Cout, ptr(part), nsplit, rps, stream_of(dz)), "mst_conv_wgrad")
    if nsplit == 1:
        return part.view(Cout, K)
    return colsum(part, torch.zeros(Cout * K, dtype=torch.float32, device=dz.device)).view(Cout, K)


def conv_dgrad_weight(weight: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """[Cout, Cin, k, k] -> the operand mst_conv_dgrad multiplies by: [Cin, (ky', kx', co)] with both kernel axes flipped."""
    Cout, Cin, k, _ = weight.shape
    return weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, k * k * Cout).to(dtype).contiguous()


def cvt32(x: torch.Tensor) -> torch.Tensor:
    """mst_cvt32: fp32 copy of a 16-bit tensor."""
    _dev(x, "cvt32")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _check(load().mst_cvt32(ptr(x), dt_of(x), x.numel(), ptr(out), stream_of(x)), "mst_cvt32")
    return out


def maxpool_nhwc(x: torch.Tensor) -> torch.Tensor:
    _dev(x, "maxpool_nhwc")
    n, H, W, Cc = x.shape
    y = torch.empty((n, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=x.dtype, device=x.device)
    if x.dtype == torch.float32:
        _check(load().mst_maxpool_nhwc(ptr(x), n, H, W, Cc, ptr(y), stream_of(x)), "mst_maxpool_nhwc")
    else:
        _check(load().mst_maxpool_nhwc16(ptr(x), dt_of(x), n, H, W, Cc, ptr(y), stream_of(x)), "mst_maxpool_nhwc16")
    return y


def avgpool_nhwc(x: torch.Tensor) -> torch.Tensor:
    _dev(x, "avgpool_nhwc")
    n, H, W, Cc = x.shape
    y = torch.empty((n, Cc), dtype=torch.float32, device=x.device)
    _check(load().mst_avgpool_nhwc(ptr(x), n, H * W, Cc, ptr(y), stream_of(x)), "mst_avgpool_nhwc")
    return y


def batchnorm_train(z: torch.Tensor, bn, residual: Optional[torch.Tensor], relu: bool, momentum: float = 0.1):
    """nn.BatchNorm2d in train mode on z [rows, C] (+ residual, ReLU): returns (y, mean, rstd); updates bn.running_* in place."""
    _dev(z, "batchnorm_train")
    rows, Cc = z.shape
    dev = z.device
    y = torch.empty_like(z)
    mean = torch.empty(Cc, dtype=torch.float32, device=dev)
    rstd = torch.empty(Cc, dtype=torch.float32, device=dev)
    scratch = torch.empty(Cc, dtype=torch.float32, device=dev)
    _check(load().mst_batchnorm_train(ptr(z), rows, Cc, ptr(bn.weight.detach()), ptr(bn.bias.detach()), bn.eps, momentum, ptr(residual),
                                      1 if relu else 0, ptr(y), ptr(mean), ptr(rstd), ptr(bn.running_mean), ptr(bn.running_var),
                                      ptr(scratch), stream_of(z)), "mst_batchnorm_train")
    return y, mean, rstd


def batchnorm_bwd(z: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor):
    """Returns (dz, dgamma, dbeta)."""
    rows, Cc = z.shape
    dg = torch.zeros(Cc, dtype=torch.float32, device=z.device)
    db = torch.zeros(Cc, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    _check(load().mst_batchnorm_bwd(ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dy), rows, Cc, ptr(dg), ptr(db), ptr(dz),
                                    stream_of(z)), "mst_batchnorm_bwd")
    return dz, dg, db


def col2im_nhwc(dcol: torch.Tensor, dx: torch.Tensor, kh: int, kw: int, stride: int, pad: int):
    """dx [n,H,W,C] += adjoint of im2col_nhwc applied to dcol [n*Ho*Wo, Kpad]."""
    n, H, W, Cc = dx.shape
    _check(load().mst_col2im_nhwc(ptr(dcol), n, H, W, Cc, kh, kw, stride, pad, dcol.shape[1], ptr(dx), stream_of(dx)), "mst_col2im_nhwc")
    return dx


def maxpool_bwd_nhwc(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    n, H, W, Cc = x.shape
    dx = torch.zeros_like(x)
    _check(load().mst_maxpool_bwd_nhwc(ptr(x), ptr(dy), n, H, W, Cc, ptr(dx), stream_of(x)), "mst_maxpool_bwd_nhwc")
    return dx


def avgpool_bwd_nhwc(dy: torch.Tensor, HW: int) -> torch.Tensor:
    n, Cc = dy.shape
    dx = torch.empty((n, HW, Cc), dtype=torch.float32, device=dy.device)
    _check(load().mst_avgpool_bwd_nhwc(ptr(dy), n, HW, Cc, ptr(dx), stream_of(dy)), "mst_avgpool_bwd_nhwc")
    return dx


def gradcampp(act: torch.Tensor, out: torch.Tensor, fc_weight: Optional[torch.Tensor]) -> torch.Tensor:
    """Grad-CAM++ map of the last ReLU output: act [n, HW, C], out [n, O] model output, fc_weight [O, C] or None -> [n, HW]."""
    n, HW, Cc = act.shape
    cam = torch.empty((n, HW), dtype=torch.float32, device=act.device)
    state = torch.empty(2, dtype=torch.float32, device=act.device)
    _check(load().mst_gradcampp(ptr(act), ptr(out), out.shape[1], ptr(fc_weight), n, HW, Cc, ptr(cam), ptr(state), stream_of(act)),
           "mst_gradcampp")
    return cam


class Profiler:
    """Caller-owned per-kernel timer (mst_profiler): attach with ``vit.profiler = prof.handle`` (DinoV2ClassifierSlice.profiler)."""

    def __init__(self):
        self.handle = load().mst_profiler_create()
        if not self.handle:
            raise RuntimeError("mst_profiler_create failed")

    def collect(self):
        """{kind name: (total ms, launches)} since the last collect (waits for the recorded events)."""
        ms = (C.c_double * K_COUNT)()
        cnt = (C.c_int64 * K_COUNT)()
        _check(load().mst_profiler_collect(self.handle, ms, cnt), "mst_profiler_collect")
        lib = load()
        return {lib.mst_kernel_kind_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(K_COUNT)}

    def close(self):
        if self.handle:
            load().mst_profiler_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
