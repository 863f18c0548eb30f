"""CPU oracle for the MST-DINOv2 hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional fp32 (or fp64) PyTorch-CPU restatement of ``DinoV2ClassifierSlice.forward`` and its
attention read-outs, written over a plain ``state_dict``.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this file; the product package
(``new-vit_amd/mst``) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tools/gen_golden.py`` ran the reference's own modules (imported from
/root/reference in the build container) on the synthetic weights of ``mst.synth`` and wrote the
fixtures in ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below against
them (the reference's own tests hold no numerical vectors: SURVEY.md section 4).

Exception -- ``fp8_e4m3`` / ``fp8_linear`` (``linear="fp8"``): PARITY UNPINNED.  The reference has no fp8 code; these restate the
arithmetic BASELINE.json configs[4] names (OCP e4m3 operands, per-tensor absmax scales) on top of the pinned fp32 functions and
are only anchored by torch's own ``float8_e4m3fn`` cast.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

VIT_CFG = {  # mst/models/extern/dinov2/vision_transformer.py:340-365
    "s": dict(embed_dim=384, depth=12, num_heads=6),
    "b": dict(embed_dim=768, depth=12, num_heads=12),
}
SLICE_HEADS = 12  # mst/models/dino.py:87
PATCH = 14        # mst/models/dino.py:66


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _block_prefix(sd: SD, i: int) -> str:
    """Both key layouts: chunked ``blocks.0.i`` (vision_transformer.py:153-160) and hub ``blocks.i``."""
    p = f"encoder.blocks.0.{i}"
    return p if (p + ".norm1.weight") in sd else f"encoder.blocks.{i}"


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim (block.py:63,75; vision_transformer.py:95,165)."""
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (mlp.py:22,30)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------------------------
# per-slice encoder (DinoVisionTransformer)
# --------------------------------------------------------------------------------------------
def patch_embed(slices: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """Gray->RGB repeat (dino.py:125-127) + Conv2d(3,E,14,14) + flatten/transpose
    (patch_embed.py:65,68-81) as one GEMM.  ``slices`` is [n,H,W]; returns [n,Np,E]."""
    n, H, W = slices.shape
    assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
    assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
    gh, gw = H // PATCH, W // PATCH
    cols = slices.reshape(n, gh, PATCH, gw, PATCH).permute(0, 1, 3, 2, 4).reshape(n, gh * gw, PATCH * PATCH)
    # the three input channels are identical copies, so summing products over c equals the conv
    wmat = w.reshape(w.shape[0], 3, PATCH * PATCH)
    out = cols @ wmat[:, 0].t() + cols @ wmat[:, 1].t() + cols @ wmat[:, 2].t()
    return out + b


def interpolate_pos_encoding(pos_embed: Tensor, npatch: int, w: int, h: int,
                             offset: float = 0.1, antialias: bool = False) -> Tensor:
    """vision_transformer.py:179-211: bicubic resampling of the stored patch grid.  offset != 0: scale_factor =
    (g + offset) / M (l.194-199, the vendored default 0.1); offset == 0: output size given (l.200-202).  `antialias` is
    interpolate_antialias (l.206).  Note the reference passes (w, h) = (x.shape[2], x.shape[3]) = (H, W): l.214,220."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    pe = pos_embed.float()
    cls_pe, patch_pe = pe[:, 0], pe[:, 1:]
    dim = pe.shape[-1]
    w0, h0 = w // PATCH, h // PATCH
    M = int(math.sqrt(N))
    assert N == M * M
    kw = dict(scale_factor=(float(w0 + offset) / M, float(h0 + offset) / M)) if offset else dict(size=(w0, h0))
    grid = F.interpolate(patch_pe.reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=antialias, **kw)
    assert (w0, h0) == tuple(grid.shape[-2:])
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), grid), dim=1).to(pos_embed.dtype)


def prepare_tokens(sd: SD, slices: Tensor) -> Tensor:
    """vision_transformer.py:213-232 (no masks): patch-embed, CLS, +pos, registers after CLS."""
    n, H, W = slices.shape
    x = patch_embed(slices, sd["encoder.patch_embed.proj.weight"], sd["encoder.patch_embed.proj.bias"])
    x = torch.cat((sd["encoder.cls_token"].expand(n, -1, -1), x), dim=1)
    reg = sd.get("encoder.register_tokens")
    # Registers only exist in the hub's `dinov2_vit*14_reg` models (dino.py:61), which facebookresearch/dinov2
    # hub/backbones.py (un-vendored: fetched by torch.hub) builds with interpolate_antialias=True, interpolate_offset=0.0;
    # every other encoder keeps the vendored defaults (vision_transformer.py:66-67: no antialias, offset 0.1).
    x = x + interpolate_pos_encoding(sd["encoder.pos_embed"], x.shape[1] - 1, H, W,
                                     offset=0.0 if reg is not None else 0.1, antialias=reg is not None)
    if reg is not None:
        x = torch.cat((x[:, :1], reg.expand(n, -1, -1), x[:, 1:]), dim=1)
    return x


F8_MAX = 448.0   # largest finite OCP e4m3


def fp8_e4m3(x: Tensor, amax: Optional[float] = None) -> Tuple[Tensor, float]:
    """Per-tensor e4m3 quantisation: (values representable in e4m3 as fp32, scale = max|x|/448), x ~ scale * values.
    The multiplier 448/max|x| is formed and applied in fp32, rounding to e4m3 is to nearest-even.  `amax` given: a calibrated
    (static) scale replaces max|x|; larger values saturate at +-448 quanta."""
    amax = x.detach().abs().max().to(torch.float32) if amax is None else torch.tensor(float(amax), dtype=torch.float32)
    if float(amax) == 0.0:
        return torch.zeros_like(x, dtype=torch.float32), 1.0
    inv = torch.tensor(F8_MAX, dtype=torch.float32) / amax
    q = (x.to(torch.float32) * inv).clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return q, float(amax) / F8_MAX


def fp8_linear(x: Tensor, w: Tensor, b: Optional[Tensor], amax: Optional[float] = None) -> Tensor:
    """F.linear with OCP-e4m3 operands, per-tensor absmax scales and wide accumulation -- the arithmetic BASELINE.json
    configs[4] / SURVEY.md 8d row c5 name ("fp8-e4m3 operands, per-tensor scales from absmax").  PARITY UNPINNED: the
    reference has no fp8 code; this restates F.linear (attention.py:58,67; mlp.py:35,38) with both operands rounded."""
    xq, sx = fp8_e4m3(x, amax)
    wq, sw = fp8_e4m3(w)
    y = (xq.double() @ wq.double().t()) * (sx * sw)
    return (y + (b.double() if b is not None else 0.0)).to(x.dtype)


def _linear_fn(linear: str, amax, j: int):
    """F.linear, or fp8_linear with the j-th calibrated activation scale of the block (None = dynamic)."""
    if linear == "exact":
        return F.linear
    assert linear == "fp8", linear
    a = None if amax is None else float(amax[j])
    return lambda x, w, b: fp8_linear(x, w, b, a)


def vit_attention(x: Tensor, sd: SD, p: str, heads: int, linear: str = "exact", amax=None) -> Tuple[Tensor, Tensor]:
    """attention.py:56-69 (== dino.py:226-243 with the softmax kept).  Returns (out, probs)."""
    n, N, C = x.shape
    d = C // heads
    qkv = _linear_fn(linear, amax, 0)(x, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    qkv = qkv.reshape(n, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]
    probs = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    out = (probs @ v).transpose(1, 2).reshape(n, N, C)
    return _linear_fn(linear, amax, 1)(out, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"]), probs


def vit_block(x: Tensor, sd: SD, p: str, heads: int, linear: str = "exact", amax=None) -> Tuple[Tensor, Tensor]:
    """block.py:89-114 eval branch: x += ls1(attn(norm1 x)); x += ls2(mlp(norm2 x)); LN eps 1e-6.
    amax: 4 calibrated activation scales (inputs of qkv, proj, fc1, fc2) for linear='fp8', or None (dynamic)."""
    a, probs = vit_attention(layer_norm(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6), sd, p, heads, linear, amax)
    if (p + ".ls1.gamma") in sd:  # layer_scale.py:26-27
        a = a * sd[p + ".ls1.gamma"]
    x = x + a
    h = layer_norm(x, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    h = _linear_fn(linear, amax, 2)(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])           # mlp.py:34-40
    h = _linear_fn(linear, amax, 3)(gelu_erf(h), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    if (p + ".ls2.gamma") in sd:
        h = h * sd[p + ".ls2.gamma"]
    return x + h, probs


def vit_encode(sd: SD, slices: Tensor, model_size: str = "s", keep: str = "none", linear: str = "exact", act_amax=None):
    """DinoVisionTransformer.forward (vision_transformer.py:254-270,324-329) -> normalised CLS [n,E].

    keep: 'none' | 'cls' (CLS row of every block's softmax, [n,h,1,N] each) | 'full' ([n,h,N,N]).
    linear: 'exact' (the reference) | 'fp8' (the blocks' four linear layers through fp8_linear; everything else unchanged).
    act_amax: [depth][4] calibrated activation scales for 'fp8' (None = per-call absmax)."""
    cfg = VIT_CFG[model_size]
    x = prepare_tokens(sd, slices)
    maps: List[Tensor] = []
    for i in range(cfg["depth"]):
        x, probs = vit_block(x, sd, _block_prefix(sd, i), cfg["num_heads"], linear, None if act_amax is None else act_amax[i])
        if keep == "cls":
            maps.append(probs[:, :, :1].clone())
        elif keep == "full":
            maps.append(probs)
    x = layer_norm(x, sd["encoder.norm.weight"], sd["encoder.norm.bias"], 1e-6)
    return x[:, 0], maps


# --------------------------------------------------------------------------------------------
# across-slice transformer
# --------------------------------------------------------------------------------------------
def rope_rotate(t: Tensor, freqs_param: Tensor) -> Tensor:
    """RotaryEmbedding.rotate_queries_or_keys on [B,h,L,hd] (rotary_embedding_torch.py:159-173,
    45-62, 273-302): positions 0..L-1, interleaved pairs (x0,x1)->(-x1,x0)."""
    L = t.shape[-2]
    ang = torch.arange(L, dtype=t.dtype)[:, None] * freqs_param.to(t.dtype)[None, :]
    ang = ang.repeat_interleave(2, dim=-1)                       # [L, hd]
    tp = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-tp[..., 1], tp[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def liere_matrix(vars_: List[Tensor]) -> Tensor:
    """AttentionLiereRotator's position-independent rotation (rotary_embedding_torch.py:319-326, 357-371):
    per block, skew generator A[i,j] = sum_p p * x[tril(i,j), p] (i > j), R_blk = matrix_exp(A) in fp32,
    R = block_diag(R_0, R_1, ...) of size [hd, hd]."""
    mats = []
    for v in vars_:
        n = int(round((1 + math.sqrt(1 + 8 * v.shape[0])) / 2))            # block size from n(n-1)/2 rows
        pos = torch.arange(v.shape[1], dtype=torch.float32)
        flat = v[:, :, 0].float() @ pos                                     # [n(n-1)/2]
        i, j = torch.tril_indices(n, n, offset=-1)
        A = torch.zeros(n, n)
        A[i, j] = flat
        A[j, i] = -flat
        mats.append(torch.linalg.matrix_exp(A))
    return torch.block_diag(*mats)


def liere_rotate(t: Tensor, R: Tensor, axes_length: int = 33) -> Tensor:
    """AttentionLiereRotator.rotate_queries_or_keys on [B,h,L,hd] + the caller's reshape
    (rotary_embedding_torch.py:388-396, 347-386; transformer_blocks.py:263-264).  Every token/head vector is
    multiplied by the same R; the result is left in [B, L, h, hd] memory order and then VIEWED as
    [B*h, L, hd] by the caller, which re-partitions (token, head) pairs into pseudo-heads.  Like the
    reference this only works for L == 33 and B == 1 (the views raise RuntimeError otherwise)."""
    B, h, L, hd = t.shape
    x = t.permute(0, 2, 1, 3).contiguous()                                  # l.390
    x = x.view(B, axes_length, h, hd)                                       # l.349 (RuntimeError unless L == 33)
    x = x.permute(0, 3, 1, 2)                                               # l.375
    x = torch.bmm(R[None].repeat(B, 1, 1), x.reshape(B, hd, axes_length * h))   # l.380
    x = x.view(B, hd, axes_length, h).permute(0, 2, 3, 1)                   # l.381 (a view, not a copy)
    return x.view(B * h, L, hd).view(B, h, L, hd)                           # transformer_blocks.py:263 (RuntimeError unless B == 1)


def slice_fusion(sd: SD, x: Tensor, key_padding_mask: Optional[Tensor] = None,
                 rotary: Optional[str] = None, heads: int = SLICE_HEADS) -> Tuple[Tensor, Tensor]:
    """nn.TransformerEncoder(1 x TransformerEncoderLayer(norm_first), norm=LayerNorm)
    (dino.py:84-96; transformer_blocks.py:565-587,29-318).  x = [B,L,E] with CLS at 0;
    key_padding_mask bool [B,L], True = ignore.  Returns (y [B,L,E], probs [B,12,L,L])."""
    p = "slice_fusion.layers.0"
    B, L, E = x.shape
    h, hd = heads, E // heads
    y = layer_norm(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    qkv = F.linear(y, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
    q, k, v = (t.reshape(B, L, h, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))  # [B,h,L,hd]
    if rotary == "RoPE":                                                             # l.262-264
        fr = sd[p + ".self_attn.rotary_positional_encoding.freqs"]
        q, k = rope_rotate(q, fr), rope_rotate(k, fr)
    elif rotary == "LiRE":
        pre = p + ".self_attn.rotary_positional_encoding.vars."
        R = liere_matrix([sd[pre + str(i)] for i in range(sum(1 for k in sd if k.startswith(pre)))])
        q, k = liere_rotate(q, R), liere_rotate(k, R)
    elif rotary is not None:
        raise NotImplementedError(rotary)
    s = (q * math.sqrt(1.0 / hd)) @ k.transpose(-2, -1)                              # l.268-275
    if key_padding_mask is not None:                                                 # l.244-252
        s = s + torch.zeros(B, 1, 1, L, dtype=s.dtype).masked_fill_(key_padding_mask[:, None, None, :], float("-inf"))
    probs = s.softmax(dim=-1)
    a = (probs @ v).transpose(1, 2).reshape(B, L, E)
    a = F.linear(a, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    x = x + a
    f = layer_norm(x, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    f = F.linear(torch.relu(F.linear(f, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])                # l.585-587, relu l.484
    x = x + f
    return layer_norm(x, sd["slice_fusion.norm.weight"], sd["slice_fusion.norm.bias"], 1e-5), probs


# --------------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------------
def forward(sd: SD, source: Tensor, *, model_size: str = "s", slice_fusion_type: str = "transformer",
            src_key_padding_mask: Optional[Tensor] = None, rotary: Optional[str] = None,
            without_linear: bool = False, keep: str = "none", linear: str = "exact") -> Dict[str, Tensor]:
    """DinoV2ClassifierSlice.forward (dino.py:110-167).  source = [B,C,D,H,W]; channels become extra slices, channel
    fastest ('b c d h w -> (b d c) h w', l.125).

    Returns dict(logits|features, emb [B*D,E], vit_maps list, slice_map [B,12,L,L] or None)."""
    B, C, D, H, W = source.shape
    emb, maps = vit_encode(sd, source.permute(0, 2, 1, 3, 4).reshape(B * D * C, H, W), model_size, keep, linear)   # l.125-131
    out = fuse(sd, emb, B, D * C, slice_fusion_type=slice_fusion_type, src_key_padding_mask=src_key_padding_mask,
               rotary=rotary, without_linear=without_linear)
    out["vit_maps"] = maps
    return out


def fuse(sd: SD, emb: Tensor, B: int, D: int, *, slice_fusion_type: str = "transformer",
         src_key_padding_mask: Optional[Tensor] = None, rotary: Optional[str] = None,
         without_linear: bool = False) -> Dict[str, Tensor]:
    """Everything of DinoV2ClassifierSlice.forward after the encoder (dino.py:134-166) on slice embeddings emb [B*D, E]."""
    x = emb
    if "bottleneck.weight" in sd:                                                    # l.134-135
        x = F.linear(x, sd["bottleneck.weight"], sd["bottleneck.bias"])
    x = x.reshape(B, D, -1)                                                          # l.138
    if "slice_pos_emb.weight" in sd:                                                 # l.140-142
        x = x + sd["slice_pos_emb.weight"][: x.shape[1]]
    slice_map = None
    if slice_fusion_type == "transformer":                                           # l.144-153
        x = torch.cat([sd["cls_token"].repeat(B, 1, 1), x], dim=1)
        m = None
        if src_key_padding_mask is not None:
            m = torch.cat([torch.zeros(B, 1, dtype=torch.bool), src_key_padding_mask.bool()], dim=1)
        x, slice_map = slice_fusion(sd, x, m, rotary)
        x = x[:, 0]
    elif slice_fusion_type == "linear":
        x = x.reshape(B, -1)
    elif slice_fusion_type == "average":
        x = x.mean(dim=1)
    out = {"emb": emb, "slice_map": slice_map, "features": x}
    if not without_linear and "linear.weight" in sd:                                 # l.164-166
        out["logits"] = F.linear(x, sd["linear.weight"], sd["linear.bias"])
    return out


def plane_attention(last_vit_map: Tensor, num_registers: int = 0) -> Tensor:
    """get_plane_attention (dino.py:189-195): CLS row of the last block, patch columns, first patch
    zeroed, renormalised.  Accepts [n,h,N,N] or the CLS-only [n,h,1,N]."""
    a = last_vit_map[:, :, 0, 1 + (4 if num_registers else 0):].clone()
    a[:, :, 0] = 0
    return a / a.sum(dim=-1, keepdim=True)


def slice_attention(slice_map: Tensor) -> Tensor:
    """get_slice_attention (dino.py:173-187): CLS row, slices only, renormalised, head-mean -> [B*D,1,1]."""
    a = slice_map[:, :, 0, 1:].clone()
    a = a / a.sum(dim=-1, keepdim=True)
    return a.mean(dim=1).reshape(-1)[:, None, None]


def attention_maps(last_vit_map: Tensor, slice_map: Tensor, num_registers: int = 0) -> Tensor:
    """get_attention_maps (dino.py:197-202)."""
    return slice_attention(slice_map) * plane_attention(last_vit_map, num_registers)


def attention_rollout(vit_maps_full: List[Tensor]) -> Tensor:
    """get_attention_cls (dino.py:204-212): A_0 @ A_1 @ ... @ A_last on full [n,h,N,N] maps."""
    acc = vit_maps_full[-1]
    for a in reversed(vit_maps_full[:-1]):
        acc = torch.matmul(a, acc)
    return acc


TTA_FLIPS = [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]   # scripts/main_predict.py:148


def trilinear_upsample(w: Tensor, size: Tuple[int, int, int]) -> Tensor:
    """F.interpolate(w, size, mode='trilinear') with align_corners=False (scripts/main_predict.py:163), restated as
    three separable 1-D linear resamplings of [1,1,D,h,w]: src = max((dst + 0.5) * in/out - 0.5, 0), i0 = floor(src),
    i1 = min(i0 + 1, in - 1), value = (1 - l) * v[i0] + l * v[i1] (torch UpSampleKernel area_pixel_compute_source_index)."""
    out = w
    for axis, n_out in zip((2, 3, 4), size):
        n_in = out.shape[axis]
        dst = torch.arange(n_out, dtype=torch.float32)
        src = ((dst + 0.5) * (n_in / n_out) - 0.5).clamp_(min=0)
        i0 = src.floor().long().clamp_(max=n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        l1 = (src - i0.float())
        shape = [1] * 5
        shape[axis] = n_out
        l1 = l1.reshape(shape)
        out = (1 - l1) * out.index_select(axis, i0) + l1 * out.index_select(axis, i1)
    return out


def saliency_lowres(maps: Tensor, slice_attn: Tensor, D: int) -> Tuple[Tensor, Tensor]:
    """_pred_trans of scripts/main_predict.py:72-105 after the forward (DinoV2 branch, B = 1): head mean of
    get_attention_maps() -> [1,1,D,g,g] (square grid assumed there: l.90-96) and get_slice_attention() head-mean -> [D]."""
    w = maps.mean(dim=1)                                    # l.75-76
    g = int(w.shape[-1] ** 0.5)                             # l.91
    w = w[:, :g * g].reshape(1, 1, D, g, g)                 # l.94-98
    ws = slice_attn.mean(dim=1).reshape(D)                  # l.101-102
    return w, ws


def run_pred(sd: SD, source: Tensor, *, use_softmax: bool = True, use_tta: bool = False,
             src_key_padding_mask: Optional[Tensor] = None, **fw) -> Tuple[Tensor, Tensor, Tensor]:
    """run_pred(save_attn=True) of scripts/main_predict.py:134-165 for a DinoV2ClassifierSlice on one volume:
    returns (pred [1,out], weight [1,1,D,H,W], weight_slice [1,1,D,H,W])."""
    def one(src):
        o = forward(sd, src, src_key_padding_mask=src_key_padding_mask, keep="cls", **fw)
        pred = o["logits"].softmax(-1) if use_softmax else o["logits"]
        D = src.shape[2]
        w, ws = saliency_lowres(attention_maps(o["vit_maps"][-1], o["slice_map"]), slice_attention(o["slice_map"]), D)
        return pred, w, ws.reshape(1, 1, D, 1, 1) * torch.ones_like(src)
    pred, w, ws = one(source)
    if use_tta:
        for dims in TTA_FLIPS:
            p_i, w_i, ws_i = one(torch.flip(source, dims))
            pred = pred + p_i
            w = w + torch.flip(w_i, dims)
            ws = ws + torch.flip(ws_i, dims)
        pred, w, ws = pred / 8, w / 8, ws / 8
    return pred, trilinear_upsample(w, tuple(source.shape[2:])), ws


def slices2rgb(tensor: Tensor) -> Tensor:
    """dino.py:10-27: [B,1,D,H,W] -> [B*ceil(D/3), 3, H, W], padded along D with the first slices."""
    B, C, D, H, W = tensor.shape
    assert C == 1, "More than one channel"
    if D % 3 != 0:
        tensor = torch.cat([tensor, tensor[:, :, : 3 - D % 3]], dim=2)
    return tensor.reshape(B, tensor.shape[2] // 3, 3, H, W).reshape(-1, 3, H, W)


def crop_or_pad(x: Tensor, target_shape, padding_mode=0) -> Tensor:
    """CropOrPad (augmentations_3d.py:144-195) with random_center=False on [C, a0, a1, a2]: bounds ini = ceil(n/2), fin = n - ini
    (l.164-172); torchio 0.19.9's Pad is numpy.pad (mode 'minimum' or constant) on the spatial axes, then Crop."""
    import numpy as np
    a = x.numpy()
    src = a.shape[1:]
    tgt = [s if t is None else int(t) for t, s in zip(target_shape, src)]
    diff = [t - s for t, s in zip(tgt, src)]
    pads = [(0, 0)] + [((max(d, 0) + 1) // 2, max(d, 0) - (max(d, 0) + 1) // 2) for d in diff]
    if any(p != (0, 0) for p in pads):
        a = np.pad(a, pads, mode="minimum") if padding_mode == "minimum" else np.pad(a, pads, mode="constant", constant_values=padding_mode)
    crops = [((max(-d, 0) + 1) // 2, max(-d, 0) - (max(-d, 0) + 1) // 2) for d in diff]
    sl = (slice(None),) + tuple(slice(c0, a.shape[i + 1] - c1) for i, (c0, c1) in enumerate(crops))
    return torch.from_numpy(np.ascontiguousarray(a[sl]))


def znormalize(x: Tensor, percentiles=(0, 100)) -> Tensor:
    """ZNormalization._znorm (augmentations_3d.py:73-86) for one channel with the datasets' masking method
    (x > x.min()) & (x < x.max()) (dataset_3d_duke.py:43) + torchio's ZNormalization.znorm (mean / unbiased std of the masked values)."""
    mask = (x > x.min()) & (x < x.max())
    cutoff = torch.quantile(x.masked_select(mask).float(), torch.tensor(percentiles, dtype=torch.float32) / 100.0)
    y = torch.clamp(x, *cutoff.to(x.dtype).tolist())
    y = y.clone().float()
    values = y[mask]
    mean, std = values.mean(), values.std()
    if std == 0:
        raise RuntimeError("Standard deviation is 0 for masked values")
    y -= mean
    y /= std
    return y


def flops_per_volume(D: int, H: int, W: int, E: int = 384, depth: int = 12) -> float:
    """Algorithmic FLOPs of one forward: 2 FLOP per MAC, softmax/LN/GELU not counted."""
    Np = (H // PATCH) * (W // PATCH)
    N = Np + 1
    f_slice = 2.0 * Np * E * 588 + depth * (24.0 * N * E * E + 4.0 * N * N * E)
    L = D + 1
    f_fusion = 2.0 * L * 3 * E * E + 4.0 * L * L * E + 2.0 * L * E * E + 4.0 * L * E * E + 4.0 * E
    return D * f_slice + f_fusion
