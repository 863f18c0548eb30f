"""MI355X-native ``DinoV2ClassifierSlice`` -- drop-in for the reference's mst/models/dino.py:32-275.

Same constructor, ``forward(source, save_attn, src_key_padding_mask, **kwargs)``, attention getters
and ``state_dict`` keys (both the ``pretrained=False`` chunked layout ``encoder.blocks.0.<i>.*`` and
the torch.hub layout ``encoder.blocks.<i>.*`` with ``ls{1,2}.gamma`` load).  The modules below are
parameter containers only: all arithmetic of the forward runs in the hand-written HIP kernels of
libmst_hip.so through ``mst.hip`` (C ABI: include/mst_hip.h).  There is no PyTorch/CPU fallback.

Build-specific (keyword-only, all optional) controls -- none changes the maths of the reference:
  compute_dtype   'bf16' (default) | 'fp16' | 'fp32' | 'fp8'  MFMA operand type of the encoder GEMMs and
                  attention (fp32 accumulate / residual / LayerNorm / softmax always). fp16 is the
                  TF32-class path (reference GPUs run TF32: main_predict.py:195), fp32 is exact.
                  'fp8' (BASELINE configs[4]): the blocks' four linear layers with OCP e4m3 operands and
                  per-tensor absmax scales (dynamic for activations), everything else as in bf16 mode.
  train_precision 'fp32' (default) | 'bf16' | 'fp16': MFMA operand type of the blocks' nn.Linear products in the TRAINING step
                  (forward and backward; fp32 accumulation, fp32 storage and every other op fp32): the reference trains under
                  Trainer(precision='16-mixed') (scripts/main_train.py:110-123).  fp32 is the exact mode the gradient parity bar is on.
  chunk_slices    slices encoded per pass (activations of a pass sized for the Infinity Cache).
  full_attention_maps  keep the complete [n,h,N,N] softmax of every block on ``save_attn`` (needed
                  only by ``get_attention_cls``); default keeps the CLS rows ([n,h,1,N]) only.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import List, Optional

import torch
import torch.nn as nn

from .base_model import BasicClassifier
from .. import hip
from ..parallel import SliceSharding

PATCH = 14
_VIT_SIZES = {  # reference: extern/dinov2/vision_transformer.py:340-395
    "s": dict(embed_dim=384, depth=12, num_heads=6),
    "b": dict(embed_dim=768, depth=12, num_heads=12),
    "l": dict(embed_dim=1024, depth=24, num_heads=16),
    "g": dict(embed_dim=1536, depth=40, num_heads=24),
}
SLICE_HEADS = 12  # reference dino.py:87


def slices2rgb(tensor):
    """[B, 1, D, H, W] -> [B*ceil(D/3), 3, H, W]: three consecutive slices as the channels of one image, D padded with the
    volume's own first slices (reference dino.py:10-27; dead code there -- its call at dino.py:129 is commented out --, kept
    for the import surface).  Runs on the device (mst_slices2rgb)."""
    return hip.slices2rgb(tensor.contiguous())


# ------------------------------------------------------------------------------------------------
# parameter containers (names = the reference's state_dict keys)
# ------------------------------------------------------------------------------------------------
class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward runs in libmst_hip.so (see DinoV2ClassifierSlice.forward)")


class _Affine(_Params):  # LayerNorm / Linear holder
    def __init__(self, *wshape, bias_shape=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        self.bias = nn.Parameter(torch.zeros(bias_shape if bias_shape is not None else wshape[0]))


def _ln(dim):
    m = _Affine(dim)
    nn.init.ones_(m.weight)
    return m


def _lin(out_f, in_f, std=None):
    m = _Affine(out_f, in_f)
    if std is None:  # nn.Linear default
        nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_f)
        nn.init.uniform_(m.bias, -bound, bound)
    else:            # init_weights_vit_timm (vision_transformer.py:332-337)
        nn.init.trunc_normal_(m.weight, std=std)
    return m


class _Gamma(_Params):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))


class _Attn(_Params):
    def __init__(self, E):
        super().__init__()
        self.qkv = _lin(3 * E, E, std=0.02)
        self.proj = _lin(E, E, std=0.02)


class _Mlp(_Params):
    def __init__(self, E):
        super().__init__()
        self.fc1 = _lin(4 * E, E, std=0.02)
        self.fc2 = _lin(E, 4 * E, std=0.02)


class _Block(_Params):  # block.py:42-87
    def __init__(self, E, layerscale: Optional[float]):
        super().__init__()
        self.norm1 = _ln(E)
        self.attn = _Attn(E)
        self.norm2 = _ln(E)
        self.mlp = _Mlp(E)
        if layerscale:
            self.ls1 = _Gamma(E, layerscale)
            self.ls2 = _Gamma(E, layerscale)


class _PatchEmbed(_Params):
    def __init__(self, E):
        super().__init__()
        self.proj = _Affine(E, 3, PATCH, PATCH, bias_shape=E)
        nn.init.kaiming_uniform_(self.proj.weight, a=math.sqrt(5))


class _ViT(_Params):
    """Parameter tree of DinoVisionTransformer (vision_transformer.py:44-170)."""

    def __init__(self, embed_dim, depth, num_heads, img_size=224, num_register_tokens=0,
                 layerscale: Optional[float] = None, chunked=True):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim = self.num_features = embed_dim
        self.depth, self.num_heads = depth, num_heads
        self.num_register_tokens = num_register_tokens
        self.patch_size = PATCH
        g = img_size // PATCH
        self.patch_embed = _PatchEmbed(embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, g * g + 1, embed_dim))
        if num_register_tokens:
            self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, embed_dim))
        blocks = [_Block(embed_dim, layerscale) for _ in range(depth)]
        self.chunked = chunked
        self.blocks = nn.ModuleList([nn.ModuleList(blocks)]) if chunked else nn.ModuleList(blocks)
        self.norm = _ln(embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        if num_register_tokens:
            nn.init.normal_(self.register_tokens, std=1e-6)

    def block_list(self) -> List[_Block]:
        return list(self.blocks[0]) if self.chunked else list(self.blocks)


class _SelfAttn(_Params):
    def __init__(self, E, rotary):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = _lin(E, E)
        assert E % SLICE_HEADS == 0, "embed_dim must be divisible by num_heads"   # nn.MultiheadAttention
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)
        if rotary == "RoPE":  # transformer_blocks.py:333-349 (theta 256, 'lang' freqs)
            hd = E // SLICE_HEADS
            holder = _Params()
            holder.freqs = nn.Parameter(1.0 / (256 ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd)),
                                        requires_grad=False)
            self.rotary_positional_encoding = holder
        elif rotary == "LiRE":  # transformer_blocks.py:350-357; rotary_embedding_torch.py:337-339
            hd = E // SLICE_HEADS
            blk = hd // 2
            holder = _Params()
            holder.vars = nn.ParameterList([nn.Parameter(torch.randn((blk * blk - blk) // 2, 33, 1))
                                            for _ in range(hd // blk)])
            self.rotary_positional_encoding = holder
        elif rotary is not None:
            raise ValueError(f"Unkown parameter {rotary} for rotary_positional_encoding")


class _EncoderLayer(_Params):
    def __init__(self, E, rotary):
        super().__init__()
        self.self_attn = _SelfAttn(E, rotary)
        self.linear1 = _lin(E, E)
        self.linear2 = _lin(E, E)
        self.norm1 = _ln(E)
        self.norm2 = _ln(E)


class _SliceFusion(_Params):  # nn.TransformerEncoder(num_layers=1, norm=LayerNorm)  dino.py:84-96
    def __init__(self, E, rotary):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(E, rotary)])
        self.norm = _ln(E)


# ------------------------------------------------------------------------------------------------
class DinoV2ClassifierSlice(BasicClassifier):
    def __init__(self, in_ch, out_ch, spatial_dims=2, pretrained=True, save_attn=False,
                 rotary_positional_encoding=None, optimizer_kwargs={"lr": 1e-6, "weight_decay": 1e-2},
                 model_size="s", use_registers=False, use_bottleneck=False, use_slice_pos_emb=False,
                 enable_linear=True, enable_trans=True, slice_fusion="transformer", freeze=False, **kwargs):
        # build-specific keywords are popped before the reference's base class sees **kwargs
        compute_dtype = str(kwargs.pop("compute_dtype", os.environ.get("MST_COMPUTE_DTYPE", "bf16"))).lower()
        chunk_slices = int(kwargs.pop("chunk_slices", os.environ.get("MST_CHUNK_SLICES", 0)))
        full_attention_maps = bool(kwargs.pop("full_attention_maps", False))
        # opt-in (default off: every block computes every token, as the reference does): the last block computes only what is read
        # behind it -- K/V of every token and the attention + MLP of the class tokens; results are the reference's
        prune_last_block = bool(int(kwargs.pop("prune_last_block", os.environ.get("MST_PRUNE_LAST_BLOCK", 0))))
        # hipGraph replay of the inference forward for small fixed shapes.  OPT-IN ("1"; "auto" = calls of at most MST_GRAPH_MAX_TOKENS
        # tokens): measured on the MI355X box the 16 x 224^2 forward takes 1.083 ms eager and 1.082 ms replayed -- the ~90 launches are
        # not the bound, the under-filled kernels are (profiles/r04d_small_shapes.txt).  Results are the eager ones bit for bit.
        use_graph = str(kwargs.pop("use_graph", os.environ.get("MST_USE_GRAPH", "0"))).lower()
        train_precision = str(kwargs.pop("train_precision", os.environ.get("MST_TRAIN_PRECISION", "fp32"))).lower()
        if train_precision not in ("fp32", "bf16", "fp16"):
            raise ValueError("train_precision must be 'fp32', 'bf16' or 'fp16'")
        if compute_dtype not in hip.DT_NAMES:
            raise ValueError(f"compute_dtype must be one of {sorted(hip.DT_NAMES)}")
        super().__init__(in_ch, out_ch, spatial_dims=spatial_dims, optimizer_kwargs=optimizer_kwargs, **kwargs)
        self.compute_dtype_name = compute_dtype
        self._fp8_amax = self._fp8_calib = None      # calibrate_fp8(): static activation scales of the fp8 mode
        self._fp8_collect = False
        self.chunk_slices = chunk_slices
        self.full_attention_maps = full_attention_maps
        self.prune_last_block = prune_last_block
        self.use_graph = use_graph
        self.train_precision = train_precision
        self._graphs = {}
        self.save_attn = save_attn
        self.attention_maps = []
        self.attention_maps_slice = []
        self.use_registers = use_registers
        self.slice_fusion_type = slice_fusion
        self.rotary = rotary_positional_encoding
        self.model_size = model_size

        cfg = _VIT_SIZES[model_size]
        if model_size == "g":
            raise NotImplementedError("model_size='g' needs the SwiGLU FFN (out of scope: SURVEY.md section 2)")
        if pretrained:
            # reference dino.py:59-63: torch.hub fetch (needs network); hub models are img 518, LayerScale, unchunked
            name = f"dinov2_vit{model_size}14_reg" if use_registers else f"dinov2_vit{model_size}14"
            hub = torch.hub.load("facebookresearch/dinov2", name)
            self.encoder = _ViT(**cfg, img_size=518, num_register_tokens=4 if use_registers else 0,
                                layerscale=1.0, chunked=False)
            self.encoder.load_state_dict(hub.state_dict(), strict=True)
        else:
            self.encoder = _ViT(**cfg, img_size=224, num_register_tokens=0, layerscale=None, chunked=True)
        if cfg["embed_dim"] != cfg["num_heads"] * 64:
            raise NotImplementedError("the HIP attention kernels are built for head_dim 64")
        if freeze:
            for p in self.encoder.parameters():
                p.requires_grad = False

        emb_ch = self.encoder.num_features
        if use_bottleneck:
            self.bottleneck = _lin(emb_ch // 4, emb_ch)
            emb_ch = emb_ch // 4
        self.emb_ch = emb_ch
        if slice_fusion == "transformer":
            if use_slice_pos_emb:
                holder = _Params()
                holder.weight = nn.Parameter(torch.randn(256, emb_ch))  # nn.Embedding(256, emb_ch), dino.py:82
                self.slice_pos_emb = holder
            self.slice_fusion = _SliceFusion(emb_ch, rotary_positional_encoding)
            self.cls_token = nn.Parameter(torch.randn(1, 1, emb_ch))
        elif slice_fusion == "linear":
            emb_ch = emb_ch * 32
        elif slice_fusion == "average":
            pass
        self.linear = _lin(out_ch, emb_ch) if enable_linear else nn.Identity()

        self._prep = None       # prepared device-side weights (struct + tensors kept alive)
        self._prep_sig = None
        self._param_epoch = 0
        self._sentinels = None
        self.profiler: Optional[hip.Profiler] = None   # bench only: set to a hip.Profiler to time this model's launches
        self._cls_last = None
        self._local = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate_prepared())
        self._pos_cache = {}
        self._ws = {}
        self._sharding: Optional[SliceSharding] = None
        self._warned_grad = False
        self._register_load_state_dict_pre_hook(self._remap_state_dict)

    def _invalidate_prepared(self):
        self._param_epoch = getattr(self, "_param_epoch", 0) + 1
        self._prep = self._prep_sig = self._sentinels = None
        self._graphs = {}                               # captured graphs hold pointers into the prepared weight images

    # ---- checkpoint compatibility ---------------------------------------------------------------
    def _remap_state_dict(self, state_dict, prefix, *args):
        """Accept the other block layout: hub ``blocks.<i>.`` <-> chunked ``blocks.0.<i>.``."""
        want_chunked = self.encoder.chunked
        for k in list(state_dict.keys()):
            if not k.startswith(prefix + "encoder.blocks."):
                continue
            rest = k[len(prefix + "encoder.blocks."):].split(".")
            is_chunked = len(rest) > 2 and rest[0].isdigit() and rest[1].isdigit()
            if is_chunked and not want_chunked:
                state_dict[prefix + "encoder.blocks." + ".".join(rest[1:])] = state_dict.pop(k)
            elif not is_chunked and want_chunked:
                state_dict[prefix + "encoder.blocks.0." + ".".join(rest)] = state_dict.pop(k)

    # ---- multi-GPU: slices of every volume sharded over the ranks of a process group --------------
    def enable_slice_sharding(self, group=None, sharding: Optional[SliceSharding] = None, gather_all_layers: bool = False):
        """Encode slices [r*D/G, (r+1)*D/G) on rank r and all-gather the slice embeddings (RCCL over
        xGMI) before the replicated Slice Transformer.  Call with the same input on all ranks.

        With ``save_attn`` a second all-gather ships the LAST block's CLS rows (2.1 MB at 64 x 518^2: all that
        ``get_plane_attention`` / ``get_attention_maps`` read, reference dino.py:190); ``attention_maps[:-1]`` then hold
        this rank's own slices only.  ``gather_all_layers=True`` gathers every block's rows (the unsharded list).
        ``get_attention_cls`` chains each rank's own full maps and gathers the result.
        ``sharding``: a prepared transport (tests rehearse several ranks on one GPU with tools/rehearsal.py)."""
        self._sharding = sharding if sharding is not None else SliceSharding(group)
        self._gather_all_layers = bool(gather_all_layers)
        return self

    def disable_slice_sharding(self):
        self._sharding = None

    # ---- weight preparation -------------------------------------------------------------------------
    # ---- fp8 mode: calibrated (static) activation scales ------------------------------------------------
    def calibrate_fp8(self, source, margin: float = 1.0, **kwargs):
        """compute_dtype='fp8' only.  Runs ``forward(source, **kwargs)`` with dynamic scales, records max|x| of the inputs
        of the four linear layers of every block ([depth, 4], kept over successive calls until ``reset_fp8_calibration``),
        and switches later forwards to those scales times ``margin``: LayerNorm and the GELU epilogue then write e4m3 directly
        and nothing is scanned.  Values beyond the calibrated range saturate.  Under slice sharding the table is max-reduced over
        the ranks.  Returns the scale table (a copy)."""
        if self.compute_dtype_name not in hip.FP8_NAMES:
            raise RuntimeError("calibrate_fp8 needs compute_dtype='fp8'")
        if self._fp8_calib is None:
            self._fp8_calib = torch.zeros(self.encoder.depth * 4, dtype=torch.float32, device=self.device)
        self._fp8_amax, self._fp8_collect = None, True
        try:
            with torch.no_grad():
                self(source, **kwargs)
        finally:
            self._fp8_collect = False
        if self._sharding is not None and self._sharding.world_size > 1:
            self._sharding.all_reduce_max(self._fp8_calib)       # every rank saw other slices: one scale table for all of them
        self._fp8_amax = (self._fp8_calib * float(margin)).contiguous()
        return self._fp8_amax.view(self.encoder.depth, 4).clone()

    def reset_fp8_calibration(self):
        """Back to dynamic per-call scales."""
        self._fp8_amax = self._fp8_calib = None
        self._fp8_collect = False

    def _signature(self, full: bool):
        """Identity of everything ``_prepare`` folded into device-side weight images.  With grad enabled (optimiser steps rewrite
        parameters in place) every parameter's (data_ptr, _version) pair is part of the signature.  With grad disabled the walk
        is ONE pass over a cached parameter list summing ``_version`` and folding ``data_ptr`` (~15 us for ~200 tensors: a tuple of
        200 pairs cost ~50 us, 5 % of the 1.1 ms c1 forward): any in-place edit of ANY parameter -- weight surgery on one block,
        ``model.slice_fusion.load_state_dict(...)``, a partial fine-tune -- changes it (four sentinel tensors did not)."""
        head = (self.compute_dtype_name, None if self._fp8_amax is None else (self._fp8_amax.data_ptr(), self._fp8_amax._version),
                self._fp8_collect, self._param_epoch)
        if self._sentinels is None:
            self._sentinels = list(self.parameters())       # rebuilt by _invalidate_prepared (``_apply`` / load_state_dict replace tensors)
        ps = self._sentinels
        if full:
            return head + tuple((q.data_ptr(), q._version) for q in ps)
        acc = 0
        for q in ps:
            acc = (acc * 1000003 + q._version + q.data_ptr()) & 0xFFFFFFFFFFFFFFFF
        return head + (acc, len(ps))

    def invalidate(self):
        """Public: drop every prepared device-side weight image (call after editing parameters through ``.data`` views that bypass
        the version counters)."""
        self._invalidate_prepared()

    def _apply(self, fn, *args, **kwargs):
        self._invalidate_prepared()
        return super()._apply(fn, *args, **kwargs)

    def _prepare(self):
        full = torch.is_grad_enabled()
        sig = self._signature(full)
        if self._prep is not None and sig == self._prep_sig[full]:
            return self._prep
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(f"DinoV2ClassifierSlice runs on an MI355X only: parameters are on {dev}; call .to('cuda'). "
                               "There is no CPU fallback.")
        cdt = hip.DT_NAMES[self.compute_dtype_name]
        tdt = hip.TORCH_DT[cdt]
        fp8 = self.compute_dtype_name in hip.FP8_NAMES
        keep: List[torch.Tensor] = []

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        def cmp(t):
            t = t.detach().to(device=dev, dtype=tdt).contiguous()
            keep.append(t)
            return t

        enc = self.encoder
        E, R = enc.embed_dim, enc.num_register_tokens
        with torch.no_grad():
            w = enc.patch_embed.proj.weight.detach().to(dev, torch.float32)          # [E,3,14,14]
            wp = torch.zeros(E, PATCH, 16, device=dev, dtype=torch.float32)
            wp[:, :, :PATCH] = w.sum(dim=1)                                          # RGB copies are identical
            wp = cmp(wp.reshape(E, PATCH * 16))
            pos = f32(enc.pos_embed[0])                                              # [1+M*M, E]
            prefix = (enc.cls_token.detach().to(dev, torch.float32)[0] + pos[:1])
            if R:
                prefix = torch.cat([prefix, enc.register_tokens.detach().to(dev, torch.float32)[0]], dim=0)
            prefix = f32(prefix)
        layers = (hip.VitLayer * enc.depth)()
        for i, b in enumerate(enc.block_list()):
            L = layers[i]
            L.ln1_w, L.ln1_b = hip.ptr(f32(b.norm1.weight)), hip.ptr(f32(b.norm1.bias))
            L.qkv_w, L.qkv_b = hip.ptr(cmp(b.attn.qkv.weight)), hip.ptr(f32(b.attn.qkv.bias))
            L.proj_w, L.proj_b = hip.ptr(cmp(b.attn.proj.weight)), hip.ptr(f32(b.attn.proj.bias))
            L.ls1 = hip.ptr(f32(b.ls1.gamma)) if hasattr(b, "ls1") else None
            L.ln2_w, L.ln2_b = hip.ptr(f32(b.norm2.weight)), hip.ptr(f32(b.norm2.bias))
            L.fc1_w, L.fc1_b = hip.ptr(cmp(b.mlp.fc1.weight)), hip.ptr(f32(b.mlp.fc1.bias))
            L.fc2_w, L.fc2_b = hip.ptr(cmp(b.mlp.fc2.weight)), hip.ptr(f32(b.mlp.fc2.bias))
            L.ls2 = hip.ptr(f32(b.ls2.gamma)) if hasattr(b, "ls2") else None
            if fp8:
                # BASELINE configs[4]: the four block matrices as OCP e4m3 bytes + per-tensor scales (mst_gemm_fp8)
                for j, (name, lin) in enumerate((("qkv_w8", b.attn.qkv), ("proj_w8", b.attn.proj),
                                                 ("fc1_w8", b.mlp.fc1), ("fc2_w8", b.mlp.fc2))):
                    q8, sc = hip.quantize_weight_fp8(lin.weight.to(dev))
                    keep.append(q8)
                    setattr(L, name, hip.ptr(q8))
                    L.w8_scale[j] = sc
                continue
            if cdt != hip.F32 and E == 384:
                # fused-LayerNorm form: norm1 folded into QKV, norm2 folded into the packed MLP (mst_mlp_fused)
                with torch.no_grad():
                    g1 = b.norm1.weight.detach().to(dev, torch.float32)
                    be1 = b.norm1.bias.detach().to(dev, torch.float32)
                    wq = b.attn.qkv.weight.detach().to(dev, torch.float32)
                    L.qkv_wf = hip.ptr(cmp(wq * g1[None, :]))
                    L.qkv_bf = hip.ptr(f32(b.attn.qkv.bias.detach().to(dev, torch.float32) + (wq * be1[None, :]).sum(dim=1)))   # (no vendor BLAS: an elementwise product + row sums, once per weight version)
                    ls = b.ls2.gamma.detach().to(dev) if hasattr(b, "ls2") else None
                    wpack, b1p, b2p = hip.pack_mlp(b.mlp.fc1.weight.detach().to(dev), b.mlp.fc1.bias.detach().to(dev),
                                                   b.mlp.fc2.weight.detach().to(dev), b.mlp.fc2.bias.detach().to(dev),
                                                   b.norm2.weight.detach().to(dev), b.norm2.bias.detach().to(dev), ls, tdt)
                    ppack, pbf = hip.pack_proj(b.attn.proj.weight.detach().to(dev), b.attn.proj.bias.detach().to(dev),
                                               b.ls1.gamma.detach().to(dev) if hasattr(b, "ls1") else None, tdt)
                    bseq, _, _, _ = hip.pack_block_seq(b.attn.proj.weight.detach().to(dev), b.attn.proj.bias.detach().to(dev),
                                                       b.ls1.gamma.detach().to(dev) if hasattr(b, "ls1") else None,
                                                       b.mlp.fc1.weight.detach().to(dev), b.mlp.fc1.bias.detach().to(dev),
                                                       b.mlp.fc2.weight.detach().to(dev), b.mlp.fc2.bias.detach().to(dev),
                                                       b.norm2.weight.detach().to(dev), b.norm2.bias.detach().to(dev), ls, tdt)
                    keep.extend([wpack, b1p, b2p, ppack, pbf, bseq])
                    L.mlp_pack, L.fc1_bf, L.fc2_bf = hip.ptr(wpack), hip.ptr(b1p), hip.ptr(b2p)
                    L.proj_pack, L.proj_bf = hip.ptr(ppack), hip.ptr(pbf)
                    L.block_seq = hip.ptr(bseq)
        vit = hip.VitWeights()
        vit.embed_dim, vit.depth, vit.num_heads, vit.num_registers = E, enc.depth, enc.num_heads, R
        vit.compute_dtype = cdt
        vit.fp8_linear = 1 if fp8 else 0
        if fp8 and self._fp8_amax is not None:
            vit.fp8_amax = hip.ptr(self._fp8_amax)
        elif fp8 and self._fp8_collect:
            vit.fp8_amax_out = hip.ptr(self._fp8_calib)
        vit.patch_w, vit.patch_b = hip.ptr(wp), hip.ptr(f32(enc.patch_embed.proj.bias))
        vit.prefix = hip.ptr(prefix)
        vit.layers = layers
        vit.norm_w, vit.norm_b = hip.ptr(f32(enc.norm.weight)), hip.ptr(f32(enc.norm.bias))

        fw = hip.FusionWeights()
        fw.emb_in, fw.emb, fw.num_heads = E, self.emb_ch, SLICE_HEADS
        fw.fusion_type = {"transformer": hip.FUSION_TRANSFORMER, "linear": hip.FUSION_LINEAR,
                          "average": hip.FUSION_AVERAGE}[self.slice_fusion_type]
        if hasattr(self, "bottleneck"):
            fw.bottleneck_w, fw.bottleneck_b = hip.ptr(f32(self.bottleneck.weight)), hip.ptr(f32(self.bottleneck.bias))
        if hasattr(self, "slice_pos_emb"):
            fw.slice_pos_emb = hip.ptr(f32(self.slice_pos_emb.weight))
        if self.slice_fusion_type == "transformer":
            lay = self.slice_fusion.layers[0]
            fw.cls_token = hip.ptr(f32(self.cls_token.reshape(-1)))
            fw.ln1_w, fw.ln1_b = hip.ptr(f32(lay.norm1.weight)), hip.ptr(f32(lay.norm1.bias))
            fw.in_proj_w, fw.in_proj_b = hip.ptr(f32(lay.self_attn.in_proj_weight)), hip.ptr(f32(lay.self_attn.in_proj_bias))
            fw.out_proj_w, fw.out_proj_b = hip.ptr(f32(lay.self_attn.out_proj.weight)), hip.ptr(f32(lay.self_attn.out_proj.bias))
            fw.ln2_w, fw.ln2_b = hip.ptr(f32(lay.norm2.weight)), hip.ptr(f32(lay.norm2.bias))
            fw.lin1_w, fw.lin1_b = hip.ptr(f32(lay.linear1.weight)), hip.ptr(f32(lay.linear1.bias))
            fw.lin2_w, fw.lin2_b = hip.ptr(f32(lay.linear2.weight)), hip.ptr(f32(lay.linear2.bias))
            fw.norm_w, fw.norm_b = hip.ptr(f32(self.slice_fusion.norm.weight)), hip.ptr(f32(self.slice_fusion.norm.bias))
            rot = getattr(lay.self_attn, "rotary_positional_encoding", None)
            if rot is not None and hasattr(rot, "freqs"):
                fw.rope_freqs = hip.ptr(f32(rot.freqs))
            elif rot is not None:                      # LieRE: R = block_diag(exp(A_blk)) once per weight version
                R = hip.liere_rotation(list(rot.vars))
                keep.append(R)
                fw.liere_rot = hip.ptr(R)
        if isinstance(self.linear, nn.Identity):
            fw.out_ch = 0
        else:
            fw.out_ch = self.out_ch
            fw.head_w, fw.head_b = hip.ptr(f32(self.linear.weight)), hip.ptr(f32(self.linear.bias))
            fw.head_in = self.linear.weight.shape[1]
        self._prep = dict(vit=vit, layers=layers, fusion=fw, keep=keep, pos=pos, cdt=cdt)
        self._prep_sig = {True: self._signature(True), False: self._signature(False)}
        self._pos_cache = {}
        return self._prep

    def _pos_patch(self, prep, H, W):
        """vision_transformer.py:179-211: stored grid when it matches, bicubic resample otherwise."""
        gh, gw = H // PATCH, W // PATCH
        key = (gh, gw, H == W)
        if key not in self._pos_cache:
            pos = prep["pos"]
            n_stored = pos.shape[0] - 1
            if gh * gw == n_stored and H == W:
                pp = pos[1:].contiguous()
            else:
                M = int(math.sqrt(n_stored))
                assert n_stored == M * M
                # Register encoders only exist as the hub's dinov2_vit*14_reg (dino.py:61), which facebookresearch/dinov2
                # hub/backbones.py builds with interpolate_antialias=True, interpolate_offset=0.0; all others keep the
                # vendored defaults (vision_transformer.py:66-67)
                reg = self.encoder.num_register_tokens > 0
                pp = hip.pos_embed_interp(pos[1:].contiguous(), M, gh, gw, 0.0 if reg else 0.1, antialias=reg)
            self._pos_cache[key] = pp
        return self._pos_cache[key]

    def _workspace(self, name, nbytes, dev):
        t = self._ws.get(name)
        if t is None or t.numel() < nbytes or t.device != dev:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            self._ws[name] = t
        return t

    def _auto_chunk(self, n, N, E, cdt):
        if self.chunk_slices > 0:
            return min(n, self.chunk_slices)
        env = os.environ.get("MST_WS_BUDGET_MB")
        budget = (int(env) << 20) if env else (8 << 30)      # activations per pass
        per_slice = N * E * (4 + (2 if cdt != hip.F32 else 4) * 5)
        return max(1, min(n, budget // per_slice))

    # ---- the hot path ------------------------------------------------------------------------------
    def encode_slices(self, slices: torch.Tensor, n_layers_probs: int = 0, full: bool = False):
        """Per-slice DINOv2 encoder on ``[n,H,W]`` device slices -> (emb [n,E], cls_probs, full_probs)."""
        prep = self._prepare()
        n, H, W = slices.shape
        assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
        assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
        vit = prep["vit"]
        vit.grid_h, vit.grid_w = H // PATCH, W // PATCH
        pp = self._pos_patch(prep, H, W)
        vit.pos_patch = hip.ptr(pp)
        vit.profiler = self.profiler.handle if self.profiler is not None else None
        vit.prune_last_block = 1 if self.prune_last_block else 0
        enc = self.encoder
        E, heads = enc.embed_dim, enc.num_heads
        N = 1 + enc.num_register_tokens + vit.grid_h * vit.grid_w
        dev = slices.device
        chunk = self._auto_chunk(n, N, E, prep["cdt"])
        ws = self._workspace("vit", hip.vit_workspace_bytes(vit, H, W, chunk), dev)
        emb = torch.empty((n, E), dtype=torch.float32, device=dev)
        cls_probs = full_probs = None
        if n_layers_probs:
            cls_probs = torch.empty((n_layers_probs, n, heads, N), dtype=torch.float32, device=dev)
            if full:
                full_probs = torch.empty((n_layers_probs, n, heads, N, N), dtype=torch.float32, device=dev)
        hip.vit_encode(vit, slices, emb, cls_probs, n_layers_probs, chunk, ws, full_probs)
        return emb, cls_probs, full_probs

    def fuse_slices(self, emb: torch.Tensor, B: int, D: int, mask: Optional[torch.Tensor], want_probs: bool,
                    want_logits: bool):
        prep = self._prepare()
        fw = prep["fusion"]
        dev = emb.device
        F = self.emb_ch * D if self.slice_fusion_type == "linear" else self.emb_ch
        if want_logits and F != self.linear.weight.shape[1]:     # what nn.Linear raises in the reference (dino.py:166)
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{F} and "
                               f"{self.linear.weight.shape[1]}x{self.linear.weight.shape[0]})")
        features = torch.empty((B, F), dtype=torch.float32, device=dev)
        logits = torch.empty((B, self.out_ch), dtype=torch.float32, device=dev) if want_logits else None
        probs = None
        if want_probs and self.slice_fusion_type == "transformer":
            probs = torch.empty((B, SLICE_HEADS, D + 1, D + 1), dtype=torch.float32, device=dev)
        m = None
        if mask is not None and self.slice_fusion_type == "transformer":
            m = mask.to(device=dev).to(torch.uint8).contiguous()
            if tuple(m.shape) != (B, D):
                raise RuntimeError(f"src_key_padding_mask must be [B, D] = {(B, D)}, got {tuple(m.shape)}")
        ws = self._workspace("fusion", hip.fusion_workspace_bytes(fw, B, D), dev)
        hip.slice_fusion(fw, emb.contiguous(), B, D, m, features, logits, probs, ws)
        return features, logits, probs

    def forward(self, source, save_attn=False, src_key_padding_mask=None, **kwargs):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training step (base_model.py:148-181): logits with ONE autograd node whose backward runs on the HIP kernels too
            if save_attn:
                raise NotImplementedError("save_attn inside a training forward: run the attention read-outs under torch.no_grad()")
            if self._sharding is not None and self._sharding.world_size > 1:
                raise NotImplementedError("training under slice sharding: use data-parallel ranks (DDP) for the training step")
            from .. import train
            return train.forward_with_grad(self, source, src_key_padding_mask, bool(kwargs.get("without_linear", False)))
        x = source.to(self.device)                      # [B, C, D, H, W]  (reference dino.py:121)
        B, C, D0, H, W = x.shape
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x = x.float()
        if self._graph_wanted(x, save_attn, kwargs):
            return self._forward_graph(x, save_attn, src_key_padding_mask, kwargs)
        return self._forward_eager(x, save_attn, src_key_padding_mask, kwargs)

    # ---- hipGraph replay (small fixed shapes) -----------------------------------------------------------------------------
    def _graph_wanted(self, x, save_attn, kwargs) -> bool:
        if self.use_graph in ("0", "false", "off") or x.device.type != "cuda":
            return False
        if (self._sharding is not None and self._sharding.world_size > 1) or self.profiler is not None or self._fp8_collect:
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        if self.use_graph in ("1", "true", "on"):
            return True
        B, C, D0, H, W = x.shape
        tokens = B * C * D0 * (1 + self.encoder.num_register_tokens + (H // PATCH) * (W // PATCH))
        return tokens <= int(os.environ.get("MST_GRAPH_MAX_TOKENS", 40000)) and not (save_attn and self.full_attention_maps)

    def _forward_graph(self, x, save_attn, mask, kwargs):
        """Replay of a captured eager forward.  Key: everything that selects kernels or buffer sizes.  The first TWO calls of a key
        run eagerly (lazy one-time work -- LDS attributes, workspaces, position-grid interpolation -- must not happen inside a
        capture); the third captures.  Inputs are copied into the graph's static buffers, outputs are cloned out of them."""
        prep = self._prepare()
        without_linear = bool(kwargs.get("without_linear", False))
        key = (tuple(x.shape), x.dtype, bool(save_attn), mask is not None, without_linear, id(prep), torch.cuda.current_device())
        ent = self._graphs.get(key)
        if ent is None or ent["graph"] is None:
            if ent is None:
                ent = self._graphs[key] = {"graph": None, "calls": 0}
            ent["calls"] += 1
            if ent["calls"] < 3:
                return self._forward_eager(x, save_attn, mask, kwargs)
            # capture (nothing executes while capturing: the replay below produces this call's results); static copies of the inputs
            sx = x.clone()
            sm = None if mask is None else mask.to(self.device).clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    sout = self._forward_eager(sx, save_attn, sm, kwargs)
                    state = (self.attention_maps, self.attention_maps_slice, getattr(self, "_cls_last", None)) if save_attn else None
            except Exception:
                self._graphs[key] = {"graph": None, "calls": -(1 << 30)}     # this key stays eager
                return self._forward_eager(x, save_attn, mask, kwargs)
            ent.update(graph=g, x=sx, mask=sm, out=sout, state=state)
        ent["x"].copy_(x)
        if ent["mask"] is not None:
            ent["mask"].copy_(mask)
        ent["graph"].replay()
        if save_attn:
            maps, maps_slice, cls_last = ent["state"]
            self.attention_maps = [m.clone() for m in maps]
            self.attention_maps_slice = [m.clone() for m in maps_slice]
            self._cls_last = None if cls_last is None else cls_last.clone()
        self._local = None
        self._last_shape = (x.shape[0], x.shape[1] * x.shape[2])
        return ent["out"].clone()

    def _forward_eager(self, x, save_attn, src_key_padding_mask, kwargs):
        B, C, D0, H, W = x.shape
        if C != 1:
            x = x.permute(0, 2, 1, 3, 4)                # 'b c d h w -> (b d c) h w'  (dino.py:125)
        D = D0 * C
        slices = x.reshape(B, D, H, W)
        n_probs = 0
        if save_attn:
            self.attention_maps, self.attention_maps_slice = [], []
            n_probs = self.encoder.depth
        want_full = bool(save_attn and self.full_attention_maps)

        without_linear = bool(kwargs.get("without_linear", False))
        want_logits = (not without_linear) and not isinstance(self.linear, nn.Identity)
        if want_logits and self.slice_fusion_type == "linear" and D * self.emb_ch != self.linear.weight.shape[1]:
            # nn.Linear's complaint in the reference (dino.py:155,166: the head is built for 32 slices), before any GPU work
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{D * self.emb_ch} and "
                               f"{self.linear.weight.shape[1]}x{self.linear.weight.shape[0]})")
        self._local = None                              # (d0, d1, dpad): this rank's slices when the maps below are sharded
        if self._sharding is not None and self._sharding.world_size > 1:
            sh = self._sharding
            d0, d1, dpad = sh.shard_range(D)
            dl = d1 - d0
            local = slices[:, d0:d1]
            if dl < dpad:                               # equal-sized shards for the collective
                local = torch.cat([local, local.new_zeros(B, dpad - dl, H, W)], dim=1)
            emb_l, probs_l, full_l = self.encode_slices(local.reshape(B * dpad, H, W).contiguous(), n_probs, want_full)
            emb = sh.all_gather_slices(emb_l.view(B, dpad, -1), D).reshape(B * D, -1)
            cls_probs = full_probs = cls_last = None
            if probs_l is not None:                     # [12, B*dpad, h, N]: gather along the slice axis
                Lp, _, hh, NN = probs_l.shape
                if getattr(self, "_gather_all_layers", False):
                    g = sh.all_gather_slices(probs_l.view(Lp, B, dpad, hh * NN).permute(1, 2, 0, 3).reshape(B, dpad, Lp * hh * NN), D)
                    cls_probs = g.view(B, D, Lp, hh, NN).permute(2, 0, 1, 3, 4).reshape(Lp, B * D, hh, NN).contiguous()
                    cls_last = cls_probs[-1]
                else:                                   # only the last block's rows are read by the getters (dino.py:190)
                    cls_last = sh.all_gather_slices(probs_l[-1].view(B, dpad, hh * NN), D).view(B * D, hh, NN)
                    cls_probs = probs_l.view(Lp, B, dpad, hh, NN)[:, :, :dl].reshape(Lp, B * dl, hh, NN)
                    self._local = (d0, d1, dpad)
            if full_l is not None:                      # [12, B*dpad, h, N, N] stays on its rank (34.6 GB unsharded at c3)
                Lp = full_l.shape[0]
                full_probs = full_l.view(Lp, B, dpad, *full_l.shape[2:])[:, :, :dl].reshape(Lp, B * dl, *full_l.shape[2:])
                self._local = (d0, d1, dpad)
        else:
            emb, cls_probs, full_probs = self.encode_slices(slices.reshape(B * D, H, W).contiguous(), n_probs, want_full)
            cls_last = cls_probs[-1] if cls_probs is not None else None

        features, logits, slice_probs = self.fuse_slices(emb, B, D, src_key_padding_mask, bool(save_attn), want_logits)

        if save_attn:
            if want_full and full_probs is not None:
                self.attention_maps = [full_probs[l] for l in range(full_probs.shape[0])]
            else:
                self.attention_maps = [cls_probs[l][:, :, None, :] for l in range(cls_probs.shape[0])]
            if slice_probs is not None:
                self.attention_maps_slice = [slice_probs]
            self._cls_last = cls_last                     # [B*D, h, N] of ALL slices (gathered when sharded)
        self._last_shape = (B, D)
        return logits if want_logits else features

    # ---- attention read-outs (reference dino.py:173-212) -------------------------------------------
    def _readout(self, plane=False, slice_attn=False, maps=False):
        B, D = self._last_shape
        enc = self.encoder
        cls_last = self._cls_last.contiguous() if (plane or maps) else None                      # [n,h,N]
        sp = self.attention_maps_slice[-1].contiguous() if (slice_attn or maps) else None
        dev = (cls_last if cls_last is not None else sp).device
        R = 4 if self.use_registers else 0            # img_slice = slice(5, None): dino.py:191
        n, heads = B * D, enc.num_heads
        N = cls_last.shape[-1] if cls_last is not None else 2 + R
        out_plane = torch.empty((n, heads, N - 1 - R), dtype=torch.float32, device=dev) if plane else None
        out_maps = torch.empty((n, heads, N - 1 - R), dtype=torch.float32, device=dev) if maps else None
        out_slice = torch.empty((n,), dtype=torch.float32, device=dev) if (slice_attn or maps) else None
        hip.attention_readout(cls_last, sp, B, D, heads, N, R, SLICE_HEADS, out_plane, out_slice, out_maps)
        return out_plane, out_slice, out_maps

    def get_slice_attention(self):
        _, s, _ = self._readout(slice_attn=True)
        return s[:, None, None]                        # [B*D, 1, 1]

    def get_plane_attention(self):
        p, _, _ = self._readout(plane=True)
        return p                                       # [B*D, heads, Np]

    def get_attention_maps(self):
        _, _, m = self._readout(maps=True)
        return m                                       # [B*D, heads, Np]

    def get_attention_cls(self, gather: bool = True):
        """Attention rollout A_0 . A_1 ... A_last over the full maps (reference dino.py:204-212; no caller
        in the reference).  Needs ``full_attention_maps=True`` at construction.  Under slice sharding every rank chains
        the maps of its own slices (the chain is independent per slice and head) and the result is all-gathered to
        ``[B*D, heads, N, N]`` on every rank; ``gather=False`` returns this rank's ``[B*d_local, heads, N, N]``."""
        if not self.attention_maps or self.attention_maps[-1].shape[-2] != self.attention_maps[-1].shape[-1]:
            raise RuntimeError("get_attention_cls needs the full [n,h,N,N] maps: construct the model with "
                               "full_attention_maps=True and run forward(save_attn=True)")
        maps = [m.contiguous() for m in self.attention_maps]
        if self._local is None or not gather:
            if maps[0].shape[0] == 0:
                return maps[0].clone()
            return hip.attention_rollout(maps)                                        # [B*D, heads, N, N]
        B, D = self._last_shape
        d0, d1, dpad = self._local
        _, hh, NN, _ = maps[0].shape
        padded = maps[0].new_zeros((B, dpad, hh * NN * NN))
        if d1 > d0:
            padded[:, : d1 - d0] = hip.attention_rollout(maps).view(B, d1 - d0, -1)
        return self._sharding.all_gather_slices(padded, D).view(B * D, hh, NN, NN)


class DinoV3ClassifierSlice(BasicClassifier):
    """Out of scope (weights only via remote hub / signed URLs).  The name exists
    for the imports and ``isinstance`` dispatch of scripts/main_train.py:19 / main_predict.py:28,142."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("DinoV3ClassifierSlice is out of scope of the MI355X build (SURVEY.md section 2)")
