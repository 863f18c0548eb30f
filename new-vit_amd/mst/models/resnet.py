"""MI355X-native ``ResNet`` / ``ResNetSliceTrans`` -- the inference path of the reference's mst/models/resnet.py:27-243
(BASELINE configs[3] names the model).

Same constructors, ``forward`` signatures, getters and ``state_dict`` keys as the reference's 2-D torchvision branch
(``self.model`` = torchvision ``resnet{18,34}`` with ``fc`` replaced: resnet.py:44-50; ``ResNetSliceTrans`` adds the Slice
Transformer with 16 heads over 512-wide slice embeddings, ``cls_token`` and ``linear``: resnet.py:146-166).  The modules below are
parameter containers; the arithmetic runs in HIP kernels through the C ABI:

  backbone   NHWC fp32 activations; every convolution behind the stem = ``mst_conv_gemm`` (implicit GEMM: the exact-fp32 MFMA GEMM
             gathering its A operand from the activation, no im2col matrix); the stem (one input channel after the gray fold) =
             ``mst_im2col_nhwc`` + ``mst_gemm``; eval-mode
             BatchNorm folded into weight and bias and ReLU / the residual add in the GEMM epilogue; ``mst_maxpool_nhwc``,
             ``mst_avgpool_nhwc``.  Gray -> RGB (``x.repeat(1, 3, ...)``, resnet.py:176) is folded into conv1 (the three input
             channels are identical, so their kernels are summed).
  fusion     ``mst_slice_fusion`` -- the kernels of DinoV2ClassifierSlice's Slice Transformer with nhead 16, E 512.

torchvision's source is not part of the reference tree: the published resnet architecture is restated, parity of
the backbone is against ``oracle/resnet_oracle.py`` (plain torch ops) and therefore UNPINNED; the fusion half is pinned by a
fixture of the reference's own ``TransformerEncoderLayer`` (tests/golden/resnet_fusion.npz).

  training   with gradients enabled in train mode the output carries one autograd node (mst/train_resnet.py): BatchNorm with batch
             statistics, convolution / pooling / slice-transformer backward in HIP kernels (BASELINE configs[3]'s step).
  Grad-CAM++ ``save_attn=True`` (resnet.py:62-118): the map of the LAST ReLU output -- the one ``get_attention_maps`` returns --
             by ``mst_gradcampp``; the maps of the earlier ReLUs, which the reference computes and never exposes, are not produced.

Bottleneck ResNets (model 50 / 101 / 152: 2048-wide slice embeddings, 16 heads of 128) run through the same kernels (round 3).
Not built (raise): the MONAI branches (3-D, and ``pretrained=False`` in the reference), a
backward pass through eval-mode BatchNorm (gradients with ``model.eval()``).  ``pretrained=True`` needs torchvision's weights (network / hub cache); when torchvision is absent the tree
is initialised like torchvision's and a warning says so -- a checkpoint's ``state_dict`` replaces it anyway.
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

_LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
_WIDTHS = [64, 128, 256, 512]


class _P(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward runs in libmst_hip.so")


class _Conv(_P):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")      # torchvision resnet.py init


class _BN(_P):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class _Linear(_P):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1 / math.sqrt(cin), 1 / math.sqrt(cin))


class _BasicBlock(_P):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = _Conv(cin, cout, 3), _BN(cout)
        self.conv2, self.bn2 = _Conv(cout, cout, 3), _BN(cout)
        self.stride = stride
        if stride != 1 or cin != cout:
            self.downsample = nn.ModuleList([_Conv(cin, cout, 1), _BN(cout)])       # keys downsample.0.* / downsample.1.*


class _Bottleneck(_P):
    """torchvision Bottleneck (v1.5): 1x1 -> 3x3 (carries the stride) -> 1x1 (x4), each with its BatchNorm (models 50 / 101 / 152:
    reference resnet.py:44-50 takes whichever torchvision model `model` names; emb_ch 2048, resnet.py:152)."""

    def __init__(self, cin, w, stride):
        super().__init__()
        self.conv1, self.bn1 = _Conv(cin, w, 1), _BN(w)
        self.conv2, self.bn2 = _Conv(w, w, 3), _BN(w)
        self.conv3, self.bn3 = _Conv(w, 4 * w, 1), _BN(4 * w)
        self.stride = stride
        if stride != 1 or cin != 4 * w:
            self.downsample = nn.ModuleList([_Conv(cin, 4 * w, 1), _BN(4 * w)])


class _TVResNet(_P):
    """Parameter tree of torchvision.models.resnet{18,34,50,101,152} (keys conv1, bn1, layer1..4.<i>.*, fc)."""

    def __init__(self, model: int, fc_out: Optional[int]):
        super().__init__()
        if model not in _LAYERS:
            raise NotImplementedError(f"resnet{model}: torchvision's resnet 18 / 34 / 50 / 101 / 152 are built on the HIP path")
        self.conv1, self.bn1 = _Conv(3, 64, 7), _BN(64)
        cin = 64
        for li, (n, w) in enumerate(zip(_LAYERS[model], _WIDTHS)):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and li > 0) else 1
                if model >= 50:
                    blocks.append(_Bottleneck(cin, w, stride))
                    cin = 4 * w
                else:
                    blocks.append(_BasicBlock(cin, w, stride))
                    cin = w
            setattr(self, f"layer{li + 1}", nn.ModuleList(blocks))
        self.out_features = cin
        self.fc = nn.Identity() if fc_out is None else _Linear(cin, fc_out)


def _conv(x: torch.Tensor, w: torch.Tensor, b, k: int, stride: int, pad: int, kpad: int, epilogue: int, out=None) -> torch.Tensor:
    """One convolution of the backbone on an NHWC activation -> [n*Ho*Wo, Cout]: the implicit GEMM (mst_conv_gemm) when the input
    channels allow it (Cin % 16 == 0: every layer behind the stem) and there are more than 1,024 output pixels, else im2col + GEMM.  MST_CONV_IM2COL=1 forces the latter (A/B)."""
    n, H, W, cin = x.shape
    if x.dtype != torch.float32:                         # compute_dtype bf16 / fp16: the implicit GEMM on 16-bit MFMA operands (Cin % 64 == 0)
        return hip.conv_gemm16(x, w, b, k, k, stride, pad, epilogue=epilogue, out=out)
    rows = n * ((H + 2 * pad - k) // stride + 1) * ((W + 2 * pad - k) // stride + 1)
    # up to 1,024 output pixels mst_gemm has its 32 x 32-tile kernel (k_gemm32s.hip), which fills the chip where 128 x 128 tiles cannot
    if cin % 16 == 0 and rows > 1024 and os.environ.get("MST_CONV_IM2COL", "0") != "1":
        return hip.conv_gemm(x, w, b, k, k, stride, pad, epilogue=epilogue, out=out)
    return hip.gemm(hip.im2col_nhwc(x, k, k, stride, pad, kpad), w, b, epilogue=epilogue, out=out)


def _fold(conv: _Conv, bn: _BN, sum_in: bool, dev, dtype: torch.dtype = torch.float32, kmul: int = 16):
    """Eval-mode BatchNorm folded into the convolution: GEMM weight [Cout, Kpad] in (ky, kx, c) order (fp32, or rounded to the 16-bit
    compute type AFTER the fold) + bias [Cout] fp32; K padded to a multiple of kmul."""
    w = conv.weight.detach().to(dev, torch.float32)
    if sum_in:
        w = w.sum(dim=1, keepdim=True)                   # identical input channels (gray -> RGB repeat): one summed kernel
    s = bn.weight.detach().to(dev, torch.float32) / torch.sqrt(bn.running_var.to(dev, torch.float32) + bn.eps)
    b = bn.bias.detach().to(dev, torch.float32) - bn.running_mean.to(dev, torch.float32) * s
    w = (w * s[:, None, None, None]).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    K = w.shape[1]
    kpad = (K + kmul - 1) // kmul * kmul
    if kpad != K:
        w = torch.cat([w, w.new_zeros(w.shape[0], kpad - K)], dim=1)
    return w.to(dtype).contiguous(), b.contiguous(), kpad


class ResNet(BasicClassifier):
    def __init__(self, in_ch, out_ch, spatial_dims=3, model=34, pretrained=False, kwargs_resnet={}, **kwargs):
        emb_ch = kwargs.pop("emb_ch", out_ch)
        self.chunk_images = int(kwargs.pop("chunk_images", 128))      # build-specific: images per backbone pass (activation memory:
        #                                                               0.4 GB for the stem's output at 128 images of 224 x 224)
        # build-specific: MFMA operand / activation type of the INFERENCE backbone: 'fp32' (default, exact: the parity mode) | 'bf16' | 'fp16'
        # (16-bit NHWC activations behind the stem's max pool, fp32 accumulation, folded BatchNorm bias in fp32; the slice transformer and
        # the head stay fp32).  The training step is fp32 whatever this says.
        cdt = str(kwargs.pop("compute_dtype", os.environ.get("MST_RESNET_DTYPE", "fp32"))).lower()
        if cdt not in ("fp32", "bf16", "fp16"):
            raise ValueError("ResNet: compute_dtype must be 'fp32', 'bf16' or 'fp16'")
        self.compute_dtype_name = cdt
        # build-specific: MFMA operand type of the TRAINING step's convolutions and their input gradients ('fp32' default | 'bf16' | 'fp16';
        # fp32 accumulation and storage; the reference trains under Trainer(precision='16-mixed'), scripts/main_train.py:110-123)
        tp = str(kwargs.pop("train_precision", os.environ.get("MST_TRAIN_PRECISION", "fp32"))).lower()
        if tp not in ("fp32", "bf16", "fp16"):
            raise ValueError("ResNet: train_precision must be 'fp32', 'bf16' or 'fp16'")
        self.train_precision = tp
        super().__init__(in_ch, out_ch, spatial_dims, **kwargs)
        self.attention_maps = []
        if spatial_dims != 2:
            raise NotImplementedError("ResNet: only the 2-D (torchvision) branch of the reference is built on the HIP path; the 3-D "
                                      "MONAI branch is not (SURVEY.md 8f-2)")
        fc_out = None if emb_ch is None else emb_ch
        self.model = _TVResNet(model, fc_out)
        if pretrained:
            try:                                         # reference resnet.py:44-45: torchvision weights="DEFAULT" (network / hub cache)
                import torchvision.models as tvm
                tv = getattr(tvm, f"resnet{model}")(weights="DEFAULT")
                sd = {k: v for k, v in tv.state_dict().items() if not k.startswith("fc.") or (emb_ch == 1000)}
                self.model.load_state_dict(sd, strict=False)
            except Exception as e:  # pragma: no cover - torchvision is not in this image
                warnings.warn(f"ResNet(pretrained=True): torchvision weights unavailable ({type(e).__name__}); the backbone keeps "
                              "torchvision's random initialisation until a checkpoint is loaded")
        else:
            warnings.warn("ResNet(pretrained=False): the reference builds a MONAI resnet here; this build keeps the torchvision "
                          "parameter layout with torchvision's initialisation")
        self._prep = None
        self._epoch = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    def _invalidate(self):
        self._prep = None

    def _apply(self, fn, *args, **kwargs):
        self._prep = None
        return super()._apply(fn, *args, **kwargs)

    # ---- backbone -------------------------------------------------------------------------------------------------------
    def _state_version(self):
        """Sum of the in-place version counters of every parameter and BatchNorm buffer plus their storage addresses: an optimiser
        step, mst_batchnorm_train's running-statistics update or any weight surgery changes it (the folded weights were
        keyed on (sum_in, device) only, so an eval forward after training steps reused the backbone folded at the first eval)."""
        v = 0
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            v += t._version + (t.data_ptr() & 0xFFFF)
        return v

    def _prepare(self, sum_in: bool):
        cdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[self.compute_dtype_name]
        key = (sum_in, str(self.device), self._state_version(), self.compute_dtype_name)
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(f"ResNet runs on an MI355X only: parameters are on {dev}; call .to('cuda'). There is no CPU fallback.")
        m = self.model
        # 16-bit: the stem's im2col rows are padded to 64 columns so that they are one "pixel" of a 1 x 1 convolution for mst_conv_gemm16
        prep = {"key": key, "dtype": cdt, "stem": _fold(m.conv1, m.bn1, sum_in, dev, cdt, 16 if cdt == torch.float32 else 64), "blocks": []}
        for li in range(4):
            for blk in getattr(m, f"layer{li + 1}"):
                e = {"stride": blk.stride, "c1": _fold(blk.conv1, blk.bn1, False, dev, cdt), "c2": _fold(blk.conv2, blk.bn2, False, dev, cdt)}
                if hasattr(blk, "conv3"):
                    e["c3"] = _fold(blk.conv3, blk.bn3, False, dev, cdt)
                if hasattr(blk, "downsample"):
                    e["ds"] = _fold(blk.downsample[0], blk.downsample[1], False, dev, cdt)
                prep["blocks"].append(e)
        if not isinstance(m.fc, nn.Identity):
            prep["fc"] = (m.fc.weight.detach().to(dev, torch.float32).contiguous(), m.fc.bias.detach().to(dev, torch.float32).contiguous())
        self._prep = prep
        return prep

    def _features(self, x_nhwc: torch.Tensor, sum_in: bool, keep_last: bool = False) -> torch.Tensor:
        """[n, H, W, C] fp32 on the device -> [n, 512] (avgpool output, before fc).  keep_last: the last ReLU output of every image
        stays in ``self._last_act`` [n, h*w, 512] for Grad-CAM++."""
        p = self._prepare(sum_in)
        outs: List[torch.Tensor] = []
        lasts: List[torch.Tensor] = []
        for i0 in range(0, x_nhwc.shape[0], self.chunk_images):
            x = x_nhwc[i0:i0 + self.chunk_images].contiguous()
            n, H, W, _ = x.shape
            w, b, kpad = p["stem"]
            Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            if p["dtype"] == torch.float32:
                y = hip.gemm(hip.im2col_nhwc(x, 7, 7, 2, 3, kpad), w, b, epilogue=hip.EPI_BIAS_RELU).view(n, Ho, Wo, 64)
                y = hip.maxpool_nhwc(y)
            else:
                # the stem's 49 taps of ONE input channel are no multiple of 64 channels: its im2col rows (padded to 64, rounded on the way out)
                # are the "pixels" of a 1 x 1 convolution; everything behind it, the max pool included, is 16-bit
                col = hip.im2col_nhwc(x, 7, 7, 2, 3, kpad, out_dtype=p["dtype"]).view(n, Ho, Wo, kpad)
                y = hip.conv_gemm16(col, w, b, 1, 1, 1, 0, epilogue=hip.EPI_BIAS_RELU).view(n, Ho, Wo, 64)
                del col
                y = hip.maxpool_nhwc(y)
            for e in p["blocks"]:
                n, H, W, Cin = y.shape
                s = e["stride"]
                Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
                w1, b1, k1 = e["c1"]
                if "c3" in e:                            # bottleneck: 1x1 -> 3x3 (stride) -> 1x1, the residual joins after the third
                    h0 = _conv(x=y, w=w1, b=b1, k=1, stride=1, pad=0, kpad=k1, epilogue=hip.EPI_BIAS_RELU).view(n, H, W, w1.shape[0])
                    w2, b2, k2 = e["c2"]
                    h1 = _conv(x=h0, w=w2, b=b2, k=3, stride=s, pad=1, kpad=k2, epilogue=hip.EPI_BIAS_RELU).view(n, Ho, Wo, w2.shape[0])
                    if "ds" in e:
                        wd, bd, kd = e["ds"]
                        idt = _conv(x=y, w=wd, b=bd, k=1, stride=s, pad=0, kpad=kd, epilogue=hip.EPI_BIAS)
                    else:
                        idt = y.reshape(n * H * W, Cin)          # the unit's input is dead after it: the sum is formed in place
                    w3, b3, k3 = e["c3"]
                    _conv(x=h1, w=w3, b=b3, k=1, stride=1, pad=0, kpad=k3, epilogue=hip.EPI_RESIDUAL_RELU, out=idt)
                    y = idt.view(n, Ho, Wo, w3.shape[0])
                    continue
                h1 = _conv(x=y, w=w1, b=b1, k=3, stride=s, pad=1, kpad=k1, epilogue=hip.EPI_BIAS_RELU).view(n, Ho, Wo, w1.shape[0])
                if "ds" in e:
                    wd, bd, kd = e["ds"]
                    idt = _conv(x=y, w=wd, b=bd, k=1, stride=s, pad=0, kpad=kd, epilogue=hip.EPI_BIAS)
                else:
                    idt = y.reshape(n * H * W, Cin)              # the unit's input is dead after it: the sum is formed in place
                w2, b2, k2 = e["c2"]
                _conv(x=h1, w=w2, b=b2, k=3, stride=1, pad=1, kpad=k2, epilogue=hip.EPI_RESIDUAL_RELU, out=idt)  # relu(identity + bn2(conv2(.)))
                y = idt.view(n, Ho, Wo, w2.shape[0])
            if y.dtype != torch.float32:
                y = hip.cvt32(y)                         # the average pool and Grad-CAM++ read fp32
            outs.append(hip.avgpool_nhwc(y))
            if keep_last:
                lasts.append(y)
        if keep_last:
            y = torch.cat(lasts, dim=0) if len(lasts) > 1 else lasts[0]
            self._last_hw = (y.shape[1], y.shape[2])
            self._last_act = y.view(y.shape[0], -1, y.shape[3])
        return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]

    def _wants_grad(self) -> bool:
        if not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters()):
            return False
        if not self.training:
            raise NotImplementedError("ResNet: gradients through eval-mode BatchNorm are not on the HIP path; call .train() for a "
                                      "training step or run the forward under torch.no_grad()")
        return True

    def _gradcam(self, out: torch.Tensor, fc_weight: Optional[torch.Tensor]):
        """resnet.py:66-70 + 93-118 for the last ReLU: attention_maps[-1] = [N, 1, h, w]."""
        h, w = self._last_hw
        cam = hip.gradcampp(self._last_act, out.detach().contiguous(), fc_weight)
        self.attention_maps = [cam.view(-1, 1, h, w)]
        self._last_act = None

    def forward(self, source, save_attn=False, **kwargs):
        x = source.to(self.device)                       # [N, C, H, W]
        if x.dim() != 4:
            raise RuntimeError(f"Expected 4D input [N, C, H, W] to the 2-D resnet, got {tuple(x.shape)}")
        if x.shape[1] != 3:
            raise RuntimeError(f"Given groups=1, weight of size [64, 3, 7, 7], expected input{list(x.shape)} to have 3 channels, "
                               f"but got {x.shape[1]} channels instead")
        x = x.float().permute(0, 2, 3, 1).contiguous()
        if self._wants_grad():
            if save_attn:
                raise NotImplementedError("ResNet: save_attn inside a training forward is not supported; compute the maps under "
                                          "torch.no_grad()")
            from .. import train_resnet
            return train_resnet.forward_with_grad(self, x, False)
        feat = self._features(x, False, keep_last=save_attn)
        p = self._prep
        out = hip.gemm(feat, p["fc"][0], p["fc"][1], epilogue=hip.EPI_BIAS) if "fc" in p else feat
        if save_attn:
            self._gradcam(out, p["fc"][0] if "fc" in p else None)
        return out

    def get_attention_maps(self):
        return self.attention_maps[-1]                   # [N, 1, h, w]  (resnet.py:75-76)


SLICE_HEADS = 16      # reference resnet.py:151


class _SliceLayer(_P):
    def __init__(self, E):
        super().__init__()
        holder = _P()
        holder.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        holder.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        holder.out_proj = _Linear(E, E)
        nn.init.xavier_uniform_(holder.in_proj_weight)
        nn.init.zeros_(holder.out_proj.bias)
        self.self_attn = holder
        self.linear1, self.linear2 = _Linear(E, E), _Linear(E, E)
        self.norm1, self.norm2 = _LN(E), _LN(E)


class _LN(_P):
    def __init__(self, E):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(E))
        self.bias = nn.Parameter(torch.zeros(E))


class _SliceFusion(_P):
    def __init__(self, E):
        super().__init__()
        self.layers = nn.ModuleList([_SliceLayer(E)])
        self.norm = _LN(E)


class ResNetSliceTrans(ResNet):
    def __init__(self, in_ch, out_ch, spatial_dims=2, model=34, pretrained=True, kwargs_resnet={},
                 rotary_positional_encoding=None, optimizer_kwargs={"lr": 1e-5, "weight_decay": 1e-2}, **kwargs):
        super().__init__(in_ch, out_ch, spatial_dims, emb_ch=None, model=model, pretrained=pretrained, kwargs_resnet=kwargs_resnet,
                         optimizer_kwargs=optimizer_kwargs, **kwargs)
        if rotary_positional_encoding is not None:
            raise NotImplementedError("ResNetSliceTrans: rotary variants are wired for DinoV2ClassifierSlice only")
        emb_ch = 512 if model <= 34 else 2048
        self.emb_ch = emb_ch
        self.attention_maps_slice = []
        self.slice_fusion = _SliceFusion(emb_ch)
        self.cls_token = nn.Parameter(torch.randn(1, 1, emb_ch))
        self.linear = _Linear(emb_ch, out_ch)
        self._fw = None

    def _invalidate(self):
        self._prep = None
        self._fw = None

    def _apply(self, fn, *args, **kwargs):
        self._fw = None
        return super()._apply(fn, *args, **kwargs)

    def _fusion_weights(self):
        # (copies exist only where a parameter is not already fp32 on the device; the version sum catches in-place updates of those too)
        ver = sum(t._version + (t.data_ptr() & 0xFFFF) for t in [self.cls_token, *self.slice_fusion.parameters(), *self.linear.parameters()])
        if self._fw is not None and self._fw[2] == ver:
            return self._fw[:2]
        dev = self.device
        keep = []

        def f32(t):
            t = t.detach().to(dev, torch.float32).contiguous()
            keep.append(t)
            return hip.ptr(t)

        lay = self.slice_fusion.layers[0]
        fw = hip.FusionWeights()
        fw.emb_in = fw.emb = self.emb_ch
        fw.num_heads, fw.fusion_type, fw.out_ch = SLICE_HEADS, hip.FUSION_TRANSFORMER, self.out_ch
        fw.cls_token = f32(self.cls_token.reshape(-1))
        fw.ln1_w, fw.ln1_b = f32(lay.norm1.weight), f32(lay.norm1.bias)
        fw.in_proj_w, fw.in_proj_b = f32(lay.self_attn.in_proj_weight), f32(lay.self_attn.in_proj_bias)
        fw.out_proj_w, fw.out_proj_b = f32(lay.self_attn.out_proj.weight), f32(lay.self_attn.out_proj.bias)
        fw.ln2_w, fw.ln2_b = f32(lay.norm2.weight), f32(lay.norm2.bias)
        fw.lin1_w, fw.lin1_b = f32(lay.linear1.weight), f32(lay.linear1.bias)
        fw.lin2_w, fw.lin2_b = f32(lay.linear2.weight), f32(lay.linear2.bias)
        fw.norm_w, fw.norm_b = f32(self.slice_fusion.norm.weight), f32(self.slice_fusion.norm.bias)
        fw.head_w, fw.head_b, fw.head_in = f32(self.linear.weight), f32(self.linear.bias), self.emb_ch
        self._fw = (fw, keep, ver)
        return self._fw[:2]

    def fuse(self, emb: torch.Tensor, B: int, D: int, src_key_padding_mask=None, save_attn: bool = False) -> torch.Tensor:
        """[B*D, 512] slice embeddings -> logits [B, out_ch] (resnet.py:180-191)."""
        fw, _ = self._fusion_weights()
        dev = emb.device
        feats = torch.empty((B, self.emb_ch), dtype=torch.float32, device=dev)
        logits = torch.empty((B, self.out_ch), dtype=torch.float32, device=dev)
        probs = torch.empty((B, SLICE_HEADS, D + 1, D + 1), dtype=torch.float32, device=dev) if save_attn else None
        m = None
        if src_key_padding_mask is not None:
            m = src_key_padding_mask.to(dev).to(torch.uint8).contiguous()
        ws = torch.empty(int(hip.fusion_workspace_bytes(fw, B, D)), dtype=torch.uint8, device=dev)
        hip.slice_fusion(fw, emb.contiguous(), B, D, m, feats, logits, probs, ws)
        if save_attn:
            self.attention_maps_slice = [probs]
        return logits

    def forward(self, source, src_key_padding_mask=None, **kwargs):
        save_attn = bool(kwargs.get("save_attn"))
        x = source.to(self.device)                       # [B, C, D, H, W]
        B, C, D, H, W = x.shape
        if C != 1:                                       # x.repeat(1, 3, ...) then a 3-channel conv1: only gray volumes fit (resnet.py:176)
            raise RuntimeError(f"Given groups=1, weight of size [64, 3, 7, 7], expected input[{B * D}, {3 * C}, {H}, {W}] to have 3 "
                               f"channels, but got {3 * C} channels instead")
        self._last_shape = (B, D)
        x = x.float().reshape(B * D, H, W, 1).contiguous()                              # 'b c d h w -> (b d) c h w', gray -> RGB folded
        if self._wants_grad():
            if save_attn:
                raise NotImplementedError("ResNetSliceTrans: save_attn inside a training forward is not supported; compute the maps "
                                          "under torch.no_grad()")
            from .. import train_resnet
            return train_resnet.forward_with_grad(self, x, True, B, D, src_key_padding_mask)
        emb = self._features(x, True, keep_last=save_attn)
        if save_attn:
            self._gradcam(emb, None)                     # super().forward(x, save_attn=True) with fc = Identity (resnet.py:181)
        return self.fuse(emb, B, D, src_key_padding_mask, save_attn)

    def get_slice_attention(self):
        B, D = self._last_shape
        sp = self.attention_maps_slice[-1].contiguous()                              # [B, heads, 1+D, 1+D]
        out = torch.empty((B * D,), dtype=torch.float32, device=sp.device)
        hip.attention_readout(None, sp, B, D, 1, 2, 0, SLICE_HEADS, None, out, None)
        return out[:, None, None]                        # [B*D, 1, 1]  (resnet.py:196-205)

    def get_attention_maps(self):
        return self.get_slice_attention().unsqueeze(-1) * super().get_attention_maps()      # [B*D, 1, h, w]  (resnet.py:207-212)
