"""Training step of ResNet / ResNetSliceTrans on the HIP path (BASELINE configs[3]).

The reference trains these through torch.autograd over torchvision's modules (mst/models/base_model.py:148-181 `_step`,
mst/models/resnet.py:172-193).  Here, as for DinoV2ClassifierSlice (mst/train.py), the logits carry ONE autograd node whose
forward runs the model op by op through the C ABI and whose backward produces every parameter gradient with HIP kernels:

  convolution      z = im2col(x) . Wg^T             (mst_conv_gemm: implicit GEMM, exact fp32 MFMA; the stem: mst_im2col_nhwc + mst_gemm;
                                                     Wg = weight in (ky, kx, c) order)
     backward      dWg = dz^T . im2col(x) as an implicit GEMM too (mst_conv_wgrad: pixel-split partial products reduced by mst_colsum),
                   dx = the stride-1 convolution of the stride-dilated dz with the flipped, transposed weight (mst_conv_dgrad); the
                   stem and MST_CONV_IM2COL=1 keep the explicit forms (mst_im2col_nhwc + mst_gemm_ex, mst_col2im_nhwc)
     16-bit        train_precision = bf16 / fp16 (the reference's Trainer(precision='16-mixed')): the three products on 16-bit MFMA
                   operands with fp32 accumulation (mst_conv_gemm16, mst_conv_dgrad, mst_conv_wgrad16); everything else and all
                   stored tensors fp32
  BatchNorm2d      batch statistics + running-stat update (mst_batchnorm_train), residual add and ReLU in the same pass
     backward      mst_batchnorm_bwd (ReLU mask first: mst_act_bwd on the saved output)
  max / avg pool   mst_maxpool_bwd_nhwc / mst_avgpool_bwd_nhwc
  slice fusion     mst.train.fusion_fwd / fusion_bwd with 16 heads over 512-wide tokens

BatchNorm in train mode normalises over ALL (B D) images of the step, so the step is not chunked: activations of the whole batch stay
resident (fp32 NHWC; about 60 MB per 224^2 image for resnet34 -- sized for 288 GB of HBM; the mixed mode keeps a 16-bit image of every
convolution input beside it).  Checked against torch.autograd of oracle/resnet_oracle.py on every parameter (tests/test_resnet_gpu.py).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import hip
from .models.resnet import _conv
from .train import _Grads, fusion_bwd, fusion_fwd


def _gemm_weight(conv, sum_in: bool) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> [Cout, Kpad] in (ky, kx, c) order (identical input channels summed when the gray volume was repeated)."""
    w = conv.weight.detach()
    if sum_in:
        w = w.sum(dim=1, keepdim=True)
    w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    K = w.shape[1]
    kpad = (K + 15) // 16 * 16
    if kpad != K:
        w = torch.cat([w, w.new_zeros(w.shape[0], kpad - K)], dim=1)
    return w.contiguous()


def _conv_bn_fwd(x: torch.Tensor, conv, bn, k: int, stride: int, pad: int, sum_in: bool, residual: Optional[torch.Tensor],
                 relu: bool, mp: Optional[torch.dtype] = None):
    """x [n,H,W,C] -> y [n,Ho,Wo,Cout] plus the record the backward needs.  mp (train_precision bf16 / fp16): the convolution and its
    two gradients on 16-bit MFMA operands (fp32 accumulation; activations, BatchNorm and everything stored stay fp32) --
    the reference's Trainer(precision='16-mixed') for F.conv2d."""
    n, H, W, Cin = x.shape
    wg = _gemm_weight(conv, sum_in)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    Cout = wg.shape[0]
    x16 = col16 = None
    if mp is not None and Cin % 64 == 0 and Cout % 64 == 0:
        x16 = hip.cvt16(x.view(-1, Cin), mp).view(x.shape)               # kept for the weight gradient (a third of the step's conversions otherwise)
        z = hip.conv_gemm16(x16, hip.cvt16(wg, mp), None, k, k, stride, pad, epilogue=hip.EPI_BIAS, out_dtype=torch.float32)
    elif mp is not None and Cout % 64 == 0 and k * k * Cin <= 64:
        # the stem (one input channel after the gray fold: 49 taps): its im2col rows, rounded on the way out and padded to 64 columns, are the
        # "pixels" of a 1 x 1 convolution -- for the forward and for the weight gradient (no fp32 im2col matrix, no strided fp32 GEMM)
        w64 = torch.zeros((Cout, 64), dtype=torch.float32, device=x.device)
        w64[:, :k * k * Cin] = wg[:, :k * k * Cin]
        col16 = hip.im2col_nhwc(x, k, k, stride, pad, 64, out_dtype=mp).view(n * Ho * Wo, 1, 1, 64)
        z = hip.conv_gemm16(col16, hip.cvt16(w64, mp), None, 1, 1, 1, 0, epilogue=hip.EPI_BIAS, out_dtype=torch.float32)
    else:
        mp = None
        z = _conv(x, wg, None, k, stride, pad, wg.shape[1], hip.EPI_BIAS)    # implicit GEMM behind the stem
    y, mean, rstd = hip.batchnorm_train(z, bn, residual, relu)
    bn.num_batches_tracked += 1
    rec = {"x": x, "z": z, "y": y if relu else None, "mean": mean, "rstd": rstd, "wg": wg, "k": k, "stride": stride, "pad": pad,
           "sum_in": sum_in, "relu": relu, "conv": conv, "bn": bn, "mp": mp, "x16": x16, "col16": col16}
    return y.view(n, Ho, Wo, wg.shape[0]), rec


def _split(n: int, hw: int):
    """Split the rows of one layer's im2col matrix for dW: (images, parts per image) with parts | hw, about 256 partial products."""
    s2 = 1
    while s2 < 16 and hw % (s2 * 2) == 0 and n * s2 < 256:
        s2 *= 2
    return n, s2


def _conv_bn_bwd(G: _Grads, rec, dy: torch.Tensor, need_dx: bool) -> Optional[torch.Tensor]:
    """dy [rows, Cout] = gradient of the unit's output (modified in place by the ReLU mask).  Returns dx [n,H,W,C] or None.  The
    caller routes the masked dy to the residual branch itself."""
    x, z, wg = rec["x"], rec["z"], rec["wg"]
    conv, bn = rec["conv"], rec["bn"]
    k, stride, pad = rec["k"], rec["stride"], rec["pad"]
    n, H, W, Cin = x.shape
    rows, Cout = z.shape
    kpad = wg.shape[1]
    dev = z.device
    if rec["relu"]:
        hip.act_bwd(rec["y"], dy, 1)
    dz, dg, db = hip.batchnorm_bwd(z, rec["mean"], rec["rstd"], bn.weight.detach(), dy)
    G.put(bn.weight, dg)
    G.put(bn.bias, db)
    implicit = os.environ.get("MST_CONV_IM2COL", "0") != "1"
    col = None
    mp = rec["mp"]
    dz16 = hip.cvt16(dz, mp) if mp is not None else None                # shared by the two gradients
    if mp is not None and rec["col16"] is not None:                      # the stem under mixed precision: d weight over its 16-bit im2col "pixels"
        dwg = torch.zeros((Cout, kpad), dtype=torch.float32, device=dev)
        dwg[:, :min(kpad, 64)] = hip.conv_wgrad(dz16, rec["col16"], 1, 1, 0)[:, :min(kpad, 64)]
    elif implicit and mp is not None and rec["x16"] is not None:
        dwg = hip.conv_wgrad(dz16, rec["x16"], k, stride, pad)           # 16-bit operands (the forward's image of x), fp32 partial products
    elif implicit and Cin % 64 == 0 and Cout % 4 == 0:
        dwg = hip.conv_wgrad(dz, x, k, stride, pad)                      # implicit GEMM: no im2col matrix
    else:
        col = hip.im2col_nhwc(x, k, k, stride, pad, kpad)
        hw = rows // n
        s1, s2 = _split(n, hw)
        part = torch.empty((s1 * s2, Cout * kpad), dtype=torch.float32, device=dev)
        ch = hw // s2                                                    # rows per partial product
        hip.gemm_ex(dz, col, part, Cout, kpad, ch, sa=(1, Cout), sb=(kpad, 1), sc=(kpad, 1), nb=(s1, s2), ba=(hw * Cout, ch * Cout),
                    bb=(hw * kpad, ch * kpad), bc=(s2 * Cout * kpad, Cout * kpad))
        dwg = hip.colsum(part, torch.zeros(Cout * kpad, dtype=torch.float32, device=dev)).view(Cout, kpad)
    K = k * k * Cin
    dw = dwg[:, :K].reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    if rec["sum_in"]:
        dw = dw.expand(Cout, conv.weight.shape[1], k, k)                  # w_eff = sum over the identical input channels
    G.put(conv.weight, dw.contiguous())
    if not need_dx:
        return None
    if Cout % 16 == 0 and Cin % 4 == 0 and stride in (1, 2) and implicit:
        # d input as a convolution of dz with the flipped, transposed weight (mst_conv_dgrad): no gradient matrix, no atomics
        del col
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        if mp is not None:
            return hip.conv_dgrad(dz16.view(n, Ho, Wo, Cout), hip.conv_dgrad_weight(conv.weight, mp), k, stride, pad, H, W)
        return hip.conv_dgrad(dz.view(n, Ho, Wo, Cout), hip.conv_dgrad_weight(conv.weight), k, stride, pad, H, W)
    if col is None:
        col = torch.empty((rows, kpad), dtype=torch.float32, device=dev)
    hip.gemm_ex(dz, wg, col, rows, kpad, Cout, sa=(Cout, 1), sb=(kpad, 1), sc=(kpad, 1))      # dcol overwrites col
    dx = torch.zeros_like(x)
    return hip.col2im_nhwc(col, dx, k, k, stride, pad)


def backbone_fwd(m, x_nhwc: torch.Tensor, sum_in: bool, mp: Optional[torch.dtype] = None):
    """torchvision resnet{18,34,50,101,152} forward in train mode up to the pooled features [n, 512 or 2048]."""
    sv = {"units": []}
    y, sv["stem"] = _conv_bn_fwd(x_nhwc.contiguous(), m.conv1, m.bn1, 7, 2, 3, sum_in, None, True, mp)
    sv["pool_in"] = y
    y = hip.maxpool_nhwc(y)
    for li in range(4):
        for blk in getattr(m, f"layer{li + 1}"):
            n, H, W, Cin = y.shape
            if hasattr(blk, "conv3"):                                    # bottleneck: 1x1 -> 3x3 (stride) -> 1x1 + residual
                h0, r0 = _conv_bn_fwd(y, blk.conv1, blk.bn1, 1, 1, 0, False, None, True, mp)
                h1, r1 = _conv_bn_fwd(h0, blk.conv2, blk.bn2, 3, blk.stride, 1, False, None, True, mp)
                rd = None
                if hasattr(blk, "downsample"):
                    idt, rd = _conv_bn_fwd(y, blk.downsample[0], blk.downsample[1], 1, blk.stride, 0, False, None, False, mp)
                    idt = idt.reshape(-1, idt.shape[-1])
                else:
                    idt = y.reshape(n * H * W, Cin)
                y, r2 = _conv_bn_fwd(h1, blk.conv3, blk.bn3, 1, 1, 0, False, idt, True, mp)
                sv["units"].append((r0, r2, rd, r1))
                continue
            h1, r1 = _conv_bn_fwd(y, blk.conv1, blk.bn1, 3, blk.stride, 1, False, None, True, mp)
            rd = None
            if hasattr(blk, "downsample"):
                idt, rd = _conv_bn_fwd(y, blk.downsample[0], blk.downsample[1], 1, blk.stride, 0, False, None, False, mp)
                idt = idt.reshape(-1, idt.shape[-1])
            else:
                idt = y.reshape(n * H * W, Cin)
            y, r2 = _conv_bn_fwd(h1, blk.conv2, blk.bn2, 3, 1, 1, False, idt, True, mp)
            sv["units"].append((r1, r2, rd, None))
    sv["last"] = y
    return hip.avgpool_nhwc(y), sv


def backbone_bwd(G: _Grads, sv, dfeat: torch.Tensor):
    y = sv["last"]
    n, H, W, Cc = y.shape
    dy = hip.avgpool_bwd_nhwc(dfeat.contiguous(), H * W).view(n * H * W, Cc)
    for r1, r2, rd, rmid in reversed(sv["units"]):
        dh1 = _conv_bn_bwd(G, r2, dy, True)                               # dy now carries the ReLU mask of the block output
        if rmid is not None:                                              # bottleneck: through the 3x3 unit to the first 1x1's output
            dh1 = _conv_bn_bwd(G, rmid, dh1.view(-1, dh1.shape[-1]), True)
        xin = r1["x"]
        if rd is not None:
            dx = _conv_bn_bwd(G, rd, dy.clone(), True)
        else:
            dx = dy.view(xin.shape).clone()
        d1 = _conv_bn_bwd(G, r1, dh1.view(-1, dh1.shape[-1]), True)
        hip.axpby_cols(d1.view(1, -1), dx.view(1, -1))
        dy = dx.view(-1, dx.shape[-1])
    dstem = hip.maxpool_bwd_nhwc(sv["pool_in"], dy.view(sv["units"][0][0]["x"].shape))
    _conv_bn_bwd(G, sv["stem"], dstem.view(-1, dstem.shape[-1]), False)


# ---- whole models ------------------------------------------------------------------------------------------------------
def forward_train(model, x_nhwc: torch.Tensor, sum_in: bool, B: Optional[int], D: Optional[int], mask: Optional[torch.Tensor]):
    """B/D given: ResNetSliceTrans (features -> slice transformer -> linear); else plain ResNet (features -> fc)."""
    feat, sv = backbone_fwd(model.model, x_nhwc, sum_in, {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[getattr(model, "train_precision", "fp32")])
    sv["feat"] = feat
    if B is None:
        fc = model.model.fc
        if isinstance(fc, nn.Identity):
            return feat, sv
        return hip.gemm(feat, fc.weight.detach(), fc.bias.detach()), sv
    from .models.resnet import SLICE_HEADS
    e = model.emb_ch
    fused, sv["fusion"] = fusion_fwd(model, feat, B, D, e, SLICE_HEADS, mask)
    sv["fused"], sv["B"], sv["D"] = fused, B, D
    return hip.gemm(fused, model.linear.weight.detach(), model.linear.bias.detach()), sv


def backward_train(model, sv, dout: torch.Tensor):
    G = _Grads()
    if "fusion" in sv:
        from .models.resnet import SLICE_HEADS
        dfused = G.lin_bwd(dout, sv["fused"], model.linear)
        dfeat = fusion_bwd(G, model, sv["fusion"], dfused, sv["B"], sv["D"], model.emb_ch, SLICE_HEADS)
    else:
        fc = model.model.fc
        dfeat = dout if isinstance(fc, nn.Identity) else G.lin_bwd(dout, sv["feat"], fc)
    backbone_bwd(G, sv, dfeat)
    return G.by_param


class _ResNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x_nhwc, sum_in, B, D, mask, *params):
        with torch.no_grad():
            out, saved = forward_train(model, x_nhwc, sum_in, B, D, mask)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            grads = backward_train(ctx.model, ctx.saved, dout.contiguous().float())
        ctx.saved = None
        out: List[Optional[torch.Tensor]] = []
        for p, need in zip(ctx.params, ctx.needs_input_grad[6:]):
            out.append(grads.get(id(p)) if need else None)
        return (None, None, None, None, None, None, *out)


def forward_with_grad(model, x_nhwc, sum_in: bool, B=None, D=None, mask=None):
    model._invalidate()                                  # a training step follows: the BatchNorm-folded inference weights are stale after it
    params = [p for p in model.parameters()]
    for p in params:
        if p.dtype != torch.float32 or p.device.type != "cuda":
            raise RuntimeError("training step: parameters must be fp32 on the MI355X (model.float().cuda()); there is no CPU fallback")
    return _ResNetFunction.apply(model, x_nhwc, sum_in, B, D, mask, *params)
