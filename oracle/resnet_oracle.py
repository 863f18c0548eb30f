"""CPU restatement of the reference's ResNet models (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product).

mst/models/resnet.py:27-243 builds on torchvision's ``resnet{18,34}`` (``weights="DEFAULT"``, resnet.py:44-45), whose source is not
part of the reference tree and which is not installed here: the published architecture (He et al. 2015; torchvision 0.19.1
torchvision/models/resnet.py -- conv7x7/2 + BN + ReLU + maxpool3x3/2, BasicBlock stages [3,4,6,3] x (64,128,256,512), downsample =
conv1x1/stride + BN on the first block of stages 2-4, adaptive average pool, fc) is restated with plain torch ops over a
``state_dict``.  **Parity UNPINNED for the backbone** (no reference code or fixture can be run for it); the across-slice half
(resnet.py:146-191) reuses ``mst_oracle.slice_fusion`` with 16 heads and IS pinned by tests/golden/resnet_fusion.npz, which the
reference's own TransformerEncoderLayer produced."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import mst_oracle as O

SD = Dict[str, torch.Tensor]
_LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


_TRAIN = False


def _bn(sd: SD, k: str, x: torch.Tensor) -> torch.Tensor:
    """nn.BatchNorm2d: eval mode = running statistics; train mode = batch statistics, running statistics of `sd` updated in
    place (momentum 0.1, unbiased variance), num_batches_tracked + 1."""
    if _TRAIN:
        if k + ".num_batches_tracked" in sd:
            sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], False, 0.0, 1e-5)


def resnet_features(sd: SD, x: torch.Tensor, model: int = 34, prefix: str = "model.", train: bool = False,
                    last: Optional[list] = None) -> torch.Tensor:
    """torchvision ResNet._forward_impl up to (and including) avgpool + flatten, then ``fc`` when the state dict has one.
    train: BatchNorm in train mode (the training step).  last: a list that receives the last ReLU output (for Grad-CAM++)."""
    global _TRAIN
    _TRAIN = train
    try:
        return _resnet_features(sd, x, model, prefix, last)
    finally:
        _TRAIN = False


def _resnet_features(sd: SD, x: torch.Tensor, model: int, prefix: str, last: Optional[list]) -> torch.Tensor:
    p = prefix
    y = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(_LAYERS[model]):
        for b in range(n):
            q = f"{p}layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            idt = y
            if q + ".conv3.weight" in sd:                # torchvision Bottleneck (v1.5): 1x1 -> 3x3 (carries the stride) -> 1x1 (x4)
                out = F.relu(_bn(sd, q + ".bn1", F.conv2d(y, sd[q + ".conv1.weight"])))
                out = F.relu(_bn(sd, q + ".bn2", F.conv2d(out, sd[q + ".conv2.weight"], stride=stride, padding=1)))
                out = _bn(sd, q + ".bn3", F.conv2d(out, sd[q + ".conv3.weight"]))
            else:                                        # BasicBlock
                out = F.relu(_bn(sd, q + ".bn1", F.conv2d(y, sd[q + ".conv1.weight"], stride=stride, padding=1)))
                out = _bn(sd, q + ".bn2", F.conv2d(out, sd[q + ".conv2.weight"], stride=1, padding=1))
            if q + ".downsample.0.weight" in sd:
                idt = _bn(sd, q + ".downsample.1", F.conv2d(y, sd[q + ".downsample.0.weight"], stride=stride))
            y = F.relu(out + idt)
    if last is not None:
        last.append(y)
    y = F.adaptive_avg_pool2d(y, 1).flatten(1)
    if p + "fc.weight" in sd:
        y = F.linear(y, sd[p + "fc.weight"], sd[p + "fc.bias"])
    return y


def fuse(sd: SD, emb: torch.Tensor, B: int, D: int, src_key_padding_mask: Optional[torch.Tensor] = None):
    """ResNetSliceTrans.forward after the backbone (resnet.py:180-191): cls concat, 16-head Slice Transformer, row 0, linear."""
    x = torch.cat([sd["cls_token"].repeat(B, 1, 1), emb.reshape(B, D, -1)], dim=1)
    m = None
    if src_key_padding_mask is not None:
        m = torch.cat([torch.zeros(B, 1, dtype=torch.bool), src_key_padding_mask.bool()], dim=1)
    y, probs = O.slice_fusion(sd, x, m, None, heads=16)
    feat = y[:, 0]
    return {"features": feat, "logits": F.linear(feat, sd["linear.weight"], sd["linear.bias"]), "slice_map": probs}


def gradcampp_last(sd: SD, x: torch.Tensor, model: int = 34) -> torch.Tensor:
    """ResNet.forward(save_attn=True) for the last ReLU (resnet.py:55-118): loss = sum of each image's largest output, gradient at
    the last ReLU output by torch.autograd, Grad-CAM++ weights (eq. 19), relu, global min / max normalisation -> [N, 1, h, w]."""
    last: list = []
    with torch.enable_grad():
        out = resnet_features({k: v.detach() for k, v in sd.items()}, x.detach().requires_grad_(True), model, last=last)
        act = last[0]
        loss = out.gather(1, out.argmax(dim=1, keepdim=True)).sum()
        (grads,) = torch.autograd.grad(loss, act)
    act = act.detach()
    g2 = grads ** 2
    g3 = g2 * grads
    denom = 2 * g2 + act.sum(dim=(2, 3), keepdim=True) * g3 + 1e-6
    denom = torch.where(denom != 0.0, denom, torch.ones_like(denom))
    weights = (F.relu(grads) * (g2 / denom)).sum(dim=(2, 3), keepdim=True)
    cam = F.relu((weights * act).sum(dim=1, keepdim=True))
    cam = cam - cam.min()
    return cam / cam.max()


def forward_slice_trans(sd: SD, source: torch.Tensor, src_key_padding_mask: Optional[torch.Tensor] = None, model: int = 34,
                        train: bool = False):
    """ResNetSliceTrans.forward (resnet.py:168-191): gray -> RGB repeat, per-slice resnet features, Slice Transformer, head."""
    B, C, D, H, W = source.shape
    x = source.repeat(1, 3, 1, 1, 1).permute(0, 2, 1, 3, 4).reshape(B * D, 3 * C, H, W)      # 'b c d h w -> (b d) c h w'
    emb = resnet_features(sd, x, model, train=train)
    out = fuse(sd, emb, B, D, src_key_padding_mask)
    out["emb"] = emb
    return out
