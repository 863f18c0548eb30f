"""Device-side `--get_attention` post-processing: a drop-in for ``run_pred`` of the reference's scripts/main_predict.py
(l.134-165) when the model is this package's DinoV2ClassifierSlice.

The reference runs, per volume and per test-time-augmentation flip, the forward, three getters, a head mean, flips and at
the end ``F.interpolate(..., mode='trilinear')`` as separate torch ops on [1,1,D,H,W] tensors.  Here the forward is the HIP
path and everything after it is two kernels of libmst_hip (mst_saliency_accumulate at the patch-grid resolution, ONE
mst_saliency_upsample); torch only flips the input volume and expands the per-slice vector (views, no arithmetic).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip
from .models.dino import DinoV2ClassifierSlice

TTA_FLIPS = [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]     # scripts/main_predict.py:148


def run_pred(model, batch, save_attn: bool = False, use_softmax: bool = True, use_tta: bool = False
             ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(pred [B,out], weight [1,1,D,H,W] | None, weight_slice [1,1,D,H,W] | None) exactly as the script's run_pred."""
    if not isinstance(model, DinoV2ClassifierSlice):
        raise TypeError("mst.saliency.run_pred drives this package's DinoV2ClassifierSlice only")
    source, mask = batch["source"], batch.get("src_key_padding_mask", None)
    source = source.to(model.device)
    if save_attn and source.shape[0] != 1:
        raise RuntimeError(f"shape '[1, 1, {source.shape[2]}, ...]' is invalid for a batch of {source.shape[0]} volumes "
                           "(main_predict.py:98 views the maps as one volume)")
    D, H, W = (int(v) for v in source.shape[2:])
    low = ws = None
    pred = None
    flips = [()] + (TTA_FLIPS if use_tta else [])
    for dims in flips:
        src = torch.flip(source, dims) if dims else source
        with torch.no_grad():
            p = model(src, src_key_padding_mask=mask, save_attn=save_attn)
        if use_softmax:
            p = torch.softmax(p, dim=-1)                    # l.63-64
        pred = p if pred is None else pred + p
        if save_attn:
            maps = model.get_attention_maps()              # [D, heads, Np]
            sa = model.get_slice_attention().reshape(-1)    # [D]
            g = int(maps.shape[-1] ** 0.5)                  # l.91: square patch grid assumed by the script
            if low is None:
                low = torch.empty((D, g, g), dtype=torch.float32, device=maps.device)
                ws = torch.empty((D,), dtype=torch.float32, device=maps.device)
            fm = sum(1 << (a - 2) for a in dims)
            hip.saliency_accumulate(maps.contiguous(), sa.contiguous(), g, g, fm, low, ws, accumulate=bool(dims))
    n = float(len(flips))
    pred = pred / n if use_tta else pred
    if not save_attn:
        return pred, None, None
    weight = hip.saliency_upsample(low, (D, H, W), scale=1.0 / n)[None, None]            # l.155-163
    weight_slice = (ws / n).view(1, 1, D, 1, 1).expand(1, 1, D, H, W)                     # l.102-103, 156
    return pred, weight, weight_slice
