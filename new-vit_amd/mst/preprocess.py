"""Device-side input pipeline: the two array transforms the reference's datasets apply to a volume before the
model -- ``CropOrPad`` (deterministic centre) and ``ZNormalization`` of mst/data/datasets/augmentations/augmentations_3d.py
(l.144-195, l.40-86; e.g. dataset_3d_duke.py:42-43) -- as HIP kernels behind the C ABI (mst_crop_or_pad, mst_znorm), so that a
volume can go loader -> HBM -> encoder without the host round trip.  Same argument meaning as the reference's classes; tensors are
``[C, D, H, W]`` (what ``ImageOrSubjectToTensor`` hands to the model), the spatial target in the same axis order."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple, Union

import torch

from . import hip


def crop_or_pad(x: torch.Tensor, target_shape: Sequence[int], padding_mode: Union[str, float] = 0) -> torch.Tensor:
    """CropOrPad(target_shape, padding_mode, random_center=False): centre crop / pad of every channel of ``x`` [C, a0, a1, a2].
    padding_mode: 'minimum' (numpy.pad 'minimum', the datasets' choice) or a number (constant)."""
    if x.dim() != 4:
        raise ValueError("crop_or_pad expects [C, D, H, W]")
    if not x.is_cuda:
        raise RuntimeError("crop_or_pad: tensor must live on a HIP device; there is no CPU path")
    if isinstance(padding_mode, str) and padding_mode != "minimum":
        raise NotImplementedError(f"padding_mode {padding_mode!r}: only 'minimum' and constants are built")
    x = x.float().contiguous()
    Cn, s0, s1, s2 = x.shape
    t = [int(s if v is None else v) for v, s in zip(target_shape, (s0, s1, s2))]
    out = torch.empty((Cn, *t), dtype=torch.float32, device=x.device)
    pn = [max(a, b) for a, b in zip((s0, s1, s2), t)]
    ws = torch.empty(pn[0] * pn[1] * pn[2], dtype=torch.float32, device=x.device)
    lib = hip.load()
    for c in range(Cn):
        hip._check(lib.mst_crop_or_pad(hip.ptr(x[c]), s0, s1, s2, hip.ptr(out[c]), t[0], t[1], t[2],
                                       1 if padding_mode == "minimum" else 0, 0.0 if padding_mode == "minimum" else float(padding_mode),
                                       hip.ptr(ws), ws.numel() * 4, hip.stream_of(x)), "mst_crop_or_pad")
    return out


class _ZState(C.Structure):
    _fields_ = [("mn", C.c_float), ("mx", C.c_float), ("count", C.c_ulonglong), ("prefix", C.c_uint * 4),
                ("rank", C.c_ulonglong * 4), ("hist", C.c_uint * 1024), ("cut_lo", C.c_float), ("cut_hi", C.c_float),
                ("sum", C.c_double), ("sq", C.c_double), ("mean", C.c_float), ("sd", C.c_float), ("zero_std", C.c_int)]


def znormalize(x: torch.Tensor, percentiles: Tuple[float, float] = (0, 100), return_stats: bool = False):
    """ZNormalization(percentiles, per_channel=True, per_slice=False, masking_method=lambda x: (x > x.min()) & (x < x.max())).
    x [1, D, H, W] (one channel: the mask's extrema are those of the whole image)."""
    if x.dim() != 4 or x.shape[0] != 1:
        raise NotImplementedError("znormalize: one channel [1, D, H, W] (multi-channel masks use the extrema of the whole image)")
    if not x.is_cuda:
        raise RuntimeError("znormalize: tensor must live on a HIP device; there is no CPU path")
    x = x.float().contiguous()
    lib = hip.load()
    nbytes = int(lib.mst_znorm_state_bytes())
    assert nbytes >= C.sizeof(_ZState)
    state = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    hip._check(lib.mst_znorm(hip.ptr(x), x.numel(), percentiles[0] / 100.0, percentiles[1] / 100.0, hip.ptr(out), hip.ptr(state),
                             hip.stream_of(x)), "mst_znorm")
    st = _ZState.from_buffer_copy(bytes(state[:C.sizeof(_ZState)].cpu().numpy()))      # one small read-back: the reference raises here too
    if st.zero_std:
        raise RuntimeError('Standard deviation is 0 for masked values in image')
    if return_stats:
        return out, {"cut_lo": st.cut_lo, "cut_hi": st.cut_hi, "mean": st.mean, "std": st.sd, "count": st.count}
    return out
