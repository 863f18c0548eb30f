"""Deterministic synthetic weights and volumes for the MST-DINOv2 hot path.

There is no network in the build or bench environment, so neither the DINOv2 hub weights
(reference: mst/models/dino.py:59-63) nor the Zenodo checkpoints (reference README.md:30) are
available.  Parity fixtures, tests and bench.py therefore run on *synthetic* parameters with the
reference's exact ``state_dict`` keys and shapes (reference: mst/models/dino.py:52-103,
mst/models/extern/dinov2/vision_transformer.py:106-168, mst/models/utils/transformer_blocks.py:483-500).

The generator is a counter-based integer hash (splitmix64 finaliser) followed by Box-Muller, in
numpy only, so the tensors are bit-identical across torch versions and machines (``torch.manual_seed``
streams are not guaranteed to be).  Distributions are fan-in scaled normals with non-zero biases,
position embeddings and CLS tokens so that every term of the forward is exercised.
"""
from __future__ import annotations

import hashlib
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

_VIT = {  # reference: vision_transformer.py:340-365
    "s": dict(embed_dim=384, depth=12, num_heads=6),
    "b": dict(embed_dim=768, depth=12, num_heads=12),
}

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n doubles in (0, 1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0x100000001B3)
                           + np.uint64(stream))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def hash_normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Standard normals (float32) of the given shape via Box-Muller on hash_uniform."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = hash_uniform(m, seed, 2 * stream)
    u2 = hash_uniform(m, seed, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2.0 * math.pi * u2), r * np.sin(2.0 * math.pi * u2)])[:n]
    return z.astype(np.float32).reshape(shape)


def _key_stream(key: str) -> int:
    return zlib.crc32(key.encode()) & 0x7FFFFFFF


def _t(key, shape, seed, std=1.0, mean=0.0):
    return torch.from_numpy(hash_normal(tuple(shape), seed, _key_stream(key)) * np.float32(std)
                            + np.float32(mean))


def synth_state_dict(model_size: str = "s", seed: int = 0, *, img_size: int = 224,
                     patch_size: int = 14, out_ch: int = 2, use_bottleneck: bool = False,
                     use_slice_pos_emb: bool = False, slice_fusion: str = "transformer",
                     rotary: str | None = None, layerscale: bool = False, chunked: bool = True,
                     num_register_tokens: int = 0, enable_linear: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """A full ``DinoV2ClassifierSlice.state_dict()`` with the reference's key layout.

    ``chunked=True`` gives the ``pretrained=False`` layout (``encoder.blocks.0.<i>.*``, reference
    vision_transformer.py:153-160); ``chunked=False, layerscale=True, img_size=518`` gives the hub
    layout (``encoder.blocks.<i>.*`` with ``ls{1,2}.gamma``).
    """
    cfg = _VIT[model_size]
    E, depth = cfg["embed_dim"], cfg["depth"]
    g = img_size // patch_size
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def lin(prefix, out_f, in_f, gain=1.0, bias_std=0.05):
        sd[prefix + ".weight"] = _t(prefix + ".weight", (out_f, in_f), seed, gain / math.sqrt(in_f))
        sd[prefix + ".bias"] = _t(prefix + ".bias", (out_f,), seed, bias_std)

    def ln(prefix, dim):
        sd[prefix + ".weight"] = _t(prefix + ".weight", (dim,), seed, 0.1, 1.0)
        sd[prefix + ".bias"] = _t(prefix + ".bias", (dim,), seed, 0.1)

    sd["encoder.cls_token"] = _t("encoder.cls_token", (1, 1, E), seed, 0.5)
    sd["encoder.pos_embed"] = _t("encoder.pos_embed", (1, g * g + 1, E), seed, 0.2)
    if num_register_tokens:
        sd["encoder.register_tokens"] = _t("encoder.register_tokens", (1, num_register_tokens, E), seed, 0.5)
    sd["encoder.mask_token"] = _t("encoder.mask_token", (1, E), seed, 0.02)
    k = "encoder.patch_embed.proj"
    sd[k + ".weight"] = _t(k + ".weight", (E, 3, patch_size, patch_size), seed,
                           1.0 / math.sqrt(3 * patch_size * patch_size))
    sd[k + ".bias"] = _t(k + ".bias", (E,), seed, 0.05)
    for i in range(depth):
        p = f"encoder.blocks.0.{i}" if chunked else f"encoder.blocks.{i}"
        ln(p + ".norm1", E)
        lin(p + ".attn.qkv", 3 * E, E, gain=1.6)
        lin(p + ".attn.proj", E, E, gain=0.7)
        if layerscale:
            sd[p + ".ls1.gamma"] = _t(p + ".ls1.gamma", (E,), seed, 0.2, 0.8)
        ln(p + ".norm2", E)
        lin(p + ".mlp.fc1", 4 * E, E, gain=1.2)
        lin(p + ".mlp.fc2", E, 4 * E, gain=0.7)
        if layerscale:
            sd[p + ".ls2.gamma"] = _t(p + ".ls2.gamma", (E,), seed, 0.2, 0.8)
    ln("encoder.norm", E)

    emb = E
    if use_bottleneck:
        lin("bottleneck", E // 4, E)
        emb = E // 4
    if slice_fusion == "transformer":
        if use_slice_pos_emb:
            sd["slice_pos_emb.weight"] = _t("slice_pos_emb.weight", (256, emb), seed, 0.3)
        p = "slice_fusion.layers.0"
        sd[p + ".self_attn.in_proj_weight"] = _t(p + ".self_attn.in_proj_weight", (3 * emb, emb), seed,
                                                 2.0 / math.sqrt(emb))
        sd[p + ".self_attn.in_proj_bias"] = _t(p + ".self_attn.in_proj_bias", (3 * emb,), seed, 0.05)
        lin(p + ".self_attn.out_proj", emb, emb)
        if rotary == "RoPE":
            hd = emb // 12
            # reference: rotary_embedding_torch.py:105 with theta=256 (transformer_blocks.py:338)
            sd[p + ".self_attn.rotary_positional_encoding.freqs"] = (
                1.0 / (256 ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd)))
        elif rotary == "LiRE":
            # reference: AttentionLiereRotator(head_dim, liere_block_size=head_dim//2, spacial_dims=1, axes_length=33)
            # (transformer_blocks.py:350-357; rotary_embedding_torch.py:337-339).  std 0.02 keeps the generator
            # norm (sum over 33 positions weighted by the position index) at a few radians.
            hd = emb // 12
            blk = hd // 2
            for i in range(hd // blk):
                key = p + f".self_attn.rotary_positional_encoding.vars.{i}"
                sd[key] = _t(key, ((blk * blk - blk) // 2, 33, 1), seed, 0.02)
        lin(p + ".linear1", emb, emb)
        lin(p + ".linear2", emb, emb)
        ln(p + ".norm1", emb)
        ln(p + ".norm2", emb)
        ln("slice_fusion.norm", emb)
        sd["cls_token"] = _t("cls_token", (1, 1, emb), seed, 1.0)
    elif slice_fusion == "linear":
        emb = emb * 32
    if enable_linear:
        lin("linear", out_ch, emb, gain=1.0)
    return sd


def synth_volume(shape, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    """A z-normalised synthetic volume ``[B, 1, D, H, W]`` (what the datasets emit:
    reference mst/data/datasets/dataset_3d_duke.py:43), N(0,1) with a smooth low-frequency term so
    neighbouring patches are correlated like real slices."""
    x = hash_normal(tuple(shape), seed, 0x5EED)
    B, C, D, H, W = shape
    yy = np.linspace(-1.0, 1.0, H, dtype=np.float32)[:, None]
    xx = np.linspace(-1.0, 1.0, W, dtype=np.float32)[None, :]
    zz = np.linspace(-1.0, 1.0, D, dtype=np.float32)[:, None, None] if D > 1 else np.zeros((1, 1, 1), np.float32)
    smooth = np.cos(2.5 * yy + 1.5 * zz) * np.sin(3.0 * xx - zz)
    x = 0.8 * x + 0.6 * smooth[None, None]
    return torch.from_numpy(x.astype(np.float32)).to(dtype)


def state_dict_digest(sd) -> str:
    """SHA-256 over keys, shapes and raw bytes: pins the generator (tests/golden/weights.json)."""
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


_RESNET_LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def synth_resnet_state_dict(seed: int = 0, model: int = 34, out_ch: int = 2, slice_trans: bool = True,
                            fc_out: int | None = None) -> "OrderedDict[str, torch.Tensor]":
    """``ResNetSliceTrans.state_dict()`` (``slice_trans=True``: ``model.*`` = torchvision resnet layout with ``fc`` = Identity,
    plus ``slice_fusion.*``, ``cls_token``, ``linear.*``: reference resnet.py:146-166) or ``ResNet.state_dict()`` with
    ``model.fc`` = Linear(512, fc_out).  BatchNorm statistics are non-trivial so that the folding is exercised."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def conv(k, cout, cin, ks):
        sd[k + ".weight"] = _t(k + ".weight", (cout, cin, ks, ks), seed, math.sqrt(2.0 / (cin * ks * ks)))

    def bn(k, c):
        sd[k + ".weight"] = _t(k + ".weight", (c,), seed, 0.1, 1.0)
        sd[k + ".bias"] = _t(k + ".bias", (c,), seed, 0.1)
        sd[k + ".running_mean"] = _t(k + ".running_mean", (c,), seed, 0.2)
        sd[k + ".running_var"] = _t(k + ".running_var", (c,), seed, 0.2, 1.0).abs() + 0.1
        sd[k + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    conv("model.conv1", 64, 3, 7)
    bn("model.bn1", 64)
    cin = 64
    bottleneck = model >= 50                             # torchvision Bottleneck (v1.5: the stride sits on the 3x3 convolution), expansion 4
    for li, (n, w) in enumerate(zip(_RESNET_LAYERS[model], (64, 128, 256, 512))):
        for b in range(n):
            p = f"model.layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            if bottleneck:
                conv(p + ".conv1", w, cin, 1)
                bn(p + ".bn1", w)
                conv(p + ".conv2", w, w, 3)
                bn(p + ".bn2", w)
                conv(p + ".conv3", 4 * w, w, 1)
                bn(p + ".bn3", 4 * w)
                cout = 4 * w
            else:
                conv(p + ".conv1", w, cin, 3)
                bn(p + ".bn1", w)
                conv(p + ".conv2", w, w, 3)
                bn(p + ".bn2", w)
                cout = w
            if stride != 1 or cin != cout:
                conv(p + ".downsample.0", cout, cin, 1)
                bn(p + ".downsample.1", cout)
            cin = cout
    E = cin                                              # 512 (BasicBlock models) or 2048 (reference resnet.py:152)

    def lin(k, out_f, in_f, gain=1.0):
        sd[k + ".weight"] = _t(k + ".weight", (out_f, in_f), seed, gain / math.sqrt(in_f))
        sd[k + ".bias"] = _t(k + ".bias", (out_f,), seed, 0.05)

    def ln(k):
        sd[k + ".weight"] = _t(k + ".weight", (E,), seed, 0.1, 1.0)
        sd[k + ".bias"] = _t(k + ".bias", (E,), seed, 0.1)

    if not slice_trans:
        if fc_out is not None:
            lin("model.fc", fc_out, E)
        return sd
    p = "slice_fusion.layers.0"
    sd[p + ".self_attn.in_proj_weight"] = _t(p + ".self_attn.in_proj_weight", (3 * E, E), seed, 1.2 / math.sqrt(E))
    sd[p + ".self_attn.in_proj_bias"] = _t(p + ".self_attn.in_proj_bias", (3 * E,), seed, 0.05)
    lin(p + ".self_attn.out_proj", E, E, 0.8)
    lin(p + ".linear1", E, E)
    lin(p + ".linear2", E, E, 0.8)
    ln(p + ".norm1")
    ln(p + ".norm2")
    ln("slice_fusion.norm")
    sd["cls_token"] = _t("cls_token", (1, 1, E), seed, 0.5)
    lin("linear", out_ch, E)
    return sd
