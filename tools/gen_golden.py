#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Run:  python tools/gen_golden.py [--only NAME ...]

The reference (/root/reference, read-only) is imported in place with the loader recipe of
SURVEY.md section 8c: synthetic parent packages whose ``__path__`` points at the reference
directories (so its ``__init__`` files, which pull in absent optional deps, never run) and two
minimal stand-in modules for the absent ``pytorch_lightning`` / ``torchmetrics`` bases, which carry
no arithmetic.  Weights and inputs come from ``mst.synth`` (build-owned hash PRNG), so the GPU box
can regenerate them bit-exactly without the reference.  Only inputs' seeds/shapes and the
reference's OUTPUTS are stored; no reference source is copied.
"""
from __future__ import annotations

import argparse
import importlib
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "new-vit_amd"))
GOLD = ROOT / "tests" / "golden"


def load_reference():
    import mst as build_mst  # the build's package: we need mst.synth before shadowing
    from mst import synth
    for k in [k for k in sys.modules if k == "mst" or k.startswith("mst.")]:
        del sys.modules[k]

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m

    pkg("refmst", REF / "mst")
    pkg("refmst.models", REF / "mst/models")
    pkg("refmst.models.utils", REF / "mst/models/utils")
    pkg("refmst.models.extern", REF / "mst/models/extern")
    pkg("refmst.models.extern.dinov2", REF / "mst/models/extern/dinov2")

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl
    tm = types.ModuleType("torchmetrics")

    class _Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def update(self, *a, **k):
            pass

        def compute(self):
            return torch.tensor(0.0)

        def reset(self):
            pass

    tm.AUROC = tm.Accuracy = tm.MeanSquaredError = _Metric
    sys.modules["torchmetrics"] = tm
    dino = importlib.import_module("refmst.models.dino")
    return dino, synth


def np_(t):
    return t.detach().cpu().float().clone().numpy()


def build(dino, synth, seed, **kw):
    """Reference model (pretrained=False) filled with synthetic weights."""
    rot = kw.pop("rotary", None)
    model = dino.DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False,
                                       rotary_positional_encoding=rot, **kw)
    sd = synth.synth_state_dict(kw.get("model_size", "s"), seed,
                                use_bottleneck=kw.get("use_bottleneck", False),
                                use_slice_pos_emb=kw.get("use_slice_pos_emb", False),
                                slice_fusion=kw.get("slice_fusion", "transformer"), rotary=rot)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("auc_roc" in m or "acc." in m) for m in missing), missing
    return model.eval(), sd


@torch.no_grad()
def case_end2end(dino, synth, name, shape, seed, *, mask=None, plane_subset=None, chunk=None, **kw):
    model, sd = build(dino, synth, seed, **kw)
    src = synth.synth_volume(shape, seed + 100)
    B, _, D, H, W = shape
    out = {"seed": seed, "shape": np.array(shape)}
    m = None
    if mask is not None:
        m = torch.zeros(B, D, dtype=torch.bool)
        for b, cnt in enumerate(mask):
            if cnt:
                m[b, D - cnt:] = True
        out["src_key_padding_mask"] = m.numpy()
    fusion = kw.get("slice_fusion", "transformer")
    if chunk is None:
        logits = model(src, src_key_padding_mask=m)
        out["logits"] = np_(logits)
        out["features"] = np_(model(src, src_key_padding_mask=m, without_linear=True))
        emb = model.encoder(src.reshape(B * D, 1, H, W).repeat(1, 3, 1, 1))
        out["emb"] = np_(emb)
        if fusion == "transformer":
            logits2 = model(src, save_attn=True, src_key_padding_mask=m)
            assert torch.allclose(logits, logits2, atol=1e-5)
            out["vit_cls_rows"] = np.stack([np_(a[:, :, 0]) for a in model.attention_maps])  # [12,n,h,N]
            out["slice_map"] = np_(model.attention_maps_slice[-1])
            out["plane_attention"] = np_(model.get_plane_attention())
            out["slice_attention"] = np_(model.get_slice_attention())
            # getters normalise in place: call on a fresh forward for the product (dino.py:197-202)
            model(src, save_attn=True, src_key_padding_mask=m)
            out["attention_maps"] = np_(model.get_attention_maps())
    else:
        # big shapes: encoder per chunk of slices with the reference's own hooks, then the reference
        # slice transformer on the concatenated embeddings (identical maths: slices are independent)
        x = src.reshape(B * D, H, W)
        embs, cls_rows = [], []
        for s0 in range(0, B * D, chunk):
            model.attention_maps, model.attention_maps_slice, model.hooks = [], [], []
            model.register_hooks()
            e = model.encoder(x[s0:s0 + chunk, None].repeat(1, 3, 1, 1))
            model.deregister_hooks()
            embs.append(e)
            cls_rows.append(model.attention_maps[-1][:, :, 0].clone())
            model.attention_maps = []
        emb = torch.cat(embs)
        last = torch.cat(cls_rows)                                  # [n,h,N]
        out["emb"] = np_(emb)
        xs = torch.cat([model.cls_token.repeat(B, 1, 1), emb.reshape(B, D, -1)], dim=1)
        model.attention_maps_slice, model.hooks = [], []
        fast = torch.backends.mha.get_fastpath_enabled()
        torch.backends.mha.set_fastpath_enabled(False)
        model.register_hooks()
        y = model.slice_fusion(xs)[:, 0]
        model.deregister_hooks()
        torch.backends.mha.set_fastpath_enabled(fast)
        out["features"] = np_(y)
        out["logits"] = np_(model.linear(y))
        out["slice_map"] = np_(model.attention_maps_slice[-1])
        model.attention_maps = [last[:, :, None, :]]
        plane = model.get_plane_attention()
        sl = model.get_slice_attention()
        out["slice_attention"] = np_(sl)
        sub = plane_subset or list(range(B * D))
        out["plane_subset"] = np.array(sub)
        out["plane_attention"] = np_(plane[sub])
        out["attention_maps"] = np_((sl * plane)[sub])
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


@torch.no_grad()
def case_ops(dino, synth):
    """Per-op known answers at tiny dims from the reference classes themselves."""
    vt = importlib.import_module("refmst.models.extern.dinov2.vision_transformer")
    layers = importlib.import_module("refmst.models.extern.dinov2.layers")
    tb = importlib.import_module("refmst.models.utils.transformer_blocks")
    out = {}
    # small ViT: E=64, depth 2, 2 heads, img 56 (4x4 grid), plus interpolated pos-embed at 84x70
    vit = vt.DinoVisionTransformer(img_size=56, patch_size=14, embed_dim=64, depth=2, num_heads=2,
                                   block_fn=lambda **k: layers.NestedTensorBlock(attn_class=layers.MemEffAttention, **k))
    sd = {}
    for i, (k, v) in enumerate(vit.state_dict().items()):
        sd[k] = torch.from_numpy(synth.hash_normal(tuple(v.shape), 7, 1000 + i)) * (0.3 if v.dim() > 1 else 0.2)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            sd[k] = sd[k] + 1.0
    vit.load_state_dict(sd)
    vit.eval()
    for k, v in sd.items():
        out["vit_sd." + k] = np_(v)
    for tag, hw in (("56", (56, 56)), ("84x70", (84, 70)), ("70x84", (70, 84))):
        x = torch.from_numpy(synth.hash_normal((3, hw[0], hw[1]), 8, 5))
        out[f"vit_in_{tag}"] = np_(x)
        out[f"vit_out_{tag}"] = np_(vit(x[:, None].repeat(1, 3, 1, 1)))
        out[f"vit_pos_{tag}"] = np_(vit.interpolate_pos_encoding(torch.zeros(1, (hw[0] // 14) * (hw[1] // 14) + 1, 64), hw[0], hw[1]))
    # pos-embed interpolation of the real geometry: 16x16 grid (224) -> 37x37, 36x36, 37x16
    vs = vt.vit_small(patch_size=14)
    pe = torch.from_numpy(synth.hash_normal((1, 257, 384), 9, 1)) * 0.2
    vs.pos_embed.data.copy_(pe)
    for tag, hw in (("518", (518, 518)), ("504", (504, 504)), ("518x224", (518, 224))):
        n = (hw[0] // 14) * (hw[1] // 14) + 1
        out[f"pos224_to_{tag}"] = np_(vs.interpolate_pos_encoding(torch.zeros(1, n, 384), hw[0], hw[1]))
    # slice transformer layer: d_model 48, 12 heads (hd 4), masks, RoPE, both need_weights paths
    for rot in (None, "RoPE"):
        tag = "rope" if rot else "plain"
        layer = tb.TransformerEncoderLayer(d_model=48, nhead=12, dim_feedforward=48, dropout=0.0,
                                           batch_first=True, norm_first=True, rotary_positional_encoding=rot)
        enc = nn.TransformerEncoder(layer, num_layers=1, norm=nn.LayerNorm(48)).eval()
        sdl = {}
        for i, (k, v) in enumerate(enc.state_dict().items()):
            if k.endswith("freqs"):
                sdl[k] = v.clone()
                continue
            sdl[k] = torch.from_numpy(synth.hash_normal(tuple(v.shape), 11, 2000 + i)) * (0.35 if v.dim() > 1 else 0.2)
            if "norm" in k and k.endswith("weight"):
                sdl[k] = sdl[k] + 1.0
        enc.load_state_dict(sdl)
        for k, v in sdl.items():
            out[f"tel_{tag}_sd.{k}"] = np_(v)
        x = torch.from_numpy(synth.hash_normal((2, 9, 48), 12, 3))
        mask = torch.zeros(2, 9, dtype=torch.bool)
        mask[1, 6:] = True
        out[f"tel_{tag}_in"] = np_(x)
        out[f"tel_{tag}_mask"] = mask.numpy()
        out[f"tel_{tag}_out"] = np_(enc(x))
        out[f"tel_{tag}_out_masked"] = np_(enc(x, src_key_padding_mask=mask))
        mha = enc.layers[0].self_attn
        y = enc.layers[0].norm1(x)
        _, w = mha(y, y, y, need_weights=True, average_attn_weights=False)
        out[f"tel_{tag}_weights"] = np_(w)
        _, wm = mha(y, y, y, need_weights=True, average_attn_weights=False, key_padding_mask=mask)
        out[f"tel_{tag}_weights_masked"] = np_(wm)
    np.savez_compressed(GOLD / "ops.npz", **out)
    print("ops", len(out), "arrays")


def case_errors(dino, synth):
    res = {}
    model, _ = build(dino, synth, 0)
    try:
        with torch.no_grad():
            model(torch.zeros(1, 1, 2, 512, 512))
    except AssertionError as e:
        res["512x512"] = {"type": "AssertionError", "message": str(e)}
    try:
        with torch.no_grad():
            model(torch.zeros(1, 1, 2, 224, 230))
    except AssertionError as e:
        res["224x230"] = {"type": "AssertionError", "message": str(e)}
    try:
        dino.DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, model_size="l")
    except AssertionError as e:
        res["model_size_l"] = {"type": "AssertionError", "message": str(e)}
    lm, _ = build(dino, synth, 9, rotary="LiRE")
    for tag, shape in (("liere_batch2", (2, 1, 32, 28, 28)), ("liere_d16", (1, 1, 16, 28, 28))):
        try:
            with torch.no_grad():
                lm(torch.zeros(*shape))
        except RuntimeError as e:
            res[tag] = {"type": "RuntimeError", "message": str(e).splitlines()[0][:120]}
    (GOLD / "errors.json").write_text(json.dumps(res, indent=1))
    print(res)


@torch.no_grad()
def case_rollout(dino, synth, name, shape, seed):
    """get_attention_cls (dino.py:204-212) on the reference's own stored full maps."""
    model, sd = build(dino, synth, seed)
    src = synth.synth_volume(shape, seed + 100)
    out = {"seed": seed, "shape": np.array(shape)}
    out["logits"] = np_(model(src, save_attn=True))
    out["vit_full_first"] = np_(model.attention_maps[0])
    out["vit_full_last"] = np_(model.attention_maps[-1])
    out["attention_cls"] = np_(model.get_attention_cls())
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


@torch.no_grad()
def case_saliency(dino, synth, name, shape, seed, use_tta):
    """`--get_attention` volume: the REFERENCE model's forward + getters, followed by the torch calls of
    scripts/main_predict.py run_pred/_pred_trans (l.72-105, 147-165), restated here because the script itself cannot be
    imported (torchio, monai, torchvision, seaborn absent).  F.interpolate is torch's own."""
    import torch.nn.functional as F
    model, sd = build(dino, synth, seed)
    source = synth.synth_volume(shape, seed + 100)

    def pred_trans(src):                                       # _pred_trans, DinoV2 branch
        pred = torch.softmax(model(src, save_attn=True), dim=-1)
        weight = model.get_attention_maps().mean(dim=1)
        g = int(weight.shape[-1] ** 0.5)
        weight = weight[:, :g * g].view(1, 1, src.shape[2], g, g)
        ws = model.get_slice_attention().mean(dim=1).view(1, 1, -1, 1, 1) * torch.ones_like(src)
        return pred, weight, ws

    pred, weight, ws = pred_trans(source)
    if use_tta:                                                # run_pred l.147-158
        for dims in [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]:
            p_i, w_i, ws_i = pred_trans(torch.flip(source, dims))
            pred, weight, ws = pred + p_i, weight + torch.flip(w_i, dims), ws + torch.flip(ws_i, dims)
        pred, weight, ws = pred / 8, weight / 8, ws / 8
    weight = F.interpolate(weight, size=source.shape[2:], mode="trilinear")   # l.163
    out = {"seed": seed, "shape": np.array(shape), "use_tta": int(use_tta), "pred": np_(pred), "weight": np_(weight),
           "weight_slice_per_slice": np_(ws[0, 0, :, 0, 0]),
           "weight_slice_is_broadcast": int(bool((ws == ws[:, :, :, :1, :1]).all()))}
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


@torch.no_grad()
def case_multichannel(dino, synth, name, shape, seed):
    """C = 3 input: channels become extra slices, channel fastest (dino.py:125); key-padding mask over D*C positions."""
    model, sd = build(dino, synth, seed)
    src = synth.synth_volume(shape, seed + 100)
    B, C, D, H, W = shape
    m = torch.zeros(B, D * C, dtype=torch.bool)
    m[1, -2:] = True
    out = {"seed": seed, "shape": np.array(shape), "src_key_padding_mask": m.numpy()}
    out["logits"] = np_(model(src, src_key_padding_mask=m, save_attn=True))
    out["slice_attention"] = np_(model.get_slice_attention())
    model(src, src_key_padding_mask=m, save_attn=True)
    out["attention_maps"] = np_(model.get_attention_maps())
    out["features"] = np_(model(src, src_key_padding_mask=m, without_linear=True))
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


@torch.no_grad()
def case_hub_reg(synth, name, seed):
    """The encoder as torch.hub's dinov2_vits14_reg configures the vendored class (facebookresearch/dinov2 hub/backbones.py:
    img_size 518, init_values 1.0, block_chunks 0, 4 register tokens, interpolate_antialias=True, interpolate_offset=0.0),
    filled with synthetic weights: registers + LayerScale + anti-aliased size-based position resampling."""
    vt = importlib.import_module("refmst.models.extern.dinov2.vision_transformer")
    enc = vt.vit_small(patch_size=14, img_size=518, init_values=1.0, block_chunks=0, num_register_tokens=4,
                       interpolate_antialias=True, interpolate_offset=0.0).eval()
    sd = synth.synth_state_dict("s", seed, img_size=518, layerscale=True, chunked=False, num_register_tokens=4)
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    missing, unexpected = enc.load_state_dict(enc_sd, strict=False)
    assert not unexpected and all("mask_token" in m for m in missing), (missing, unexpected)
    out = {"seed": seed}
    for tag, shape in (("224", (3, 224, 224)), ("112x140", (2, 112, 140)), ("518", (1, 518, 518))):
        x = synth.synth_volume((1, 1) + shape, seed + 100)[0, 0]                 # [n, H, W] slices
        out[f"emb_{tag}"] = np_(enc(x[:, None].repeat(1, 3, 1, 1)))
        out[f"shape_{tag}"] = np.array(shape)
    out["pos_16x16"] = np_(enc.interpolate_pos_encoding(torch.zeros(1, 257, 384), 224, 224))
    out["pos_8x10"] = np_(enc.interpolate_pos_encoding(torch.zeros(1, 81, 384), 112, 140))
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


def case_resnet_fusion(dino, synth):
    """The across-slice half of ResNetSliceTrans (reference resnet.py:146-166,180-191) built from the reference's own
    TransformerEncoderLayer(d_model=512, nhead=16, dim_feedforward=512, norm_first) + final LayerNorm, cls_token, linear head,
    on synthetic slice embeddings.  (The class itself cannot be constructed here: torchvision / MONAI are absent.)"""
    tb = importlib.import_module("refmst.models.utils.transformer_blocks")
    sd = synth.synth_resnet_state_dict(21, 34, 2)
    layer = tb.TransformerEncoderLayer(d_model=512, nhead=16, dim_feedforward=512, dropout=0.0, batch_first=True, norm_first=True,
                                       rotary_positional_encoding=None)
    enc = nn.TransformerEncoder(layer, num_layers=1, norm=nn.LayerNorm(512)).eval()
    enc.load_state_dict({k[len("slice_fusion."):]: v for k, v in sd.items() if k.startswith("slice_fusion.")})
    B, D = 2, 7
    emb = torch.from_numpy(synth.hash_normal((B, D, 512), 22, 4)) * 0.8
    mask = torch.zeros(B, D, dtype=torch.bool)
    mask[1, 4:] = True
    out = {"seed": np.array(21), "emb": np_(emb), "src_key_padding_mask": mask.numpy()}
    lin = nn.Linear(512, 2)
    lin.load_state_dict({"weight": sd["linear.weight"], "bias": sd["linear.bias"]})
    with torch.no_grad():
        for tag, m in (("", None), ("_masked", mask)):
            x = torch.cat([sd["cls_token"].repeat(B, 1, 1), emb], dim=1)                    # resnet.py:180
            mm = None if m is None else torch.cat([torch.zeros(B, 1, dtype=torch.bool), m], dim=1)
            y = enc(x, src_key_padding_mask=mm)[:, 0]                                       # resnet.py:187-188
            out["features" + tag] = np_(y)
            out["logits" + tag] = np_(lin(y))
            mha = enc.layers[0].self_attn
            yn = enc.layers[0].norm1(x)
            _, w = mha(yn, yn, yn, need_weights=True, average_attn_weights=False, key_padding_mask=mm)
            out["slice_map" + tag] = np_(w)
            a = w[:, :, 0, 1:].clone()                                                      # get_slice_attention, resnet.py:196-205
            a /= a.sum(dim=-1, keepdim=True)
            out["slice_attention" + tag] = np_(a.mean(dim=1).view(-1)[:, None, None])
    np.savez_compressed(GOLD / "resnet_fusion.npz", **out)
    print("wrote resnet_fusion.npz")


def case_slices2rgb(dino, synth):
    """slices2rgb (reference dino.py:10-27) on seed-generated volumes: D % 3 = 1, 0 and 2."""
    out = {}
    for i, shape in enumerate([(2, 1, 7, 6, 10), (1, 1, 9, 4, 4), (3, 1, 5, 3, 5)]):
        x = synth.synth_volume(shape, 300 + i)
        out[f"shape{i}"] = np.array(shape)
        out[f"seed{i}"] = np.array(300 + i)
        out[f"out{i}"] = dino.slices2rgb(x).numpy()
    np.savez_compressed(GOLD / "slices2rgb.npz", **out)
    print("wrote slices2rgb.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    torch.set_num_threads(8)
    dino, synth = load_reference()
    GOLD.mkdir(parents=True, exist_ok=True)
    cases = {
        "weights": lambda: (GOLD / "weights.json").write_text(json.dumps({
            "s_seed0": synth.state_dict_digest(synth.synth_state_dict("s", 0)),
            "s_seed1_hub518": synth.state_dict_digest(synth.synth_state_dict("s", 1, img_size=518, layerscale=True, chunked=False)),
            "volume_1x1x2x28x28_seed3": float(synth.synth_volume((1, 1, 2, 28, 28), 3).double().sum()),
        }, indent=1)),
        "ops": lambda: case_ops(dino, synth),
        "errors": lambda: case_errors(dino, synth),
        "c1_1x16x224": lambda: case_end2end(dino, synth, "c1_1x16x224", (1, 1, 16, 224, 224), 0),
        "b2_mask": lambda: case_end2end(dino, synth, "b2_mask", (2, 1, 6, 112, 140), 1, mask=[0, 2]),
        "bottleneck_pos": lambda: case_end2end(dino, synth, "bottleneck_pos", (1, 1, 5, 112, 112), 2,
                                               use_bottleneck=True, use_slice_pos_emb=True),
        "rope": lambda: case_end2end(dino, synth, "rope", (2, 1, 7, 112, 112), 3, rotary="RoPE", mask=[3, 0]),
        "average": lambda: case_end2end(dino, synth, "average", (2, 1, 4, 112, 112), 4, slice_fusion="average"),
        "linear32": lambda: case_end2end(dino, synth, "linear32", (1, 1, 32, 56, 56), 5, slice_fusion="linear"),
        "size_b": lambda: case_end2end(dino, synth, "size_b", (1, 1, 3, 112, 112), 6, model_size="b"),
        "c3_1x64x518": lambda: case_end2end(dino, synth, "c3_1x64x518", (1, 1, 64, 518, 518), 2, chunk=8,
                                            plane_subset=[0, 31, 63]),
        "s504_1x4x504": lambda: case_end2end(dino, synth, "s504_1x4x504", (1, 1, 4, 504, 504), 7, chunk=4,
                                             plane_subset=[0, 3]),
        "liere": lambda: case_end2end(dino, synth, "liere", (1, 1, 32, 56, 56), 9, rotary="LiRE", mask=[5]),
        "saliency_1x5x84": lambda: case_saliency(dino, synth, "saliency_1x5x84", (1, 1, 5, 84, 84), 10, False),
        "saliency_tta_1x4x56x84": lambda: case_saliency(dino, synth, "saliency_tta_1x4x56x84", (1, 1, 4, 56, 56), 11, True),
        "multichannel": lambda: case_multichannel(dino, synth, "multichannel", (2, 3, 2, 56, 70), 13),
        "hub_reg": lambda: case_hub_reg(synth, "hub_reg", 12),
        "rollout_1x3x84": lambda: case_rollout(dino, synth, "rollout_1x3x84", (1, 1, 3, 84, 84), 8),
        "slices2rgb": lambda: case_slices2rgb(dino, synth),
        "resnet_fusion": lambda: case_resnet_fusion(dino, synth),
    }
    for name, fn in cases.items():
        if args.only and name not in args.only:
            continue
        fn()


if __name__ == "__main__":
    main()
