"""Pins oracle/mst_oracle.py to the fixtures the REFERENCE produced (tools/gen_golden.py).

CPU only.  Tolerances: the reference's own run-to-run / thread-count noise floor is 2.4e-7
(SURVEY.md 8c), so 2e-5 absolute on O(1) quantities and 1e-4 relative on attention maps."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_l2
from mst import synth
from oracle import mst_oracle as O

CASES = {  # name -> synth/forward kwargs  (must mirror tools/gen_golden.py)
    "c1_1x16x224": dict(),
    "b2_mask": dict(),
    "bottleneck_pos": dict(use_bottleneck=True, use_slice_pos_emb=True),
    "rope": dict(rotary="RoPE"),
    "liere": dict(rotary="LiRE"),
    "average": dict(slice_fusion="average"),
    "linear32": dict(slice_fusion="linear"),
    "size_b": dict(model_size="b"),
}


def test_weight_generator_pinned():
    ref = json.loads((GOLDEN / "weights.json").read_text())
    assert synth.state_dict_digest(synth.synth_state_dict("s", 0)) == ref["s_seed0"]
    assert synth.state_dict_digest(synth.synth_state_dict("s", 1, img_size=518, layerscale=True, chunked=False)) == ref["s_seed1_hub518"]
    assert float(synth.synth_volume((1, 1, 2, 28, 28), 3).double().sum()) == ref["volume_1x1x2x28x28_seed3"]


@pytest.mark.parametrize("name", list(CASES))
def test_end_to_end_matches_reference(name):
    g = load_golden(name)
    kw = dict(CASES[name])
    size = kw.pop("model_size", "s")
    fusion = kw.get("slice_fusion", "transformer")
    sd = synth.synth_state_dict(size, int(g["seed"]), **kw)
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    mask = torch.from_numpy(g["src_key_padding_mask"]) if "src_key_padding_mask" in g else None
    with torch.no_grad():
        out = O.forward(sd, src, model_size=size, slice_fusion_type=fusion, src_key_padding_mask=mask,
                        rotary=kw.get("rotary"), keep="cls")
    assert np.abs(out["emb"].numpy() - g["emb"]).max() < 2e-5
    assert np.abs(out["features"].numpy() - g["features"]).max() < 2e-5
    assert np.abs(out["logits"].numpy() - g["logits"]).max() < 2e-5
    if fusion == "transformer":
        rows = torch.stack([m[:, :, 0] for m in out["vit_maps"]]).numpy()
        assert np.abs(rows - g["vit_cls_rows"]).max() < 2e-5 and rel_l2(rows, g["vit_cls_rows"]) < 1e-4
        assert np.abs(out["slice_map"].numpy() - g["slice_map"]).max() < 2e-5
        nreg = 0
        assert rel_l2(O.plane_attention(out["vit_maps"][-1], nreg), g["plane_attention"]) < 1e-4
        assert rel_l2(O.slice_attention(out["slice_map"]), g["slice_attention"]) < 1e-4
        assert rel_l2(O.attention_maps(out["vit_maps"][-1], out["slice_map"], nreg), g["attention_maps"]) < 1e-4


def _big(name):
    p = GOLDEN / f"{name}.npz"
    if not p.exists():
        pytest.skip(f"{p.name} not generated")
    return load_golden(name)


@pytest.mark.parametrize("name,seed_slices", [("s504_1x4x504", None)])
def test_large_grid_matches_reference(name, seed_slices):
    """504x504 (36x36 grid, interpolated pos-embed), 4 slices: cheap enough for the CPU suite."""
    g = _big(name)
    sd = synth.synth_state_dict("s", int(g["seed"]))
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    with torch.no_grad():
        out = O.forward(sd, src, keep="cls")
    assert np.abs(out["emb"].numpy() - g["emb"]).max() < 5e-5
    assert np.abs(out["logits"].numpy() - g["logits"]).max() < 5e-5
    sub = g["plane_subset"].tolist()
    assert rel_l2(O.plane_attention(out["vit_maps"][-1])[sub], g["plane_attention"]) < 1e-4
    assert rel_l2(O.slice_attention(out["slice_map"]), g["slice_attention"]) < 1e-4


def test_ops_small_vit_and_pos_interpolation():
    g = load_golden("ops")
    sd = {"encoder." + k[len("vit_sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("vit_sd.")}
    O.VIT_CFG["_tiny"] = dict(embed_dim=64, depth=2, num_heads=2)
    try:
        for tag in ("56", "84x70", "70x84"):
            x = torch.from_numpy(g[f"vit_in_{tag}"])
            with torch.no_grad():
                y, _ = O.vit_encode(sd, x, "_tiny")
            assert np.abs(y.numpy() - g[f"vit_out_{tag}"]).max() < 5e-5, tag
    finally:
        del O.VIT_CFG["_tiny"]
    pe = torch.from_numpy(synth.hash_normal((1, 257, 384), 9, 1)) * 0.2
    for tag, hw in (("518", (518, 518)), ("504", (504, 504)), ("518x224", (518, 224))):
        n = (hw[0] // 14) * (hw[1] // 14)
        got = O.interpolate_pos_encoding(pe, n, hw[0], hw[1])
        assert np.abs(got.numpy() - g[f"pos224_to_{tag}"]).max() < 1e-6, tag


@pytest.mark.parametrize("tag", ["plain", "rope"])
def test_ops_slice_transformer_layer(tag):
    g = load_golden("ops")
    pre = f"tel_{tag}_sd."
    sd = {}
    for k, v in g.items():
        if k.startswith(pre):
            kk = k[len(pre):]
            sd[("slice_fusion." + kk)] = torch.from_numpy(v)
    x = torch.from_numpy(g[f"tel_{tag}_in"])
    mask = torch.from_numpy(g[f"tel_{tag}_mask"])
    rot = "RoPE" if tag == "rope" else None
    old = O.SLICE_HEADS
    try:
        with torch.no_grad():
            y, w = O.slice_fusion(sd, x, None, rot)
            ym, wm = O.slice_fusion(sd, x, mask, rot)
    finally:
        O.SLICE_HEADS = old
    assert np.abs(y.numpy() - g[f"tel_{tag}_out"]).max() < 1e-5
    assert np.abs(ym.numpy() - g[f"tel_{tag}_out_masked"]).max() < 1e-5
    assert np.abs(w.numpy() - g[f"tel_{tag}_weights"]).max() < 5e-6
    assert np.abs(wm.numpy() - g[f"tel_{tag}_weights_masked"]).max() < 5e-6


def test_liere_restrictions_match_reference():
    """The reference's LieRE path only works for B == 1 and D == 32 (views raise RuntimeError otherwise)."""
    err = json.loads((GOLDEN / "errors.json").read_text())
    assert err["liere_batch2"]["type"] == err["liere_d16"]["type"] == "RuntimeError"
    sd = synth.synth_state_dict("s", 9, rotary="LiRE")
    for shape in ((2, 1, 32, 28, 28), (1, 1, 16, 28, 28)):
        with pytest.raises(RuntimeError), torch.no_grad():
            O.forward(sd, torch.zeros(*shape), rotary="LiRE")


def test_error_fixture_messages():
    ref = json.loads((GOLDEN / "errors.json").read_text())
    sd = synth.synth_state_dict("s", 0)
    with pytest.raises(AssertionError) as e:
        O.vit_encode(sd, torch.zeros(2, 512, 512))
    assert str(e.value) == ref["512x512"]["message"]
    with pytest.raises(AssertionError) as e:
        O.vit_encode(sd, torch.zeros(2, 224, 230))
    assert str(e.value) == ref["224x230"]["message"]


def test_attention_rollout_matches_reference():
    """get_attention_cls (dino.py:204-212) on the reference's stored full maps."""
    g = load_golden("rollout_1x3x84")
    sd = synth.synth_state_dict("s", int(g["seed"]))
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    with torch.no_grad():
        out = O.forward(sd, src, keep="full")
    assert np.abs(out["logits"].numpy() - g["logits"]).max() < 2e-5
    assert np.abs(out["vit_maps"][0].numpy() - g["vit_full_first"]).max() < 5e-6
    assert np.abs(out["vit_maps"][-1].numpy() - g["vit_full_last"]).max() < 5e-6
    roll = O.attention_rollout(out["vit_maps"])
    assert roll.shape == g["attention_cls"].shape
    assert rel_l2(roll, g["attention_cls"]) < 1e-5


@pytest.mark.parametrize("name", ["saliency_1x5x84", "saliency_tta_1x4x56x84"])
def test_saliency_volume_matches_reference(name):
    """run_pred(save_attn=True[, use_tta]) of scripts/main_predict.py:134-165 on the reference model's outputs."""
    g = load_golden(name)
    sd = synth.synth_state_dict("s", int(g["seed"]))
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    with torch.no_grad():
        pred, weight, ws = O.run_pred(sd, src, use_softmax=True, use_tta=bool(g["use_tta"]))
    assert np.abs(pred.numpy() - g["pred"]).max() < 2e-5
    assert weight.shape == g["weight"].shape
    assert rel_l2(weight, g["weight"]) < 1e-4
    assert rel_l2(ws[0, 0, :, 0, 0], g["weight_slice_per_slice"]) < 1e-4
    assert int(g["weight_slice_is_broadcast"]) == 1 and bool((ws == ws[:, :, :, :1, :1]).all())


def test_trilinear_restatement_equals_torch_interpolate():
    import torch.nn.functional as F
    w = torch.rand(1, 1, 5, 7, 9, generator=torch.Generator().manual_seed(0))
    for size in [(5, 98, 126), (9, 14, 20), (5, 7, 9), (3, 100, 4)]:
        assert (O.trilinear_upsample(w, size) - F.interpolate(w, size=size, mode="trilinear")).abs().max() < 1e-6


def test_hub_register_encoder_matches_reference_class():
    """Encoder configured as torch.hub's dinov2_vits14_reg (registers, LayerScale, anti-aliased size-based pos resampling)."""
    g = load_golden("hub_reg")
    sd = synth.synth_state_dict("s", int(g["seed"]), img_size=518, layerscale=True, chunked=False, num_register_tokens=4)
    pe = sd["encoder.pos_embed"]
    assert rel_l2(O.interpolate_pos_encoding(pe, 256, 224, 224, offset=0.0, antialias=True), g["pos_16x16"]) < 1e-6
    assert rel_l2(O.interpolate_pos_encoding(pe, 80, 112, 140, offset=0.0, antialias=True), g["pos_8x10"]) < 1e-6
    for tag in ("224", "112x140", "518"):
        shape = tuple(int(v) for v in g[f"shape_{tag}"])
        x = synth.synth_volume((1, 1) + shape, int(g["seed"]) + 100)[0, 0]
        with torch.no_grad():
            emb, _ = O.vit_encode(sd, x)
        assert np.abs(emb.numpy() - g[f"emb_{tag}"]).max() < 5e-5, tag


def test_multichannel_input_matches_reference():
    """C = 3: 'b c d h w -> (b d c) h w' (dino.py:125) and a key-padding mask over the D*C pseudo-slices."""
    g = load_golden("multichannel")
    sd = synth.synth_state_dict("s", int(g["seed"]))
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    mask = torch.from_numpy(g["src_key_padding_mask"])
    with torch.no_grad():
        out = O.forward(sd, src, src_key_padding_mask=mask, keep="cls")
        feat = O.forward(sd, src, src_key_padding_mask=mask, without_linear=True)
    assert np.abs(out["logits"].numpy() - g["logits"]).max() < 2e-5
    assert np.abs(feat["features"].numpy() - g["features"]).max() < 5e-5
    assert rel_l2(O.slice_attention(out["slice_map"]), g["slice_attention"]) < 1e-4
    assert rel_l2(O.attention_maps(out["vit_maps"][-1], out["slice_map"]), g["attention_maps"]) < 1e-4


def test_slices2rgb_restatement_matches_reference_fixture(golden):
    """a16: the oracle's slices2rgb against what the reference's own function (dino.py:10-27) returned."""
    from mst import synth
    g = golden("slices2rgb")
    for i in range(3):
        x = synth.synth_volume(tuple(int(v) for v in g[f"shape{i}"]), int(g[f"seed{i}"]))
        assert np.array_equal(O.slices2rgb(x).numpy(), g[f"out{i}"])


def test_resnet_fusion_restatement_matches_reference_fixture(golden):
    """8f-2: the across-slice half of ResNetSliceTrans (16 heads, E 512) as restated by oracle/resnet_oracle.py against what the
    reference's own TransformerEncoderLayer produced."""
    from oracle import resnet_oracle as R
    g = golden("resnet_fusion")
    sd = synth.synth_resnet_state_dict(int(g["seed"]), 34, 2)
    emb = torch.from_numpy(g["emb"])
    B, D, E = emb.shape
    for tag, m in (("", None), ("_masked", torch.from_numpy(g["src_key_padding_mask"]))):
        out = R.fuse(sd, emb.reshape(B * D, E), B, D, m)
        assert np.abs(out["logits"].numpy() - g["logits" + tag]).max() < 1e-5
        assert np.abs(out["features"].numpy() - g["features" + tag]).max() < 1e-5
        assert np.abs(out["slice_map"].numpy() - g["slice_map" + tag]).max() < 1e-6
