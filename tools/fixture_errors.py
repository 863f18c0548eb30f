#!/usr/bin/env python3
"""Diagnostic: logits / embedding / attention-map errors of every end-to-end fixture in each precision mode (run on the
GPU box).  The tolerances of tests/test_model_gpu.py are set from this table (<= 2x the measured maximum per mode)."""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd"), str(ROOT / "tests")]
import numpy as np, torch
from conftest import load_golden, rel_l2
from mst import synth
import test_model_gpu as T

worst = {}
for name, kw in list(T.CASES.items()) + [("c3_1x64x518", {})]:
    g = load_golden(name)
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    mask = torch.from_numpy(g["src_key_padding_mask"]) if "src_key_padding_mask" in g else None
    row = []
    for mode in ("fp32", "fp16", "bf16"):
        model = T.build(kw, int(g["seed"]), mode)
        with torch.no_grad():
            logits = model(src, src_key_padding_mask=mask, save_attn="attention_maps" in g)
            B, _, D, H, W = src.shape
            emb, _, _ = model.encode_slices(src.cuda().reshape(B * D, H, W))
        e = {"dlogit": float(np.abs(logits.cpu().numpy() - g["logits"]).max()), "emb": rel_l2(emb.cpu(), g["emb"])}
        if "attention_maps" in g:
            sub = g["plane_subset"].tolist() if "plane_subset" in g else slice(None)
            if "vit_cls_rows" in g:
                e["rows"] = rel_l2(torch.stack([m[:, :, 0] for m in model.attention_maps]).cpu(), g["vit_cls_rows"])
            e["slice_map"] = rel_l2(model.attention_maps_slice[-1].cpu(), g["slice_map"])
            e["plane"] = rel_l2(model.get_plane_attention().cpu()[sub], g["plane_attention"])
            e["slice_attn"] = rel_l2(model.get_slice_attention().cpu(), g["slice_attention"])
            e["maps"] = rel_l2(model.get_attention_maps().cpu()[sub], g["attention_maps"])
        for k, v in e.items():
            worst.setdefault(mode, {})[k] = max(worst.get(mode, {}).get(k, 0.0), v)
        row.append(f"{mode}: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
        del model
    print(f"{name:16s}", " | ".join(row), flush=True)
print("WORST", json.dumps(worst))
