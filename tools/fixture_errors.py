#!/usr/bin/env python3
"""Diagnostic: logits / embedding errors of every end-to-end fixture in each precision mode (run on the GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd"), str(ROOT / "tests")]
import numpy as np, torch
from conftest import load_golden, rel_l2
from mst import synth
import test_model_gpu as T

for name, kw in T.CASES.items():
    g = load_golden(name)
    src = synth.synth_volume(tuple(int(v) for v in g["shape"]), int(g["seed"]) + 100)
    mask = torch.from_numpy(g["src_key_padding_mask"]) if "src_key_padding_mask" in g else None
    row = []
    for mode in ("fp32", "fp16", "bf16"):
        model = T.build(kw, int(g["seed"]), mode)
        with torch.no_grad():
            logits = model(src, src_key_padding_mask=mask)
            B, _, D, H, W = src.shape
            emb, _, _ = model.encode_slices(src.cuda().reshape(B * D, H, W))
        row.append(f"{mode}: dlogit {np.abs(logits.cpu().numpy() - g['logits']).max():.2e} emb {rel_l2(emb.cpu(), g['emb']):.2e}")
    print(f"{name:16s}", " | ".join(row), flush=True)
