#!/usr/bin/env python3
"""Forward latency vs number of tokens, fused (MST_FUSED_MIN_TOKENS=0) or unfused (=10**12) encoder path: run twice."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice
model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype="bf16")
model.load_state_dict(synth.synth_state_dict("s", 0))
model = model.cuda().eval()
res = {}
for D in (16, 32, 64, 96, 128, 192, 256):
    src = torch.randn(1, 1, D, 224, 224, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            model(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(src)
        torch.cuda.synchronize()
    res[D * 257] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
print(os.environ.get("MST_FUSED_MIN_TOKENS"), res)
