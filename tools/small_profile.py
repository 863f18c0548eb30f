#!/usr/bin/env python3
"""c1 shape (1 x 16 x 224^2, bf16) forward, 20 repetitions: target of `rocprofv3 --kernel-trace --stats`."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice
model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype="bf16")
model.load_state_dict(synth.synth_state_dict("s", 0))
model = model.cuda().eval()
src = torch.randn(1, 1, 16, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(20):
        model(src)
torch.cuda.synchronize()
