#!/usr/bin/env python3
"""`--get_attention` path at the c3 shape (1 x 64 x 518^2, fp32 parity mode and bf16): run_pred with and without TTA, and
the up-sampling kernel alone against its HBM roofline (68.7 MB written once)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
import torch
from mst import hip, synth
from mst.models import DinoV2ClassifierSlice
from mst.saliency import run_pred

low = torch.rand(64, 37, 37, device="cuda")
for _ in range(3):
    out = hip.saliency_upsample(low, (64, 518, 518), 0.125)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    out = hip.saliency_upsample(low, (64, 518, 518), 0.125)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print({"upsample_ms": round(ms, 4), "GB_per_s_written": round(out.numel() * 4 / ms / 1e6, 1), "hbm_peak_GB_per_s": 8000})

src = torch.randn(1, 1, 64, 518, 518, device="cuda")
for mode in ("bf16", "fp32"):
    model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=mode)
    model.load_state_dict(synth.synth_state_dict("s", 0))
    model = model.cuda().eval()
    for tta in (False, True):
        run_pred(model, {"source": src}, save_attn=True, use_tta=tta)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            run_pred(model, {"source": src}, save_attn=True, use_tta=tta)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print({"mode": mode, "use_tta": tta, "ms_per_volume": round(dt * 1e3, 2), "volumes_per_s": round(1 / dt, 2)})
