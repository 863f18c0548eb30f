#!/usr/bin/env python3
"""c3 at full size: forward with the FULL attention maps of all 12 blocks (34.6 GB fp32) + get_attention_cls rollout."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice
mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=mode, full_attention_maps=True)
model.load_state_dict(synth.synth_state_dict("s", 0))
model = model.cuda().eval()
src = torch.randn(1, 1, 64, 518, 518, device="cuda")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logits = model(src, save_attn=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    roll = model.get_attention_cls()
    torch.cuda.synchronize(); t2 = time.perf_counter()
rows = roll.sum(-1)
print({"mode": mode, "forward_with_full_maps_s": round(t1 - t0, 3), "rollout_s": round(t2 - t1, 3), "rollout_shape": tuple(roll.shape),
       "rollout_tflops": round(11 * 64 * 6 * 2 * 1370 ** 3 / (t2 - t1) / 1e12, 1), "row_sum_min_max": (float(rows.min()), float(rows.max())),
       "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)})
