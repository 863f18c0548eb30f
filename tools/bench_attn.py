#!/usr/bin/env python3
"""Micro-benchmark of mst_attention at the bench shape (256 slices x 1370 tokens, 6 heads x 64)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
n, N, heads = 256, (int(sys.argv[1]) if len(sys.argv) > 1 else 1370), 6
dt = torch.bfloat16
torch.manual_seed(0)
qkv = (torch.randn(n * N, 3 * heads * 64, device="cuda") * 0.5).to(dt)
out = torch.empty(n * N, heads * 64, device="cuda", dtype=dt)
for _ in range(3):
    out = hip.attention(qkv, n, N, heads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 20
for _ in range(reps):
    out = hip.attention(qkv, n, N, heads)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print({"N": N, "attn_ms": round(ms, 4), "tflops": round(4.0 * n * N * N * heads * 64 / ms / 1e9, 1)})
