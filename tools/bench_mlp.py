#!/usr/bin/env python3
"""Micro-benchmark of mst_mlp_fused (HIP events)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 350720
dt = torch.bfloat16
torch.manual_seed(0)
E, H = 384, 1536
x = torch.randn(M, E, device="cuda")
w1 = torch.randn(H, E, device="cuda") / E ** 0.5; b1 = torch.randn(H, device="cuda") * 0.1
w2 = torch.randn(E, H, device="cuda") / H ** 0.5; b2 = torch.randn(E, device="cuda") * 0.1
g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
wpack, b1p, b2p = hip.pack_mlp(w1, b1, w2, b2, g, be, None, dt)
xn = torch.empty(M, E, device="cuda", dtype=dt)
for _ in range(2):
    hip.mlp_fused(x, wpack, b1p, b2p, xn, dt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 10
for _ in range(reps):
    hip.mlp_fused(x, wpack, b1p, b2p, xn, dt)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print({"mlp_ms": round(ms, 4), "tflops": round(4.0 * M * H * E / ms / 1e9, 1)})
