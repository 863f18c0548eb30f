#!/usr/bin/env python3
"""Micro-benchmark of mst_gemm on the encoder's shapes (HIP events, interleaved reps)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 350720
shapes = {"qkv": (1152, 384, 0), "proj": (384, 384, 3), "fc1": (1536, 384, 1), "fc2": (384, 1536, 3)}
dt = torch.bfloat16
torch.manual_seed(0)
only = sys.argv[2:]
res = {}
for name, (N, K, epi) in shapes.items():
    if only and name not in only:
        continue
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 3 else dt)
    for _ in range(3):
        hip.gemm(a, w, b, epilogue=epi, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        hip.gemm(a, w, b, epilogue=epi, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res[name] = (round(ms, 4), round(2.0 * M * N * K / ms / 1e9, 1))
print(res)
