#!/usr/bin/env python3
"""QKV-shaped mst_gemm only (M x 1152 x 384, bias + q scaling), a few repetitions: for rocprofv3 --pmc passes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M, N, K = 350720, 1152, 384
dt = torch.bfloat16
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").to(dt)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=dt)
for _ in range(6):
    hip.gemm(a, w, b, epilogue=0, out=out)
torch.cuda.synchronize()
