#!/usr/bin/env python3
"""Correctness of mst_gemm on the full QKV shape (bias + q scaling) against torch.matmul on the same 16-bit operands."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M, N, K = 350720 - 37, 1152, 384
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
hip.gemm(a, w, b, epilogue=0, out=out, col_scale=0.125, scale_cols=384)
ref = a.float() @ w.float().t() + b
ref[:, :384] *= 0.125
err = (out.float() - ref).abs().max().item()
print({"max_abs_err": err, "ok": err < 0.05})
assert err < 0.05
