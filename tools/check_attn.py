#!/usr/bin/env python3
"""Correctness of mst_attention at the bench shape and a ragged small one against torch (fp32 softmax of the same 16-bit operands)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
ok = True
for n, N, heads in ((4, 1370, 6), (3, 257, 6), (2, 90, 6), (5, 17, 6)):
    torch.manual_seed(N)
    qkv = (torch.randn(n * N, 3 * heads * 64, device="cuda") * 0.5).bfloat16()
    out = hip.attention(qkv, n, N, heads)
    t = qkv.float().view(n, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(t[0] @ t[1].transpose(-1, -2), dim=-1) @ t[2]
    ref = ref.permute(0, 2, 1, 3).reshape(n * N, heads * 64)
    err = float((out.float() - ref).abs().max())
    print({"n": n, "N": N, "max_abs_err": err})
    ok &= err < 2e-2
assert ok
