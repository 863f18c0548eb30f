#!/usr/bin/env python3
"""Correctness of mst_gemm on the full QKV shape (bias + q scaling, ragged M) against torch.matmul on the same 16-bit operands,
for both operand types and with strided rows."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
ok_all = True
for dt in (torch.bfloat16, torch.float16):
    for M, lda in ((350720 - 37, 384), (70001, 392), (65536, 384)):
        N, K = 1152, 384
        torch.manual_seed(0)
        a = torch.randn(M, lda, device="cuda").to(dt)[:, :K]
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
        b = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=dt)
        lib = hip.load()
        hip._check(lib.mst_gemm(a.data_ptr(), hip._DT[dt], lda, hip.ptr(w), K, hip.ptr(b), hip.ptr(out), hip._DT[dt], N, M, N, K, 0,
                                None, 0.125, 384, hip.stream_of(out)), "mst_gemm")
        ref = a.float() @ w.float().t() + b
        ref[:, :384] *= 0.125
        err = (out.float() - ref).abs().max().item()
        tol = 0.05 if dt == torch.bfloat16 else 0.01
        print({"dtype": str(dt), "M": M, "lda": lda, "max_abs_err": err, "ok": err < tol})
        ok_all &= err < tol
assert ok_all
