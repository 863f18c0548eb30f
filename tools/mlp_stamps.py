#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused MLP kernel (needs a -DMLP_STAMPS build of the library)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M = 350720; dt = torch.bfloat16; E, H = 384, 1536
x = torch.randn(M, E, device="cuda")
b2 = torch.zeros(E, device="cuda"); xn = torch.empty(M, E, device="cuda", dtype=dt)
wpack, b1p, b2p = hip.pack_mlp(torch.randn(H, E, device="cuda") / E ** .5, torch.randn(H, device="cuda") * .1,
                               torch.randn(E, H, device="cuda") / H ** .5, b2, torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), None, dt)
for _ in range(3):
    hip.mlp_fused(x, wpack, b1p, b2p, xn, dt)
torch.cuda.synchronize()
lib = hip.load()
buf = (C.c_ulonglong * (256 * 8 * 4))()
lib.mst_debug_mlp_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.mst_debug_mlp_stamps(buf, 256 * 8 * 4))
import numpy as np
a = np.array(buf, dtype=np.float64).reshape(256, 8, 4)
steps = 11 * 48 + 1
print("per-step cycles (median over workgroups)")
print("producer [barrier, LN(c==0), GEMM1, GELU+handoff]:", np.median(a[:, :4].reshape(-1, 4), axis=0) / steps)
print("consumer [wait+barrier, DMA issue, GEMM2, epilogue]:", np.median(a[:, 4:].reshape(-1, 4), axis=0) / steps)
print("sum producer", np.median(a[:, :4].sum(-1)) / steps, "consumer", np.median(a[:, 4:].sum(-1)) / steps)
