#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the block kernel (needs a -DBLOCK_STAMPS build of the library: MST_HIP_LIB)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import numpy as np
import torch
from mst import hip
M = 350720; dt = torch.bfloat16; E, H = 384, 1536
torch.manual_seed(0)
x = torch.randn(M, E, device="cuda"); att = torch.randn(M, E, device="cuda").to(dt)
wpack, b1p, b2p = hip.pack_mlp(torch.randn(H, E, device="cuda") / E ** .5, torch.randn(H, device="cuda") * .1,
                               torch.randn(E, H, device="cuda") / H ** .5, torch.zeros(E, device="cuda"), torch.ones(E, device="cuda"),
                               torch.zeros(E, device="cuda"), None, dt)
ppack, pbf = hip.pack_proj(torch.randn(E, E, device="cuda") / E ** .5, torch.zeros(E, device="cuda"), None, dt)
xn = torch.empty(M, E, device="cuda", dtype=dt)
scratch = torch.empty(int(hip.load().mst_block_fused_scratch_bytes()), dtype=torch.uint8, device="cuda")
for _ in range(3):
    x.normal_()
    hip.block_fused(x, att, ppack, pbf, wpack, b1p, b2p, xn, scratch=scratch)
torch.cuda.synchronize()
lib = hip.load()
buf = (C.c_ulonglong * (256 * 8 * 8))()
lib.mst_debug_block_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.mst_debug_block_stamps(buf, 256 * 8 * 8))
a = np.array(buf, dtype=np.float64).reshape(256, 8, 8)
tiles = 2740 / 256.0
prod, cons = a[:, :4].reshape(-1, 8), a[:, 4:].reshape(-1, 8)
med = lambda v: float(np.median(v))
print("cycles per TILE (median over waves; a tile = 62 steps: 12 proj, 2 idle, 48 MLP)")
print("consumer: total %.0f | MLP steps: wait+barrier %.0f, DMA+GEMM2 %.0f (per step %.0f + %.0f) | proj steps %.0f (per step %.0f) | "
      "x load+bias %.0f | LN2+scratch+bias %.0f | last step + epilogue %.0f | idle steps %.0f" % (
          med(cons[:, 7]) / tiles, med(cons[:, 0]) / tiles, med(cons[:, 1]) / tiles, med(cons[:, 0]) / tiles / 47, med(cons[:, 1]) / tiles / 47,
          med(cons[:, 2]) / tiles, med(cons[:, 2]) / tiles / 12, med(cons[:, 3]) / tiles, med(cons[:, 4]) / tiles, med(cons[:, 5]) / tiles,
          med(cons[:, 6]) / tiles))
print("producer: total %.0f | GEMM1 steps: barrier wait %.0f, work %.0f (per step %.0f + %.0f) | idle-step barrier waits %.0f | scratch load + GEMM1(0) %.0f" % (
    med(prod[:, 7]) / tiles, med(prod[:, 0]) / tiles, med(prod[:, 1]) / tiles, med(prod[:, 0]) / tiles / 48, med(prod[:, 1]) / tiles / 48,
    med(prod[:, 2]) / tiles, med(prod[:, 3]) / tiles))
