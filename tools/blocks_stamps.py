#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the single-role block kernel (needs a -DBLOCKS_STAMPS build: MST_HIP_LIB)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import numpy as np
import torch
from mst import hip
LAYOUT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = 350720; dt = torch.bfloat16; E, H = 384, 1536
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
x = g(M, E); att = g(M, E).to(dt)
seq, b1f, pbf, b2f = hip.pack_block_seq(g(E, E) / E ** .5, g(E) * .1, None, g(H, E) / E ** .5, g(H) * .1, g(E, H) / H ** .5, g(E) * .1,
                                        torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), None, dt)
xn = torch.empty(M, E, device="cuda", dtype=dt)
for _ in range(3):
    x.normal_()
    hip.block_fused_s(x, att, seq, b1f, pbf, b2f, xn, layout=LAYOUT)
torch.cuda.synchronize()
lib = hip.load()
buf = (C.c_ulonglong * (256 * 4 * 16))()
lib.mst_debug_blocks_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.mst_debug_blocks_stamps(buf, 256 * 4 * 16))
a = np.array(buf, dtype=np.float64).reshape(-1, 16)
tiles = 2740 / 256.0
m = lambda i: float(np.median(a[:, i])) / tiles
print("cycles per TILE (median over waves; 108 phases: 12 proj, G1(0), 47 x (A, B), B(47))")
print("total %.0f | proj: wait %.0f body %.0f (per phase %.0f + %.0f) | A: wait %.0f body %.0f (per phase %.0f + %.0f) | B: wait %.0f body %.0f (per phase %.0f + %.0f) | "
      "G1(0) %.0f + %.0f | row loads issue %.0f | LN2 %.0f | epilogue %.0f" % (
          m(15), m(0), m(1), m(0) / 12, m(1) / 12, m(2), m(3), m(2) / 47, m(3) / 47, m(4), m(5), m(4) / 48, m(5) / 48, m(6), m(7), m(8), m(9), m(10)))
