#!/usr/bin/env python3
"""Diagnostic for mst_block_fused: isolate the out-projection, the LayerNorm hand-off and the MLP by zeroing operands."""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
torch.manual_seed(0)
E, H = 384, 1536
dt = torch.bfloat16

def run(M, zero_proj=False, zero_w2=False, zero_w1=False, label=""):
    x = torch.randn(M, E, device="cuda")
    att = torch.randn(M, E, device="cuda").to(dt)
    wp = torch.randn(E, E, device="cuda") / E ** 0.5; bp = torch.randn(E, device="cuda") * 0.1
    w1 = torch.randn(H, E, device="cuda") / E ** 0.5; b1 = torch.randn(H, device="cuda") * 0.1
    w2 = torch.randn(E, H, device="cuda") / H ** 0.5; b2 = torch.randn(E, device="cuda") * 0.1
    if zero_proj: wp.zero_(); bp.zero_()
    if zero_w2: w2.zero_()
    if zero_w1: w1.zero_()
    g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
    wpack, b1p, b2p = hip.pack_mlp(w1, b1, w2, b2, g, be, None, dt)
    ppack, pbf = hip.pack_proj(wp, bp, None, dt)
    xc = x.clone(); xn = torch.empty(M, E, device="cuda", dtype=dt)
    hip.block_fused(xc, att, ppack, pbf, wpack, b1p, b2p, xn)
    torch.cuda.synchronize()
    xd = x.double()
    xmid = xd + att.double() @ wp.to(dt).double().t() + bp.double()
    h = torch.nn.functional.layer_norm(xmid, (E,)).to(dt).double() @ w1.to(dt).double().t() + b1.double()
    h = (0.5 * h * (1 + torch.erf(h / math.sqrt(2)))).to(dt).double()
    ref = xmid + h @ w2.to(dt).double().t() + b2.double()
    err = (xc.double() - ref).abs()
    bad = ~torch.isfinite(xc)
    rowerr = err.max(dim=1).values
    colerr = err.max(dim=0).values
    print(f"[{label}] M={M} max err {float(err[torch.isfinite(err)].max()):.3e} nonfinite {int(bad.sum())} "
          f"bad rows(>0.1) {int((rowerr > 0.1).sum())} of {M}; bad cols {int((colerr > 0.1).sum())}")
    br = torch.nonzero(rowerr > 0.1).flatten().tolist()
    print("   bad rows:", br[:40], "..." if len(br) > 40 else "")
    bc = torch.nonzero(colerr > 0.1).flatten().tolist()
    print("   bad cols:", bc[:48], "..." if len(bc) > 48 else "")
    xnr = torch.nn.functional.layer_norm(ref, (E,))
    e2 = (xn.double() - xnr).abs().max(dim=1).values
    print("   xn bad rows:", torch.nonzero(e2 > 0.2).flatten().tolist()[:40])

for M in (16, 128, 300):
    run(M, label="full")
run(128, zero_proj=True, label="no proj")
run(128, zero_w2=True, label="no W2 (proj + epilogue)")
run(128, zero_w1=True, label="no W1")
run(128 * 40, label="full 40 tiles")
