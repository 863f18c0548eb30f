#!/usr/bin/env python3
"""Stage-by-stage check of mst_block_fused_s against fp64: zero out parts of the block to isolate a wrong stage."""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
torch.manual_seed(0)
E, Hd = 384, 1536
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = torch.bfloat16

def run(name, wp, bp, w1, b1, w2, b2, g, be, x, att):
    seq, b1f, pbf, b2f = hip.pack_block_seq(wp.cuda(), bp.cuda(), None, w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(), be.cuda(), None, dt)
    xc = x.cuda().clone(); attc = att.cuda().clone()
    xn = torch.empty(M, E, dtype=dt, device="cuda")
    hip.block_fused_s(xc, attc, seq, b1f, pbf, b2f, xn)
    torch.cuda.synchronize()
    xd = x.double()
    proj = att.double() @ wp.double().t() + bp.double()
    xmid = xd + proj
    h = torch.nn.functional.layer_norm(xmid, (E,), g.double(), be.double(), 1e-6)
    h = h @ w1.double().t() + b1.double()
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2)))
    y = h @ w2.double().t() + b2.double()
    ref = xmid + y
    err = (xc.double().cpu() - ref).abs()
    refn = torch.nn.functional.layer_norm(ref, (E,))
    errn = (xn.double().cpu() - refn).abs()
    print(f"{name:28s} max|dx| {float(err.max()):.4e} (|y|max {float(y.abs().max()):.3f} |proj|max {float(proj.abs().max()):.3f})  max|dxn| {float(errn.max()):.4e}"
          f"  worst row {int(err.max(1).values.argmax())} col {int(err.max(0).values.argmax())}")
    return err

Z = torch.zeros
x = torch.randn(M, E) * 1.5 + 0.3
att = torch.randn(M, E).to(dt)
wp, bp = torch.randn(E, E) / math.sqrt(E), torch.randn(E) * 0.1
w1, b1 = torch.randn(Hd, E) / math.sqrt(E), torch.randn(Hd) * 0.1
w2, b2 = torch.randn(E, Hd) / math.sqrt(Hd), torch.randn(E) * 0.1
g, be = torch.ones(E), Z(E)
run("identity (all zero)", Z(E, E), Z(E), Z(Hd, E), Z(Hd), Z(E, Hd), Z(E), g, be, x, att)
run("bproj only", Z(E, E), bp, Z(Hd, E), Z(Hd), Z(E, Hd), Z(E), g, be, x, att)
run("b2 only", Z(E, E), Z(E), Z(Hd, E), Z(Hd), Z(E, Hd), b2, g, be, x, att)
run("proj only", wp, bp, Z(Hd, E), Z(Hd), Z(E, Hd), Z(E), g, be, x, att)
run("b1 -> gelu -> w2 (w1=0)", Z(E, E), Z(E), Z(Hd, E), b1 * 10, w2, Z(E), g, be, x, att)
e = run("mlp only (proj=0)", Z(E, E), Z(E), w1, b1, w2, b2, g, be, x, att)
run("mlp, w1 chunk 0 only", Z(E, E), Z(E), torch.cat([w1[:32], Z(Hd - 32, E)]), Z(Hd), w2, Z(E), g, be, x, att)
run("mlp, w1 chunk 1 only", Z(E, E), Z(E), torch.cat([Z(32, E), w1[32:64], Z(Hd - 64, E)]), Z(Hd), w2, Z(E), g, be, x, att)
run("mlp, w1 chunk 47 only", Z(E, E), Z(E), torch.cat([Z(Hd - 32, E), w1[-32:]]), Z(Hd), w2, Z(E), g, be, x, att)
run("full", wp, bp, w1, b1, w2, b2, g, be, x, att)
print("per-32-row-group max err of mlp only:", [round(float(e[i:i + 32].max()), 4) for i in range(0, M, 32)][:8])
