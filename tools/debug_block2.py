#!/usr/bin/env python3
"""Diagnostic build (-DBLOCK_DEBUG): compare the LayerNorm2 hand-off as written by the consumers and as read by the producers."""
import ctypes as C, math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
torch.manual_seed(0)
E, H = 384, 1536
dt = torch.bfloat16
M = 128
x = torch.randn(M, E, device="cuda"); att = torch.randn(M, E, device="cuda").to(dt)
wp = torch.randn(E, E, device="cuda") / E ** 0.5; bp = torch.randn(E, device="cuda") * 0.1
w1 = torch.randn(H, E, device="cuda") / E ** 0.5; b1 = torch.randn(H, device="cuda") * 0.1
w2 = torch.randn(E, H, device="cuda") / H ** 0.5; b2 = torch.randn(E, device="cuda") * 0.1
g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
wpack, b1p, b2p = hip.pack_mlp(w1, b1, w2, b2, g, be, None, dt)
ppack, pbf = hip.pack_proj(wp, bp, None, dt)
lib = hip.load()
dbg = torch.full((1, 2, 4, 24, 64, 8), float("nan"), device="cuda", dtype=dt)
lib.mst_debug_block_set.argtypes = [C.c_void_p]
print("set", lib.mst_debug_block_set(dbg.data_ptr()))
xc = x.clone(); xn = torch.empty(M, E, device="cuda", dtype=dt)
hip.block_fused(xc, att, ppack, pbf, wpack, b1p, b2p, xn)
torch.cuda.synchronize()
cons, prod = dbg[0, 0].float(), dbg[0, 1].float()       # [pr, i, lane, 8]
xmid = x.double() + att.double() @ wp.to(dt).double().t() + bp.double()
ref = torch.nn.functional.layer_norm(xmid, (E,)).float()
# lane (frow, g), i = 12*mt + ks  ->  row pr*32 + 16*mt + frow, cols 32*ks + 8*g + 0..7
lane = torch.arange(64, device="cuda"); frow, gg = lane & 15, lane >> 4
def unpack(t):
    out = torch.zeros(M, E, device="cuda")
    for pr in range(4):
        for mt in range(2):
            for ks in range(12):
                rows = pr * 32 + 16 * mt + frow
                cols = (32 * ks + 8 * gg)[:, None] + torch.arange(8, device="cuda")[None, :]
                out[rows[:, None], cols] = t[pr, 12 * mt + ks]
    return out
uc, up = unpack(cons), unpack(prod)
print("consumer wrote vs LN(x_mid): max err", float((uc - ref).abs().max()), "nan", int(torch.isnan(uc).sum()))
print("producer read  vs consumer : mismatching elements", int((up != uc).sum()), "nan", int(torch.isnan(up).sum()))
bad = torch.nonzero((up != uc).any(dim=1)).flatten().tolist()
print("rows where the producer saw something else:", bad)
bl = torch.nonzero((prod != cons).flatten(2).any(dim=2))
print("(pr, i) pieces with mismatches:", bl.tolist()[:60])
m = (prod != cons)
print("mismatching lanes (any piece):", torch.nonzero(m.any(dim=3).any(dim=1).any(dim=0)).flatten().tolist())
xd = x.double()
h = torch.nn.functional.layer_norm(xmid, (E,)).to(dt).double() @ w1.to(dt).double().t() + b1.double()
h = (0.5 * h * (1 + torch.erf(h / math.sqrt(2)))).to(dt).double()
refo = xmid + h @ w2.to(dt).double().t() + b2.double()
rowerr = (xc.double() - refo).abs().max(dim=1).values
print("bad output rows:", torch.nonzero(rowerr > 0.1).flatten().tolist())
