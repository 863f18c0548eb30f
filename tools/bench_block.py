#!/usr/bin/env python3
"""A/B in one process (HIP events, interleaved rounds): out-projection launch + mst_mlp_fused against mst_block_fused."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 350720
dt = torch.bfloat16
torch.manual_seed(0)
E, H = 384, 1536
x = torch.randn(M, E, device="cuda")
att = torch.randn(M, E, device="cuda").to(dt)
wp = (torch.randn(E, E, device="cuda") / E ** 0.5); bp = torch.randn(E, device="cuda") * 0.1
w1 = torch.randn(H, E, device="cuda") / E ** 0.5; b1 = torch.randn(H, device="cuda") * 0.1
w2 = torch.randn(E, H, device="cuda") / H ** 0.5; b2 = torch.randn(E, device="cuda") * 0.1
g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
wpack, b1p, b2p = hip.pack_mlp(w1, b1, w2, b2, g, be, None, dt)
ppack, pbf = hip.pack_proj(wp, bp, None, dt)
wp16 = wp.to(dt)
xn = torch.empty(M, E, device="cuda", dtype=dt)
scratch = torch.empty(int(hip.load().mst_block_fused_scratch_bytes()), dtype=torch.uint8, device="cuda")

def old():
    hip.gemm(att, wp16, bp, epilogue=hip.EPI_RESIDUAL, out=x)
    hip.mlp_fused(x, wpack, b1p, b2p, xn, dt)

def new():
    hip.block_fused(x, att, ppack, pbf, wpack, b1p, b2p, xn, scratch=scratch)

seq, b1s, pbs, b2s = hip.pack_block_seq(wp, bp, None, w1, b1, w2, b2, g, be, None, dt)

def single():
    hip.block_fused_s(x, att, seq, b1s, pbs, b2s, xn)

def single_blk():
    hip.block_fused_s(x, att, seq, b1s, pbs, b2s, xn, layout=7)

for f in (old, new, single, single_blk):
    for _ in range(2):
        f()
torch.cuda.synchronize()
res = {"old": [], "new": [], "single": [], "single_blk": []}
for rnd in range(6):
    for name, f in (("old", old), ("new", new), ("single", single), ("single_blk", single_blk)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
        x.normal_()                                   # keep the residual stream bounded
fl = 2.0 * M * E * E + 4.0 * M * H * E
for k, v in res.items():
    v = sorted(v)
    print(k, {"min_ms": round(v[0], 4), "median_ms": round(v[len(v) // 2], 4), "tflops_at_median": round(fl / v[len(v) // 2] / 1e9, 1)})
