#!/usr/bin/env python3
"""Shader clock held during the QKV GEMM (diagnostic builds with -DWREG_CLOCK: every workgroup leaves
(delta s_memtime, delta s_memrealtime) in the first 16 bytes of its first output row).  After 2 s of back-to-back launches."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
import torch
from mst import hip
M, N, K = 350720, 1152, 384
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
n = 0
while time.time() - t0 < 2.0:
    for _ in range(50):
        hip.gemm(a, w, b, epilogue=0, out=out)
    torch.cuda.synchronize()
    n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    hip.gemm(a, w, b, epilogue=0, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
nchunks = M // 32
per_xcd = (nchunks + 7) // 8
vals = []
raw = out.view(torch.int64).view(M, N // 4)
for xcd in range(8):
    for g in range(32):
        nt, j = g % 3, g // 3
        row = (xcd * per_xcd + j) * 32
        d = raw[row, nt * 96: nt * 96 + 2].tolist()
        if d[1] > 0:
            vals.append((d[0] / d[1] * 100e6, d[0]))
vals = [v for v in vals if 0.5e9 < v[0] < 3e9]
if not vals:
    print({"ms": round(ms, 4), "clock": "no stamps (not a -DWREG_CLOCK build)"})
    sys.exit(0)
vals.sort()
med = vals[len(vals) // 2]
print({"ms": round(ms, 4), "clock_GHz_median": round(med[0] / 1e9, 3), "clock_min": round(vals[0][0] / 1e9, 3), "clock_max": round(vals[-1][0] / 1e9, 3),
       "kernel_cycles_median": sorted(v[1] for v in vals)[len(vals) // 2], "wgs": len(vals)})
