#!/usr/bin/env python3
"""Latency of the small configs (BASELINE configs[0]: 1 x 16 x 224^2) with and without hipGraph replay of the forward."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice

only = "--bf16-16-only" in sys.argv                # profiling: one mode, one shape, no graph
for mode in (("bf16",) if only else ("bf16", "fp32")):
    model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=mode)
    model.load_state_dict(synth.synth_state_dict("s", 0))
    model = model.cuda().eval()
    for shape in (((1, 1, 16, 224, 224),) if only else ((1, 1, 16, 224, 224), (1, 1, 32, 224, 224))):
        src = torch.randn(*shape, device="cuda")
        with torch.no_grad():
            for _ in range(5):
                out = model(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 50
            for _ in range(n):
                out = model(src)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / n
            if only:
                print({"mode": mode, "shape": shape, "eager_ms": round(eager * 1e3, 3)})
                continue
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                out = model(src)
            torch.cuda.current_stream().wait_stream(s)
            try:
                with torch.cuda.graph(g):
                    out_g = model(src)
                g.replay()
                torch.cuda.synchronize()
                ok = bool(torch.equal(out_g, out))
                t0 = time.perf_counter()
                for _ in range(n):
                    g.replay()
                torch.cuda.synchronize()
                graph = (time.perf_counter() - t0) / n
            except Exception as e:  # noqa: BLE001
                ok, graph = repr(e)[:200], float("nan")
        print({"mode": mode, "shape": shape, "eager_ms": round(eager * 1e3, 3), "graph_ms": round(graph * 1e3, 3), "graph_equal": ok})
