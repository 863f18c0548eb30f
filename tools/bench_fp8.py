"""fp8 (e4m3) linear layers on the GPU box: the four block GEMM shapes of the bench workload, mst_gemm_fp8 (MX-scaled or plain
fp8 MFMA: MST_FP8_MX=0/1) against the bf16 mst_gemm, the quantise pass, and the whole forward in fp8 / bf16 mode."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd"))
from mst import hip, synth  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    hip.load()
    M = 256 * 1370
    out = {"mx": os.environ.get("MST_FP8_MX", "1")}
    for name, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5)
        bias = torch.zeros(N, device="cuda")
        w8, sw = hip.quantize_weight_fp8(w)
        a8, amax = hip.quantize_fp8(a)
        odt = torch.float32 if name in ("proj", "fc2") else torch.bfloat16
        c8 = torch.empty(M, N, device="cuda", dtype=odt)
        c16 = torch.empty(M, N, device="cuda", dtype=odt)
        wb = w.to(torch.bfloat16)
        t8 = timeit(lambda: hip.gemm_fp8(a8, amax, w8, sw, bias, out=c8))
        t16 = timeit(lambda: hip.gemm(a, wb, bias, out=c16))
        tq = timeit(lambda: hip.quantize_fp8(a, amax))
        fl = 2.0 * M * N * K
        out[name] = {"fp8_ms": round(t8, 4), "fp8_tflops": round(fl / t8 / 1e9, 1), "bf16_ms": round(t16, 4),
                     "bf16_tflops": round(fl / t16 / 1e9, 1), "quantize_ms": round(tq, 4)}
        del a, a8, c8, c16
    if "--forward" in sys.argv:
        from mst.models import DinoV2ClassifierSlice
        src = synth.synth_volume((4, 1, 64, 518, 518), 1).to(torch.bfloat16).cuda()
        for mode in ("bf16", "fp8"):
            model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=mode)
            model.load_state_dict(synth.synth_state_dict("s", 0))
            model = model.cuda().eval()
            with torch.no_grad():
                t = timeit(lambda: model(src), n=5, warm=2)
            out["forward_" + mode] = {"ms_per_step": round(t, 2), "volumes_per_s": round(4000.0 / t, 1)}
            if mode == "fp8":       # calibrated (static) scales: producers write e4m3, nothing is scanned
                model.calibrate_fp8(src)
                with torch.no_grad():
                    t = timeit(lambda: model(src), n=5, warm=2)
                out["forward_fp8_calibrated"] = {"ms_per_step": round(t, 2), "volumes_per_s": round(4000.0 / t, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
