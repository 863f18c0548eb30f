#!/usr/bin/env python3
"""Timings that are NOT the headline, for the record: the training steps (forward + backward through the
HIP kernels + AdamW) of both model families, and the ViT-B forward on the unfused path.  One JSON line per case."""
import json
import sys
import time
import warnings
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "new-vit_amd")]
warnings.simplefilter("ignore")
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice, ResNetSliceTrans


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def train_case(name, model, shape, n=5):
    model = model.cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
    src = synth.synth_volume(shape, 3).cuda()
    tgt = torch.arange(shape[0]).cuda() % 2

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(src), tgt)
        loss.backward()
        opt.step()

    def fwd():
        with torch.no_grad():
            model(src)
    torch.cuda.reset_peak_memory_stats()
    ms = timed(step, n)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    model.eval()
    ms_f = timed(fwd, n)
    print(json.dumps({"case": name, "shape": list(shape), "train_step_ms": round(ms, 2), "eval_forward_ms": round(ms_f, 2),
                      "volumes_per_s_training": round(shape[0] / ms * 1e3, 2), "peak_GiB": round(peak, 2)}), flush=True)


def main():
    dino_shapes = ((1, 1, 16, 224, 224),) if "--only-dino-c1" in sys.argv else ((2, 1, 32, 224, 224),) if "--only-dino-2x32" in sys.argv else ((1, 1, 16, 224, 224), (2, 1, 32, 224, 224))
    for shape in (() if "--only-resnet" in sys.argv else dino_shapes):
        for prec in (("fp32", "bf16", "fp16") if "--mixed" in sys.argv else ("fp16",) if "--fp16" in sys.argv else ("fp32",)):
            m = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, train_precision=prec)
            m.load_state_dict(synth.synth_state_dict("s", 0))
            train_case(f"DinoV2ClassifierSlice training step ({prec} linear products, HIP backward)", m, shape)
    if "--only-dino-c1" in sys.argv or "--only-dino-2x32" in sys.argv:
        return
    for shape in (() if "--c3-only" in sys.argv else ((2, 1, 32, 224, 224),)):
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=34)
        m.load_state_dict(synth.synth_resnet_state_dict(0, 34, 2), strict=True)
        train_case("ResNetSliceTrans(resnet34) training step (fp32 HIP backward)", m, shape)
    if "--c3" in sys.argv:                               # BASELINE configs[3] per-GPU shape: one LIDC-shaped 128 x 512 x 512 volume
        for prec in (("fp32", "bf16", "fp16") if "--mixed" in sys.argv else ("fp16",) if "--fp16" in sys.argv else ("fp32",)):
            m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=34, train_precision=prec)
            m.load_state_dict(synth.synth_resnet_state_dict(0, 34, 2), strict=True)
            train_case(f"ResNetSliceTrans(resnet34) training step ({prec} convolutions), BASELINE configs[3] shape", m, (1, 1, 128, 512, 512), n=3)
            del m
            torch.cuda.empty_cache()
        return
    if "--only-resnet" in sys.argv:
        return
    # ViT-B forward at the bench batch (unfused path: LayerNorm + four GEMMs + attention per block)
    m = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, model_size="b", compute_dtype="bf16")
    m.load_state_dict(synth.synth_state_dict("b", 0))
    m = m.cuda().eval()
    src = torch.randn(4, 1, 64, 518, 518, device="cuda", dtype=torch.bfloat16)

    def fwd():
        with torch.no_grad():
            m(src)
    ms = timed(fwd, 5)
    N, E, depth = 1370, 768, 12
    fl = 256 * (2 * 1369 * E * 588 + depth * (24 * N * E * E + 4 * N * N * E))
    print(json.dumps({"case": "ViT-B/14 forward, 4 x 64 x 518^2 bf16 (unfused blocks)", "ms": round(ms, 2), "volumes_per_s": round(4 / ms * 1e3, 2),
                      "tflops": round(fl / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
