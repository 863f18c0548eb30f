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


def backward_rooflines():
    """`roofline` lines for the dominant BACKWARD kernels of the two training steps: algorithmic FLOPs of one launch
    / its HIP-event time against the dense MFMA peak of the operand type (bf16 / fp16 2.5 PF, fp32 MFMA 157 TF at the nominal clock)."""
    from mst import hip

    def ev(fn, n=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def line(case, kernel, flops, ms, peak_tf, dtype):
        tf = flops / ms / 1e9
        print(json.dumps({"case": case, "roofline": {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 1), "peak": peak_tf, "unit": "TFLOP/s",
                                                     "frac": round(tf / peak_tf, 3), "flops_per_launch": flops, "avg_launch_ms": round(ms, 4), "dtype": dtype}}),
              flush=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    # (1) MST-DINOv2 mixed step, 2 x 32 x 224^2 = 16,448 tokens: d weight of fc1 (N 1536, K 384), the largest of the four per block
    M, N, K = 16448, 1536, 384
    dY, X = torch.randn(M, N, device="cuda", generator=g), torch.randn(M, K, device="cuda", generator=g)
    for dt, nm in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        d16, x16 = hip.cvt16(dY, dt), hip.cvt16(X, dt)
        sp, kc = 14, 1216
        at, xt = hip.cvt16(dY, dt, transpose=True, rows_pad=sp * kc), hip.cvt16(X, dt, transpose=True, rows_pad=sp * kc)
        line("DINOv2 mixed step 2x32x224^2: d weight fc1 = dY^T . X (transposed images, split-K 128x128x64 GEMM)", "gemm16_kernel (split-K)", 2.0 * M * N * K,
             ev(lambda: hip.gemm16_splitk(at, xt, sp)), 2500.0, nm)
        line("same product through the transposing-read kernel", "wgrad16_kernel", 2.0 * M * N * K, ev(lambda: hip.conv_wgrad(d16, x16.view(M, 1, 1, K), 1, 1, 0)), 2500.0, nm)
        w16 = hip.cvt16(torch.randn(N, K, device="cuda", generator=g), dt, transpose=True)
        line("DINOv2 mixed step: d input of fc1 = dY . W", "gemm16 family", 2.0 * M * N * K, ev(lambda: hip.gemm(d16, w16, None, out_dtype=torch.float32)), 2500.0, nm)
    dW = torch.empty(N, K, device="cuda")
    line("DINOv2 fp32 step: d weight fc1 (strided fp32 MFMA GEMM, 16 slabs)", "gemm_ex_kernel<2,2>", 2.0 * M * N * K,
         ev(lambda: hip.gemm_ex(dY, X, torch.empty(16, N * K, device="cuda"), N, K, M // 16, sa=(1, N), sb=(K, 1), sc=(K, 1), nb=(16, 1), ba=(M // 16 * N, 0),
                                bb=(M // 16 * K, 0), bc=(N * K, 0))), 157.0, "fp32")
    # (2) ResNet-34 step at the configs[3] per-GPU shape (128 images of 512^2): the layer-1 3x3 convolution (64 -> 64 channels at 128 x 128)
    n, H, C, k = 128, 128, 64, 3
    x = torch.randn(n, H, H, C, device="cuda", generator=g)
    dz = torch.randn(n * H * H, C, device="cuda", generator=g)
    fl = 2.0 * n * H * H * C * C * k * k
    line("ResNet-34 fp32 step 128x512^2: d weight of a layer-1 convolution (implicit GEMM)", "wgrad32_kernel", fl, ev(lambda: hip.conv_wgrad(dz, x, k, 1, 1), 5), 157.0, "fp32")
    x16, dz16 = hip.cvt16(x.view(-1, C), torch.bfloat16).view(x.shape), hip.cvt16(dz, torch.bfloat16)
    line("ResNet-34 mixed step: the same d weight on bf16 operands (transposing LDS reads)", "wgrad16_kernel", fl, ev(lambda: hip.conv_wgrad(dz16, x16, k, 1, 1), 5), 2500.0, "bf16")
    wt = torch.randn(C, k * k * C, device="cuda", generator=g).to(torch.bfloat16)
    line("ResNet-34 mixed step: d input of that convolution (implicit GEMM, gathered LDS-DMA)", "conv16_kernel", fl, ev(lambda: hip.conv_dgrad(dz16.view(n, H, H, C), wt, k, 1, 1, H, H), 5), 2500.0, "bf16")


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
    if "--rooflines" in sys.argv:
        backward_rooflines()
        return
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
