"""Implicit-GEMM convolution (mst_conv_gemm) against mst_im2col_nhwc + mst_gemm at the ResNet-34 layer shapes of BASELINE configs[3]
(4 volumes x 32 slices of 224 x 224 -> 128 images), and the end-to-end ResNetSliceTrans forward either way.  GPU box: python tools/bench_conv.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "new-vit_amd"))
from mst import hip  # noqa: E402


def timeit(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / it * 1e3


def main():
    n = 128
    if "--only-model" not in sys.argv:
        layers(n)
    model()


def layers(n):
    print(f"{'layer':34s} {'im2col+gemm ms':>15s} {'implicit ms':>12s} {'TFLOP/s':>8s}")
    for name, cin, cout, k, s, hw in [("layer1 3x3 64->64 56x56", 64, 64, 3, 1, 56), ("layer2 3x3 s2 64->128", 64, 128, 3, 2, 56),
                                      ("layer2 3x3 128->128 28x28", 128, 128, 3, 1, 28), ("layer3 3x3 256->256 14x14", 256, 256, 3, 1, 14),
                                      ("layer4 3x3 512->512 7x7", 512, 512, 3, 1, 7), ("layer2 1x1 s2 64->128", 64, 128, 1, 2, 56)]:
        pad = 1 if k == 3 else 0
        x = torch.randn(n, hw, hw, cin, device="cuda")
        K = k * k * cin
        w = torch.randn(cout, K, device="cuda") * 0.02
        b = torch.randn(cout, device="cuda")
        t0 = timeit(lambda: hip.gemm(hip.im2col_nhwc(x, k, k, s, pad, K), w, b, epilogue=hip.EPI_BIAS_RELU))
        t1 = timeit(lambda: hip.conv_gemm(x, w, b, k, k, s, pad, epilogue=hip.EPI_BIAS_RELU))
        ho = (hw + 2 * pad - k) // s + 1
        fl = 2.0 * n * ho * ho * cout * K
        print(f"{name:34s} {t0:15.3f} {t1:12.3f} {fl / t1 / 1e9:8.1f}")


def model():
    import warnings
    from mst import synth
    from mst.models import ResNetSliceTrans
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False)
    m.load_state_dict(synth.synth_resnet_state_dict(0, 34, 2), strict=True)
    m = m.cuda().eval()
    src = torch.randn(4, 1, 32, 224, 224, device="cuda")
    for mode in (("0",) if "--only-model" in sys.argv else ("1", "0")):
        os.environ["MST_CONV_IM2COL"] = mode
        with torch.no_grad():
            t = timeit(lambda: m(src), it=5)
        print(f"ResNetSliceTrans(34) forward [4,1,32,224,224]  MST_CONV_IM2COL={mode}: {t:.2f} ms  peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
        torch.cuda.reset_peak_memory_stats()
    os.environ["MST_CONV_IM2COL"] = "0"
    ref = None
    for shape in ((4, 1, 32, 224, 224), (1, 1, 128, 512, 512)):
        src = torch.randn(*shape, device="cuda")
        for cdt in ("fp32", "bf16", "fp16"):
            m.compute_dtype_name = cdt
            with torch.no_grad():
                out = m(src)
                t = timeit(lambda: m(src), it=5)
            ref = out if cdt == "fp32" else ref
            print(f"ResNetSliceTrans(34) forward {list(shape)} compute_dtype={cdt}: {t:.2f} ms  peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  "
                  f"max |dlogits| vs fp32 {float((out - ref).abs().max()):.2e}")
            torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
