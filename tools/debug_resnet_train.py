"""Per-parameter gradient error of the ResNetSliceTrans training step (HIP) against torch.autograd through the oracle in fp64,
next to the oracle's own fp32-vs-fp64 noise (the step is ill-conditioned: train-mode BatchNorm over few samples + ReLU flips)."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "new-vit_amd"), ROOT]
warnings.simplefilter("ignore")
import torch

from mst import synth
from mst.models import ResNetSliceTrans
from oracle import resnet_oracle as R


def oracle_step(sd, src, mask, target, model, dt):
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd.update(leaves)
    out = R.forward_slice_trans(sd, src.to(dt), mask, model, train=True)
    torch.nn.functional.cross_entropy(out["logits"], target).backward()
    return {k: v.grad for k, v in leaves.items()}


def main():
    for shape, masked, model in (((2, 1, 4, 96, 64), True, 34), ((2, 1, 3, 64, 64), False, 34), ((2, 1, 4, 128, 128), False, 34)):
        sd = synth.synth_resnet_state_dict(41, model, 2)
        src = synth.synth_volume(shape, 42)
        mask = None
        if masked:
            mask = torch.zeros(shape[0], shape[2], dtype=torch.bool)
            mask[-1, -1:] = True
        target = torch.tensor([1, 0])
        g32 = oracle_step(sd, src, mask, target, model, torch.float32)
        g64 = oracle_step(sd, src, mask, target, model, torch.float64)
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=model)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        torch.nn.functional.cross_entropy(m(src, src_key_padding_mask=mask), target.cuda()).backward()
        rows = []
        for k, p in m.named_parameters():
            n = float(g64[k].norm()) + 1e-300
            rows.append((float((p.grad.cpu().double() - g64[k]).norm()) / n, float((g32[k].double() - g64[k]).norm()) / n, k))
        print(f"== {shape} masked={masked} resnet{model}: worst HIP {max(r[0] for r in rows):.2e}, worst oracle-fp32 {max(r[1] for r in rows):.2e}")
        for i, (a, b, k) in enumerate(rows):
            print(f"  {a:.2e} {b:.2e} {k}")


if __name__ == "__main__":
    main()
