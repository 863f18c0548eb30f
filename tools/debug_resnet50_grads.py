#!/usr/bin/env python3
"""Repeat the resnet50 training step and list the worst gradients per run (diagnosis of a flaky bar)."""
import sys, warnings
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "new-vit_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from mst import synth
from mst.models import ResNetSliceTrans
from test_resnet_gpu import _oracle_step
seed, shape = 53, (2, 1, 3, 64, 64)
sd = synth.synth_resnet_state_dict(seed, 50, 2)
src = synth.synth_volume(shape, seed + 1)
mask = torch.zeros(2, 3, dtype=torch.bool); mask[1, -1] = True
target = torch.tensor([1, 0])
_, _, ref_grads, _ = _oracle_step(sd, src, mask, target, 50, torch.float64)
for run in range(6):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=50)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    loss = torch.nn.functional.cross_entropy(m(src, src_key_padding_mask=mask), target.cuda())
    loss.backward()
    errs = {k: float((p.grad.cpu().double() - ref_grads[k]).abs().max()) / max(float(ref_grads[k].abs().max()), 1e-30) for k, p in m.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    tail = {k: round(v, 5) for k, v in errs.items() if k.startswith(("linear.", "cls_token")) or "slice_fusion.norm" in k or "layer4.2.bn3" in k or "layer4.2.conv3" in k}
    print(run, [(k, round(v, 4)) for k, v in top], tail)
