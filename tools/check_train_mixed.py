"""Gradient accuracy of the mixed-precision training step (train_precision bf16 / fp16) against the fp32 step of the same model on the same
inputs: per parameter max |dP - dP32| / max |dP32|, summarised.  GPU box: python tools/check_train_mixed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "new-vit_amd"))
from mst import synth  # noqa: E402
from mst.models import DinoV2ClassifierSlice  # noqa: E402


def grads(prec, shape, scale=1.0):
    m = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, train_precision=prec)
    m.load_state_dict(synth.synth_state_dict("s", 0))
    m = m.cuda().train()
    src = synth.synth_volume(shape, 3).cuda()
    tgt = torch.arange(shape[0]).cuda() % 2
    loss = torch.nn.functional.cross_entropy(m(src), tgt)
    (loss * scale).backward()
    return float(loss), {k: (p.grad / scale).clone() for k, p in m.named_parameters() if p.grad is not None}


def main():
    shape = (1, 1, 8, 224, 224)
    l32, g32 = grads("fp32", shape)
    for prec, scale in (("bf16", 1.0), ("fp16", 1.0), ("fp16", 65536.0)):
        l, g = grads(prec, shape, scale)
        errs = sorted(((float((g[k] - g32[k]).abs().max() / g32[k].abs().max().clamp_min(1e-30)), k) for k in g32), reverse=True)
        rel = sorted((float((g[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)) for k in g32))
        print({"precision": prec, "loss_scale": scale, "loss": round(l, 6), "loss_fp32": round(l32, 6), "worst_max_norm_err": round(errs[0][0], 4), "worst_param": errs[0][1],
               "median_max_norm_err": round(errs[len(errs) // 2][0], 4), "median_rel_l2": round(rel[len(rel) // 2], 4), "worst_rel_l2": round(rel[-1], 4)})


def resnet():
    import warnings
    from mst.models import ResNetSliceTrans
    shape = (2, 1, 4, 96, 64)
    src = synth.synth_volume(shape, 42).cuda()
    tgt = torch.tensor([1, 0]).cuda()
    sd = synth.synth_resnet_state_dict(41, 34, 2)

    def grads(prec):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=34, train_precision=prec)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        logits = m(src)
        loss = torch.nn.functional.cross_entropy(logits, tgt)
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    l32, g32 = grads("fp32")
    l32b, g32b = grads("fp32")
    for prec, (l, g) in (("fp32 again", (l32b, g32b)), ("bf16", grads("bf16")), ("fp16", grads("fp16"))):
        rel = sorted(float((g[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)) for k in g32)
        glob = (sum(float((g[k] - g32[k]).square().sum()) for k in g32) / sum(float(g32[k].square().sum()) for k in g32)) ** 0.5
        print({"model": "ResNetSliceTrans(34)", "precision": prec, "loss": round(l, 6), "loss_fp32": round(l32, 6), "median_rel_l2": round(rel[len(rel) // 2], 4),
               "worst_rel_l2": round(rel[-1], 4), "global_rel_l2": round(glob, 4)})


if __name__ == "__main__":
    if "--resnet" not in sys.argv:
        main()
    resnet()
