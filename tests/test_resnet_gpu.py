"""ResNet / ResNetSliceTrans inference, training step and Grad-CAM++ on the HIP path (reference mst/models/resnet.py:27-243).
The across-slice half is checked against a fixture the reference's own TransformerEncoderLayer(512, nhead 16) produced
(tests/golden/resnet_fusion.npz); the torchvision backbone is not in the reference tree, so its parity is against the restated
architecture of oracle/resnet_oracle.py (UNPINNED), fp32, relative 1e-4."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from mst import synth

pytestmark = pytest.mark.gpu


def _model(seed, **kw):
    import warnings
    from mst.models import ResNetSliceTrans
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, **kw)
    m.load_state_dict(synth.synth_resnet_state_dict(seed, 34, 2), strict=True)
    return m.cuda().eval()


def test_slice_fusion_half_matches_reference_fixture():
    g = load_golden("resnet_fusion")
    m = _model(int(g["seed"]))
    emb = torch.from_numpy(g["emb"]).cuda()
    B, D, E = emb.shape
    mask = torch.from_numpy(g["src_key_padding_mask"])
    m._last_shape = (B, D)
    for tag, mk in (("", None), ("_masked", mask)):
        logits = m.fuse(emb.reshape(B * D, E), B, D, mk, save_attn=True)
        assert np.abs(logits.cpu().numpy() - g["logits" + tag]).max() < 1e-4
        assert rel_l2(m.attention_maps_slice[-1].cpu(), g["slice_map" + tag]) < 1e-4
        for _ in range(2):
            assert rel_l2(m.get_slice_attention().cpu(), g["slice_attention" + tag]) < 1e-4


@pytest.mark.parametrize("shape,masked", [((1, 1, 3, 64, 64), False), ((2, 1, 4, 96, 80), True), ((1, 1, 2, 224, 224), False)])
def test_resnet_slice_trans_forward_matches_oracle(shape, masked):
    from oracle import resnet_oracle as R
    seed = 31
    m = _model(seed, chunk_images=3)
    sd = synth.synth_resnet_state_dict(seed, 34, 2)
    src = synth.synth_volume(shape, seed + 100)
    mask = None
    if masked:
        mask = torch.zeros(shape[0], shape[2], dtype=torch.bool)
        mask[-1, -2:] = True
    with torch.no_grad():
        ref = R.forward_slice_trans(sd, src, mask)
        emb = m._features(src.cuda().float().reshape(-1, shape[3], shape[4], 1).contiguous(), True)
        logits = m(src, src_key_padding_mask=mask)
    assert rel_l2(emb.cpu(), ref["emb"]) < 1e-4
    assert float((logits.cpu() - ref["logits"]).abs().max()) < 1e-3 * max(1.0, float(ref["logits"].abs().max()))
    assert logits.shape == (shape[0], 2)


def test_plain_resnet_with_fc_and_error_behaviour():
    import warnings
    from mst.models import ResNet, ResNetSliceTrans
    from oracle import resnet_oracle as R
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNet(in_ch=3, out_ch=2, spatial_dims=2, pretrained=False, model=34)       # tests/models/test_resnet.py of the reference
    sd = synth.synth_resnet_state_dict(5, 34, 2, slice_trans=False, fc_out=2)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(synth.hash_normal((2, 3, 96, 96), 6, 1))
    with torch.no_grad():
        got = m(source=x)
        ref = R.resnet_features(sd, x)
    assert got.shape == (2, 2) and float((got.cpu() - ref).abs().max()) < 1e-3 * float(ref.abs().max())
    with pytest.raises(NotImplementedError):
        m(x)                                                                              # gradients through eval-mode BatchNorm
    with pytest.raises(NotImplementedError):
        ResNet(in_ch=1, out_ch=2, spatial_dims=3)
    st = _model(7)
    with pytest.raises(RuntimeError, match="channels"), torch.no_grad():
        st(torch.zeros(1, 2, 3, 32, 32))                                                  # only gray volumes fit the 3-channel stem


# ---- bottleneck ResNets (reference resnet.py:44-50 takes any torchvision model; emb_ch 2048 for model > 34, resnet.py:152) -------
def test_resnet50_slice_trans_forward_training_step_and_gradcam():
    """model=50 (torchvision Bottleneck v1.5, 2048-wide slice embeddings, 16 heads of 128): forward against the oracle, one training
    step (every gradient: the 10 % smoke bar of the BasicBlock test, last stage 5e-3, loss / logits / running statistics tight),
    and the Grad-CAM++ map of a plain resnet50.  VERDICT r2 item 7c: these models used to raise."""
    import warnings
    from mst.models import ResNet, ResNetSliceTrans
    from oracle import resnet_oracle as R
    seed, shape = 53, (2, 1, 3, 64, 64)
    sd = synth.synth_resnet_state_dict(seed, 50, 2)
    assert sd["model.layer4.2.conv3.weight"].shape == (2048, 512, 1, 1) and sd["cls_token"].shape[-1] == 2048
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=50)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    src = synth.synth_volume(shape, seed + 1)
    mask = torch.zeros(2, 3, dtype=torch.bool)
    mask[1, -1] = True
    with torch.no_grad():
        ref = R.forward_slice_trans(sd, src, mask, model=50)
        emb = m._features(src.cuda().float().reshape(-1, 64, 64, 1).contiguous(), True)
        logits = m(src, src_key_padding_mask=mask)
    assert emb.shape == (6, 2048) and rel_l2(emb.cpu(), ref["emb"]) < 1e-4
    assert float((logits.cpu() - ref["logits"]).abs().max()) < 1e-3 * max(1.0, float(ref["logits"].abs().max()))
    # training step
    target = torch.tensor([1, 0])
    ref_logits, ref_loss, ref_grads, ref_sd = _oracle_step(sd, src, mask, target, 50, torch.float64)
    m.train()
    out = m(src, src_key_padding_mask=mask)
    loss = torch.nn.functional.cross_entropy(out, target.cuda())
    loss.backward()
    assert float((out.detach().cpu() - ref_logits).abs().max()) < 1e-3 * max(1.0, float(ref_logits.abs().max()))
    assert float(loss) == pytest.approx(ref_loss, rel=1e-3, abs=1e-4)
    err = []
    for k, prm in m.named_parameters():
        assert prm.grad is not None and float(ref_grads[k].abs().max()) > 0, k
        err.append(rel_l2(prm.grad.cpu(), ref_grads[k]))
    assert max(err) < 0.1 and float(np.median(err)) < 0.05, (max(err), float(np.median(err)))
    # tight bar on what sits behind NO further ReLU decision (head, final norm): run to run the atomically summed backbone moves the
    # 2048-wide embeddings by ~1e-4, which flips a ReLU of the slice transformer's FFN or of layer4 now and then (24 BatchNorm samples
    # at this size: up to 20 % on single tensors in one run of six, tools/debug_resnet50_grads.py) -- those stay under the smoke bar
    last = [k for k in ref_grads if k.startswith(("linear.", "slice_fusion.norm."))]
    errs_last = {k: float((dict(m.named_parameters())[k].grad.cpu().double() - ref_grads[k]).abs().max()) / max(float(ref_grads[k].abs().max()), 1e-30) for k in last}
    assert max(errs_last.values()) < 2e-3, errs_last
    # directional derivatives against the fp64 oracle loss (see the BasicBlock test): insensitive to single flips
    names = [k for k, _ in m.named_parameters()]
    gen = torch.Generator().manual_seed(11)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}

    def loss_at(shift):
        s2 = dict(sd64)
        for k, dv in shift.items():
            s2[k] = sd64[k] + dv
        with torch.no_grad():
            return float(torch.nn.functional.cross_entropy(R.forward_slice_trans(s2, src.double(), mask, 50, train=True)["logits"], target))

    dd_err = []
    for trial in range(4):
        pick = [k for k in names if float(torch.rand((), generator=gen)) < 0.34] or names[:1]
        v = {k: torch.randn(sd64[k].shape, generator=gen, dtype=torch.float64) * float(sd64[k].abs().mean() + 1e-3) for k in pick}
        eps = 1e-7
        fd = (loss_at({k: eps * d for k, d in v.items()}) - loss_at({k: -eps * d for k, d in v.items()})) / (2 * eps)
        dd_hip = sum(float((dict(m.named_parameters())[k].grad.cpu().double() * v[k]).sum()) for k in pick)
        dd_err.append(abs(dd_hip - fd) / max(abs(fd), 1e-6))
    assert float(np.median(dd_err)) < 2e-2 and max(dd_err) < 8e-2, dd_err
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_l2(v.cpu(), ref_sd[k]) < 1e-4, k
    # plain resnet50 with fc + Grad-CAM++ of the last ReLU output
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r50 = ResNet(in_ch=3, out_ch=2, spatial_dims=2, pretrained=False, model=50, chunk_images=2)
    sd2 = synth.synth_resnet_state_dict(9, 50, 2, slice_trans=False, fc_out=2)
    r50.load_state_dict(sd2, strict=True)
    r50 = r50.cuda().eval()
    x = torch.from_numpy(synth.hash_normal((3, 3, 96, 64), 10, 1))
    with torch.no_grad():
        o = r50(x, save_attn=True)
    assert float((o.cpu() - R.resnet_features(sd2, x, 50)).abs().max()) < 1e-3 * max(1.0, float(o.abs().max()))
    cam, cref = r50.get_attention_maps(), R.gradcampp_last(sd2, x, 50)
    assert cam.shape == cref.shape == (3, 1, 3, 2) and float((cam.cpu() - cref).abs().max()) < 1e-3


# ---- Grad-CAM++ (resnet.py:55-118) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("with_fc", [True, False])
def test_gradcampp_last_map_matches_autograd_oracle(with_fc):
    import warnings
    from mst.models import ResNet
    from oracle import resnet_oracle as R
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNet(in_ch=3, out_ch=2, spatial_dims=2, pretrained=False, model=34, **({} if with_fc else {"emb_ch": None}), chunk_images=2)
    sd = synth.synth_resnet_state_dict(9, 34, 2, slice_trans=False, fc_out=2 if with_fc else None)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(synth.hash_normal((3, 3, 128, 96), 10, 1))
    with torch.no_grad():
        out = m(x, save_attn=True)
    ref = R.gradcampp_last(sd, x)
    got = m.get_attention_maps()
    assert got.shape == ref.shape == (3, 1, 4, 3)
    assert float((got.cpu() - ref).abs().max()) < 1e-3
    assert float(got.max()) == pytest.approx(1.0) and float(got.min()) == 0.0
    assert out.shape == (3, 2 if with_fc else 512)


def test_slice_trans_attention_maps_combine_slice_attention_and_gradcam():
    from oracle import resnet_oracle as R
    m = _model(12, chunk_images=2)
    sd = synth.synth_resnet_state_dict(12, 34, 2)
    src = synth.synth_volume((1, 1, 3, 96, 96), 13)
    with torch.no_grad():
        m(src, save_attn=True)
        got = m.get_attention_maps()
    bsd = {k: v for k, v in sd.items() if k.startswith("model.")}
    cam = R.gradcampp_last(bsd, src.repeat(1, 3, 1, 1, 1).permute(0, 2, 1, 3, 4).reshape(3, 3, 96, 96))
    ref = R.forward_slice_trans(sd, src)
    sa = ref["slice_map"][:, :, 0, 1:]
    sa = (sa / sa.sum(dim=-1, keepdim=True)).mean(dim=1).reshape(-1)
    want = sa[:, None, None, None] * cam
    assert got.shape == want.shape == (3, 1, 3, 3)
    assert float((got.cpu() - want).abs().max()) < 1e-3 * float(want.abs().max())


# ---- training step (BASELINE configs[3]; reference base_model.py:148-181 over torch.autograd) ------------------------------
def _oracle_step(sd, src, mask, target, model=34, dt=torch.float32):
    from oracle import resnet_oracle as R
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    src = src.to(dt)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd.update(leaves)
    with torch.enable_grad():
        out = R.forward_slice_trans(sd, src, mask, model, train=True)
        loss = torch.nn.functional.cross_entropy(out["logits"], target)
        loss.backward()
    return out["logits"].detach(), float(loss.detach()), {k: v.grad for k, v in leaves.items()}, sd


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw,residual,relu,sum_in", [
    (64, 64, 3, 1, 1, (12, 10), True, True, False), (64, 128, 3, 2, 1, (11, 14), False, True, False),
    (64, 128, 1, 2, 0, (12, 10), False, False, False), (3, 64, 7, 2, 3, (40, 36), False, True, True)])
def test_conv_batchnorm_unit_forward_and_backward_match_torch(cin, cout, k, stride, pad, hw, residual, relu, sum_in):
    """One convolution + train-mode BatchNorm (+ residual, ReLU) unit: the well-conditioned check of the training step's kernels
    (im2col GEMM, mst_batchnorm_train / _bwd, split dW product, col2im) against torch.autograd on the same operands, fp64."""
    import torch.nn.functional as F
    from mst import train_resnet as T
    from mst.models.resnet import _BN, _Conv
    n = 5
    g = torch.Generator().manual_seed(k * 100 + cout)
    conv, bn = _Conv(cin, cout, k), _BN(cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        bn.weight.copy_(1 + 0.2 * torch.randn(cout, generator=g))
        bn.bias.copy_(0.2 * torch.randn(cout, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(cout, generator=g))
        bn.running_var.copy_(1 + 0.1 * torch.rand(cout, generator=g))
    xin = torch.randn(n, 1 if sum_in else cin, *hw, generator=g)
    Ho, Wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, Ho, Wo, generator=g) if residual else None
    dy = torch.randn(n, cout, Ho, Wo, generator=g)
    # reference: torch ops in fp64
    w64 = conv.weight.detach().double().requires_grad_(True)
    ga, be = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    x64 = xin.double().requires_grad_(True)
    rm, rv = bn.running_mean.double().clone(), bn.running_var.double().clone()
    z = F.conv2d(x64.repeat(1, 3, 1, 1) if sum_in else x64, w64, stride=stride, padding=pad)
    y = F.batch_norm(z, rm, rv, ga, be, True, 0.1, 1e-5)
    if residual:
        y = y + res.double()
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    # HIP
    conv, bn = conv.cuda(), bn.cuda()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    yh, rec = T._conv_bn_fwd(nhwc(xin), conv, bn, k, stride, pad, sum_in, nhwc(res).view(-1, cout) if residual else None, relu)
    assert rel_l2(yh.permute(0, 3, 1, 2).cpu(), y.detach()) < 1e-5
    assert rel_l2(bn.running_mean.cpu(), rm) < 1e-5 and rel_l2(bn.running_var.cpu(), rv) < 1e-5
    G = T._Grads()
    dyh = nhwc(dy).view(-1, cout).clone()
    dx = T._conv_bn_bwd(G, rec, dyh, True)
    assert rel_l2(G.by_param[id(conv.weight)].cpu(), w64.grad) < 2e-5
    assert rel_l2(G.by_param[id(bn.weight)].cpu(), ga.grad) < 2e-5
    assert rel_l2(G.by_param[id(bn.bias)].cpu(), be.grad) < 2e-5
    assert rel_l2(dx.permute(0, 3, 1, 2).cpu(), x64.grad) < 2e-5
    if residual:                                        # the masked gradient is what the caller routes to the shortcut
        assert rel_l2(dyh.view(n, Ho, Wo, cout).permute(0, 3, 1, 2).cpu(), dy.double() * (y.detach() > 0)) < 1e-6


@pytest.mark.parametrize("n,cin,cout,k,stride,pad,hw,kextra", [
    (5, 64, 64, 3, 1, 1, (16, 15), 0), (3, 64, 128, 3, 2, 1, (41, 37), 0), (5, 128, 256, 1, 2, 0, (35, 27), 0), (2, 16, 96, 3, 1, 1, (7, 5), 16),
    (1, 256, 64, 1, 1, 0, (3, 3), 0), (2, 80, 200, 3, 2, 1, (53, 49), 0), (7, 512, 512, 3, 1, 1, (2, 2), 0), (9, 16, 40, 3, 1, 1, (13, 11), 32)])
@pytest.mark.parametrize("epi", ["bias", "relu", "residual", "residual_relu"])
def test_implicit_gemm_convolution_equals_im2col_gemm_and_torch(n, cin, cout, k, stride, pad, hw, kextra, epi):
    """mst_conv_gemm gathers the A operand the im2col matrix would hold, in the same k order and tiles: bit-identical to
    mst_im2col_nhwc + mst_gemm (where that runs the same 128 x 128-tile kernel: more than 1,024 rows) for every epilogue, ragged row
    counts, padding rows/columns, a zero-padded K; and equal to torch's fp64 conv2d to fp32 accuracy."""
    import torch.nn.functional as F
    from mst import hip
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + k)
    x = torch.randn(n, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    K = k * k * cin
    kpad = K + kextra
    wg = torch.zeros(cout, kpad)
    wg[:, :K] = w.permute(0, 2, 3, 1).reshape(cout, K)
    Ho, Wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
    res = torch.randn(n * Ho * Wo, cout, generator=g)
    e = {"bias": hip.EPI_BIAS, "relu": hip.EPI_BIAS_RELU, "residual": hip.EPI_RESIDUAL, "residual_relu": hip.EPI_RESIDUAL_RELU}[epi]
    xh, wh, bh = x.permute(0, 2, 3, 1).contiguous().cuda(), wg.cuda(), b.cuda()
    o1 = res.cuda().clone() if epi.startswith("residual") else None
    o2 = res.cuda().clone() if epi.startswith("residual") else None
    got = hip.conv_gemm(xh, wh, bh, k, k, stride, pad, epilogue=e, out=o1)
    via = hip.gemm(hip.im2col_nhwc(xh, k, k, stride, pad, kpad), wh, bh, epilogue=e, out=o2)
    assert got.shape == (n * Ho * Wo, cout)
    if n * Ho * Wo > 1024:                              # same kernel, same k order, same tiles
        assert torch.equal(got, via)
    else:                                               # mst_gemm takes its 32 x 32-tile kernel here: another summation order
        assert rel_l2(got.cpu(), via.cpu()) < 1e-6
    want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(-1, cout)
    want = F.relu(want) if epi == "relu" else want + res.double() if epi == "residual" else F.relu(want + res.double()) if epi == "residual_relu" else want
    assert rel_l2(got.cpu(), want) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,k,stride,pad,hw,epi,f32out", [
    (3, 64, 64, 3, 1, 1, (16, 15), "relu", False), (2, 64, 128, 3, 2, 1, (41, 37), "bias", False), (5, 128, 256, 1, 2, 0, (35, 27), "bias", True),
    (2, 64, 200, 3, 1, 1, (9, 7), "relu", True), (1, 256, 64, 1, 1, 0, (3, 3), "residual_relu", False), (2, 128, 128, 3, 1, 1, (20, 13), "residual_relu", False),
    (7, 512, 512, 3, 1, 1, (2, 2), "relu", False), (2, 192, 64, 1, 1, 0, (30, 30), "relu", True)])
def test_implicit_gemm_convolution_on_16_bit_operands(dt, n, cin, cout, k, stride, pad, hw, epi, f32out):
    """mst_conv_gemm16 (the 16-bit inference path of the backbone): gathered A operand with a zero page for padding taps and rows beyond M,
    64- and 128-column tiles, partial column tiles, every epilogue and output type, against fp64 conv2d of the SAME rounded operands."""
    import torch.nn.functional as F
    from mst import hip
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + k)
    x = torch.randn(n, cin, *hw, generator=g).to(tdt)
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(tdt)
    b = torch.randn(cout, generator=g)
    Ho, Wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
    res = torch.randn(n * Ho * Wo, cout, generator=g).to(tdt)
    e = {"bias": hip.EPI_BIAS, "relu": hip.EPI_BIAS_RELU, "residual_relu": hip.EPI_RESIDUAL_RELU}[epi]
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    wh = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
    out = res.cuda().clone() if epi == "residual_relu" else None
    got = hip.conv_gemm16(xh, wh, b.cuda(), k, k, stride, pad, epilogue=e, out=out, out_dtype=torch.float32 if f32out else None)
    assert got.shape == (n * Ho * Wo, cout) and got.dtype == (torch.float32 if f32out else tdt)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(-1, cout)
    want = F.relu(want) if epi == "relu" else F.relu(want + res.double()) if epi == "residual_relu" else want
    tol = 1e-5 if f32out else {"bf16": 4e-3, "fp16": 5e-4}[dt]                      # 16-bit outputs: one rounding of the result
    assert rel_l2(got.float().cpu(), want) < tol
    with pytest.raises(RuntimeError, match="multiple of 64"):
        hip.conv_gemm16(torch.zeros(1, 4, 4, 32, dtype=tdt, device="cuda"), torch.zeros(8, 32, dtype=tdt, device="cuda"), None, 1, 1, 1, 0)


@pytest.mark.parametrize("cdt,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
@pytest.mark.parametrize("model", [34, 50])
def test_16_bit_inference_backbone_against_the_fp32_one(cdt, tol, model):
    """ResNetSliceTrans(compute_dtype=bf16 / fp16): 16-bit NHWC activations and MFMA operands in the backbone (fp32 accumulation, folded
    BatchNorm bias, pooling, slice transformer and head in fp32) against the exact fp32 path of the same weights: slice embeddings and
    logits within the stated 16-bit bars (measured: see profiles/r04l_resnet_16bit.txt), Grad-CAM++ maps produced and finite."""
    import warnings
    from mst.models import ResNetSliceTrans
    sd = synth.synth_resnet_state_dict(5, model, 2)
    src = synth.synth_volume((2, 1, 5, 96, 80), 6).cuda()
    outs = {}
    for c in ("fp32", cdt):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=model, compute_dtype=c)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            logits = m(src, save_attn=True)
            emb = m._features(src.float().reshape(10, 96, 80, 1).contiguous(), True)
        outs[c] = (logits.float().cpu(), emb.cpu(), m.get_attention_maps().cpu())
    assert rel_l2(outs[cdt][1], outs["fp32"][1]) < tol
    assert float((outs[cdt][0] - outs["fp32"][0]).abs().max()) < tol * max(1.0, float(outs["fp32"][0].abs().max()))
    assert torch.isfinite(outs[cdt][2]).all() and outs[cdt][2].shape == outs["fp32"][2].shape
    with pytest.raises(ValueError):
        ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, compute_dtype="fp8")


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,k,stride,pad,hw", [
    (3, 64, 64, 3, 1, 1, (16, 15)), (2, 64, 128, 3, 2, 1, (41, 37)), (2, 64, 128, 3, 2, 1, (12, 10)), (5, 128, 256, 1, 2, 0, (35, 27)),
    (2, 256, 64, 1, 1, 0, (9, 7)), (2, 128, 128, 3, 1, 1, (20, 13)), (1, 200, 64, 3, 2, 1, (11, 14))])
def test_input_gradient_as_a_convolution(dt, n, cin, cout, k, stride, pad, hw):
    """mst_conv_dgrad: the gradient of a convolution's input computed as a convolution of dz (dilated by the stride) with the flipped,
    transposed weight -- against torch.autograd of F.conv2d on the same (rounded) operands; odd sizes where the last input rows and
    columns fall outside every window included."""
    import torch.nn.functional as F
    from mst import hip
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    g = torch.Generator().manual_seed(n * 100 + cin + cout + k + stride)
    x = torch.randn(n, cin, *hw, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(tdt)
    y = F.conv2d(x, w.double(), stride=stride, padding=pad)
    dz = (torch.randn(y.shape, generator=g)).to(tdt)
    y.backward(dz.double())
    got = hip.conv_dgrad(dz.permute(0, 2, 3, 1).contiguous().cuda(), hip.conv_dgrad_weight(w.cuda(), tdt), k, stride, pad, *hw)
    assert got.shape == (n, *hw, cin) and got.dtype == torch.float32
    assert rel_l2(got.permute(0, 3, 1, 2).cpu(), x.grad) < 2e-6                      # same operands, fp32 accumulation either way


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,k,stride,pad,hw", [
    (3, 64, 64, 3, 1, 1, (16, 15)), (2, 64, 128, 3, 2, 1, (41, 37)), (5, 128, 256, 1, 2, 0, (35, 27)), (2, 256, 64, 1, 1, 0, (9, 7)),
    (9, 128, 192, 3, 1, 1, (20, 13)), (1, 192, 64, 3, 2, 1, (11, 14)), (70, 64, 64, 3, 1, 1, (30, 28)), (6, 512, 512, 3, 1, 1, (7, 7))])
def test_weight_gradient_on_16_bit_operands(dt, n, cin, cout, k, stride, pad, hw):
    """mst_conv_wgrad16: pixel-major 16-bit tiles in LDS, both MFMA operands through the hardware-transposing LDS read, x rows gathered per
    tap with the zero page -- against fp64 autograd of F.conv2d on the SAME rounded operands (fp32 accumulation either way); images narrower
    than a DMA piece (7 x 7), ragged splits, column tiles that end inside the 128-wide tile (K = 576, Cout = 64 / 192)."""
    import torch.nn.functional as F
    from mst import hip
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    g = torch.Generator().manual_seed(n * 100 + cin + cout + k + stride)
    x = torch.randn(n, cin, *hw, generator=g).to(tdt)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) / math.sqrt(cin * k * k)).requires_grad_(True)
    y = F.conv2d(x.double(), w, stride=stride, padding=pad)
    dz = torch.randn(y.shape, generator=g).to(tdt)
    y.backward(dz.double())
    got = hip.conv_wgrad(dz.permute(0, 2, 3, 1).reshape(-1, cout).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(), k, stride, pad)
    assert got.shape == (cout, k * k * cin) and got.dtype == torch.float32
    assert rel_l2(got.view(cout, k, k, cin).permute(0, 3, 1, 2).cpu(), w.grad) < 3e-6


@pytest.mark.parametrize("n,cin,cout,k,stride,pad,hw", [
    (3, 64, 64, 3, 1, 1, (16, 15)), (2, 64, 128, 3, 2, 1, (41, 37)), (5, 128, 256, 1, 2, 0, (35, 27)), (2, 256, 68, 1, 1, 0, (9, 7)),
    (9, 128, 128, 3, 1, 1, (20, 13)), (1, 192, 64, 3, 2, 1, (11, 14)), (70, 64, 64, 3, 1, 1, (30, 28))])
def test_weight_gradient_as_an_implicit_gemm(n, cin, cout, k, stride, pad, hw):
    """mst_conv_wgrad (+ mst_colsum over its partial products): the gradient of a convolution's weight with the B operand gathered from the
    NHWC activation -- against torch.autograd of F.conv2d in fp64; one and many splits, ragged last split, Cout no multiple of 64."""
    import torch.nn.functional as F
    from mst import hip
    g = torch.Generator().manual_seed(n * 100 + cin + cout + k + stride)
    x = torch.randn(n, cin, *hw, generator=g)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) / math.sqrt(cin * k * k)).requires_grad_(True)
    y = F.conv2d(x.double(), w, stride=stride, padding=pad)
    dz = torch.randn(y.shape, generator=g)
    y.backward(dz.double())
    got = hip.conv_wgrad(dz.permute(0, 2, 3, 1).reshape(-1, cout).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(), k, stride, pad)
    assert got.shape == (cout, k * k * cin)
    assert rel_l2(got.view(cout, k, k, cin).permute(0, 3, 1, 2).cpu(), w.grad) < 2e-6


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_16_bit_stem_ops_are_the_fp32_ones_rounded(dt):
    """mst_im2col_nhwc16 = mst_im2col_nhwc rounded to the 16-bit type on the way out; mst_maxpool_nhwc16 = the 3 x 3 / 2 max pool of 16-bit
    activations (a maximum of representable values is representable: bit-exact against torch's max_pool2d)."""
    import torch.nn.functional as F
    from mst import hip
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 23, 18, 1, generator=g).cuda()
    a = hip.im2col_nhwc(x, 7, 7, 2, 3, 64)
    b = hip.im2col_nhwc(x, 7, 7, 2, 3, 64, out_dtype=tdt)
    assert b.dtype == tdt and torch.equal(b, a.to(tdt))
    y = torch.randn(2, 16, 13, 11, generator=g).to(tdt)                               # NCHW for torch
    want = F.max_pool2d(y.float(), 3, 2, 1).to(tdt)
    got = hip.maxpool_nhwc(y.permute(0, 2, 3, 1).contiguous().cuda())
    assert got.dtype == tdt and torch.equal(got.permute(0, 3, 1, 2).cpu(), want)


def test_implicit_gemm_convolution_rejects_what_it_cannot_gather():
    from mst import hip
    x = torch.zeros(1, 4, 4, 3, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 16"):
        hip.conv_gemm(x, torch.zeros(8, 32, device="cuda"), None, 3, 3, 1, 1)
    x = torch.zeros(1, 4, 4, 16, device="cuda")
    with pytest.raises(RuntimeError, match="Kpad"):
        hip.conv_gemm(x, torch.zeros(8, 128, device="cuda"), None, 3, 3, 1, 1)


def test_pool_backward_kernels_match_torch():
    import torch.nn.functional as F
    from mst import hip
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 16, 13, 10, generator=g).double().requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    nhwc = lambda t: t.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
    dx = hip.maxpool_bwd_nhwc(nhwc(x), nhwc(dy))
    assert rel_l2(dx.permute(0, 3, 1, 2).cpu(), x.grad) < 1e-6
    d = torch.randn(3, 16, generator=g)
    da = hip.avgpool_bwd_nhwc(d.cuda(), 20)
    assert torch.equal(da.cpu(), (d / 20)[:, None, :].expand(3, 20, 16))


@pytest.mark.parametrize("shape,masked,model", [((2, 1, 3, 64, 64), False, 18), ((2, 1, 4, 96, 64), True, 34)])
def test_training_step_matches_autograd_of_oracle(shape, masked, model):
    """Whole step: a WIRING check (residual routing, pooling, slice transformer, parameter mapping, running statistics).  The
    step is ill-conditioned: ~3e6 ReLU decisions sit behind 34 layers of fp32 rounding, a handful flip under a 1e-7 perturbation,
    and one flip in layer 4 (train-mode BatchNorm over 48 samples here) moves every upstream gradient by about 1/48.  torch's own
    fp32 autograd is 2e-3 .. 1e-2 away from its fp64 run (profiles/r02t_resnet_train_errors.txt); the HIP step sums in another
    order (atomics: not even run-to-run identical) and has measured 1.6e-3 .. 2.8e-2 on the same inputs.  Hence: fp64 oracle as
    the reference, every gradient within 10 % (a wiring mistake gives O(1) or a missing gradient), logits and loss tight; the
    arithmetic itself is pinned at 2e-5 by the per-unit tests above, which are well-conditioned."""
    import warnings
    from mst.models import ResNetSliceTrans
    seed = 41
    sd = synth.synth_resnet_state_dict(seed, model, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=model)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    src = synth.synth_volume(shape, seed + 1)
    target = torch.tensor([1, 0][:shape[0]])
    mask = None
    if masked:
        mask = torch.zeros(shape[0], shape[2], dtype=torch.bool)
        mask[-1, -1:] = True
    ref_logits, ref_loss, ref_grads, ref_sd = _oracle_step(sd, src, mask, target, model, torch.float64)
    _, _, g32, _ = _oracle_step(sd, src, mask, target, model, torch.float32)
    logits = m(src, src_key_padding_mask=mask)
    loss = torch.nn.functional.cross_entropy(logits, target.cuda())
    loss.backward()
    assert float((logits.detach().cpu() - ref_logits).abs().max()) < 1e-3 * max(1.0, float(ref_logits.abs().max()))
    assert float(loss) == pytest.approx(ref_loss, rel=1e-3, abs=1e-4)
    err, noise = [], []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert float(ref_grads[k].abs().max()) > 0, k
        err.append(rel_l2(p.grad.cpu(), ref_grads[k]))
        noise.append(rel_l2(g32[k], ref_grads[k]))
    assert max(err) < 0.1 and float(np.median(err)) < 0.05, (max(err), float(np.median(err)), max(noise), float(np.median(noise)))
    # Well-conditioned checks on top of the smoke bar above.
    # (1) The LAST stage (head, slice transformer, class token, layer4) sees no ReLU-flip amplification behind it: tighter bars.
    last = [k for k in ref_grads if k.startswith(("linear.", "slice_fusion.", "cls_token")) or ".layer4." in k]
    assert len(last) > 20
    # (max-norm per parameter held 5e-3 in most runs and measured 4.4e-2 once: a ReLU of layer4 itself flips now and then -- batch statistics
    # over 48 samples -- and moves single entries by about 1/48; the norms below do not hang on single entries)
    P = dict(m.named_parameters())
    worst_last = max(rel_l2(P[k].grad.cpu().double(), ref_grads[k]) for k in last)
    glob_last = (sum(float((P[k].grad.cpu().double() - ref_grads[k]).square().sum()) for k in last) / sum(float(ref_grads[k].square().sum()) for k in last)) ** 0.5
    assert worst_last < 5e-2 and glob_last < 1e-2, (worst_last, glob_last)
    # (2) Directional derivatives: <grad_HIP, v> against a central finite difference of the fp64 oracle LOSS along random parameter
    # directions.  A single ReLU flip changes one gradient entry by O(1/N) but the loss by O(eps^2): the inner product averages
    # over ~2e7 entries, so a wrong BatchNorm momentum term or a mis-scaled shortcut gradient (O(1) on whole tensors) shows at the
    # 1e-2 bar while flips do not.
    from oracle import resnet_oracle as R
    names = [k for k, _ in m.named_parameters()]
    gen = torch.Generator().manual_seed(7)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}

    def loss_at(shift):
        s2 = dict(sd64)
        for k, dv in shift.items():
            s2[k] = sd64[k] + dv
        with torch.no_grad():
            return float(torch.nn.functional.cross_entropy(R.forward_slice_trans(s2, src.double(), mask, model, train=True)["logits"], target))

    dd_err, dd_noise = [], []
    for trial in range(8):
        # direction: per-tensor normalised noise on a random third of the parameter tensors (so every stage is hit over the trials)
        pick = [k for k in names if float(torch.rand((), generator=gen)) < 0.34] or names[:1]
        v = {k: torch.randn(sd64[k].shape, generator=gen, dtype=torch.float64) * float(sd64[k].abs().mean() + 1e-3) for k in pick}
        eps = 1e-7                                       # the loss is piecewise smooth: the difference quotient's error is proportional to the
        fd = (loss_at({k: eps * d for k, d in v.items()}) - loss_at({k: -eps * d for k, d in v.items()})) / (2 * eps)
        dd_hip = sum(float((dict(m.named_parameters())[k].grad.cpu().double() * v[k]).sum()) for k in pick)
        dd_ref = sum(float((ref_grads[k] * v[k]).sum()) for k in pick)
        # ReLU / max-pool kinks inside +-eps (measured on this case: 1.1e-2 at 1e-4, 2.7e-4 at 1e-6, 1.7e-8 at 1e-7)
        assert abs(dd_ref - fd) <= 2e-3 * max(abs(fd), 1e-6) + 1e-9, (trial, dd_ref, fd)          # the oracle's own autograd is consistent (a kink inside +-eps now and then: 3.9e-4 seen)
        # relative to the sum of the per-tensor |<g, v>| rather than to |fd|: the tensors' contributions cancel in some directions (a
        # direction with a near-zero total measured 12.8 % of |fd| at 1 % of that scale)
        den = max(sum(abs(float((ref_grads[k] * v[k]).sum())) for k in pick), 1e-6)
        dd_err.append(abs(dd_hip - fd) / den)
        dd_noise.append(abs(sum(float((g32[k].double() * v[k]).sum()) for k in pick) - fd) / den)   # torch's own fp32 autograd
    # median 2e-2, worst 6e-2: a run whose ReLU pattern differs from the fp64 oracle's in a late layer moves every upstream gradient
    # coherently (train-mode BatchNorm over 48 samples), and the fp32 sums are not run-to-run identical (atomics): medians of 4e-3 ..
    # 1.4e-2 and worst directions of 2.4e-2 .. 3.5e-2 have been measured on unchanged inputs.  A wrong BatchNorm momentum term or a
    # mis-scaled shortcut moves EVERY direction by 10 % or more.
    assert float(np.median(dd_err)) < 2e-2 and max(dd_err) < 6e-2, (dd_err, dd_noise)
    # running statistics as nn.BatchNorm2d updates them in train mode
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_l2(v.cpu(), ref_sd[k]) < 1e-4, k
        if k.endswith("num_batches_tracked") and k in ref_sd:
            assert int(v) == int(ref_sd[k]) == 1


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_mixed_precision_convolution_unit_and_training_steps(prec):
    """train_precision = bf16 / fp16 (the reference's precision='16-mixed' for F.conv2d): the convolution and its input gradient on 16-bit
    MFMA operands, fp32 accumulation, BatchNorm / weight gradient / storage fp32.  Checked where it is well-conditioned -- ONE convolution +
    BatchNorm + ReLU unit against the fp32 unit on the same operands -- and functionally on the whole model (finite gradients for every
    parameter, the first loss within 10 % of the fp32 step's -- 5.5 % measured under bf16 --, AdamW steps lower it).  Whole-step gradients are NOT compared at a tight bar: this
    randomly initialised 34-layer net with batch statistics over 8 images amplifies fp32's own rounding (6e-8) to 2e-3 between two
    identical runs, and a 16-bit rounding of the convolution operands to 0.2 (fp16) / 0.5 (bf16) (tools/check_train_mixed.py --resnet)."""
    import warnings
    from mst import train_resnet as T
    from mst.models import ResNetSliceTrans
    from mst.models.resnet import _BN, _Conv
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[prec]
    g = torch.Generator().manual_seed(9)
    torch.manual_seed(9)                                 # _Conv draws its kaiming weights from the global generator
    for cin, cout, k, stride, pad, hw in ((64, 128, 3, 2, 1, (22, 18)), (128, 128, 3, 1, 1, (12, 10)), (64, 256, 1, 2, 0, (12, 10)), (64, 64, 3, 1, 1, (14, 12))):
        conv, bn = _Conv(cin, cout, k).cuda(), _BN(cout).cuda()
        x = torch.randn(4, hw[0], hw[1], cin, generator=g).cuda()
        Ho, Wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
        dy = torch.randn(4 * Ho * Wo, cout, generator=g).cuda()
        res = {}
        for mp in (None, tdt):
            y, rec = T._conv_bn_fwd(x, conv, bn, k, stride, pad, False, None, True, mp)
            assert (rec["mp"] is None) == (mp is None)
            G = T._Grads()
            dx = T._conv_bn_bwd(G, rec, dy.clone(), True)
            res[mp] = (y, dx, G.by_param[id(conv.weight)])
        tol = {"bf16": 1e-1, "fp16": 4e-2}[prec]                             # measured 3.0e-2 / 2.2e-2 worst (d input: ReLU-mask flips + BatchNorm backward)
        for a, b in zip(res[tdt], res[None]):
            assert rel_l2(a.cpu(), b.cpu()) < tol
    sd = synth.synth_resnet_state_dict(41, 34, 2)
    src = synth.synth_volume((2, 1, 4, 96, 64), 42).cuda()
    tgt = torch.tensor([1, 0]).cuda()
    losses = {}
    for p in ("fp32", prec):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=34, train_precision=p)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
        hist = []
        for _ in range(4):
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(m(src), tgt)
            loss.backward()
            assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.parameters())
            opt.step()
            hist.append(float(loss.detach()))
        losses[p] = hist
    assert abs(losses[prec][0] - losses["fp32"][0]) < 0.1 * losses["fp32"][0], losses
    assert losses[prec][-1] < losses[prec][0], losses
    # bottleneck units (1 x 1 -> 3 x 3 -> 1 x 1, 2048-wide embeddings) take the same three 16-bit products
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=50, train_precision=prec)
    m.load_state_dict(synth.synth_resnet_state_dict(7, 50, 2), strict=True)
    m = m.cuda().train()
    loss = torch.nn.functional.cross_entropy(m(src[:, :, :2]), tgt)
    loss.backward()
    assert torch.isfinite(loss) and all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.parameters())
    with pytest.raises(ValueError):
        ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, train_precision="fp8")


def test_training_loss_goes_down_with_adamw():
    import warnings
    from mst.models import ResNetSliceTrans
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=18)
    m.load_state_dict(synth.synth_resnet_state_dict(3, 18, 2), strict=True)
    from oracle import resnet_oracle as R
    m = m.cuda()
    src = synth.synth_volume((2, 1, 4, 64, 64), 77)
    # the reference Trainer validates BEFORE and BETWEEN training epochs (main_train.py:110-123: num_sanity_val_steps=2,
    # check_val_every_n_epoch=1): an eval forward first, so that the BatchNorm-folded inference weights exist when training starts
    with torch.no_grad():
        before = m.eval()(src).clone()
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    target = torch.tensor([0, 1]).cuda()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(src), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
    # ... and the eval forward AFTER the optimiser steps runs on the updated parameters and running statistics (the folded
    # backbone used to be cached on (sum_in, device) only): it must match the oracle on the model's current state_dict
    with torch.no_grad():
        after = m.eval()(src)
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        ref = R.forward_slice_trans(sd, src, None, model=18)
    assert float((after.cpu() - ref["logits"]).abs().max()) < 1e-3 * max(1.0, float(ref["logits"].abs().max()))
    assert float((after - before).abs().max()) > 1e-3      # the six steps did move the logits: a stale cache would reproduce `before`


def test_full_config3_shape_training_step_properties():
    """BASELINE configs[3] at its full per-GPU size: one ResNetSliceTrans(resnet34) training step on a 1 x 128 x 512 x 512 volume
    (128 images through the backbone with batch statistics over all of them, ~28 GiB of saved activations).  The fp64 oracle does
    not finish at this size, so size-independent properties: a finite loss and a finite, non-zero gradient for every parameter; a
    second identical step from the same weights reproduces the loss to 1e-4 (the atomically accumulated sums reorder); the
    BatchNorm counters moved by exactly one step; one AdamW step on those gradients lowers the loss on the same volume; peak
    memory stays under 40 GiB.  The arithmetic itself is pinned at small sizes above."""
    import warnings
    from mst.models import ResNetSliceTrans
    sd = synth.synth_resnet_state_dict(61, 34, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, model=34)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    src = synth.synth_volume((1, 1, 128, 512, 512), 62)
    target = torch.tensor([1]).cuda()
    losses = []
    for rep in range(2):
        m.load_state_dict(sd, strict=True)
        m.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(src), target)
        loss.backward()
        losses.append(float(loss))
        assert math.isfinite(losses[-1])
        for k, p in m.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, k
        assert int(m.state_dict()["model.bn1.num_batches_tracked"]) == 1
    assert abs(losses[0] - losses[1]) < 1e-4 * max(1.0, abs(losses[0])), losses
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    opt.step()
    with torch.no_grad():
        after = float(torch.nn.functional.cross_entropy(m.train()(src), target))
    assert after < losses[1], (after, losses)
    assert torch.cuda.max_memory_allocated() < 40 * 2 ** 30
