"""ResNet / ResNetSliceTrans inference on the HIP path (reference mst/models/resnet.py:27-243).
The across-slice half is checked against a fixture the reference's own TransformerEncoderLayer(512, nhead 16) produced
(tests/golden/resnet_fusion.npz); the torchvision backbone is not in the reference tree, so its parity is against the restated
architecture of oracle/resnet_oracle.py (UNPINNED), fp32, relative 1e-4."""
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
    with pytest.raises(NotImplementedError), torch.no_grad():
        m(x, save_attn=True)                                                              # Grad-CAM++ needs the backward
    m.train()
    with pytest.raises(NotImplementedError):
        m(x)                                                                              # training step of the backbone: not built
    with pytest.raises(NotImplementedError):
        ResNet(in_ch=1, out_ch=2, spatial_dims=3)
    st = _model(7)
    with pytest.raises(RuntimeError, match="channels"), torch.no_grad():
        st(torch.zeros(1, 2, 3, 32, 32))                                                  # only gray volumes fit the 3-channel stem
