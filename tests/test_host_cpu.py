"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
the host module keeps the reference's surface, and the slice-sharding plumbing works over gloo."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_header_symbol():
    from mst import hip
    header = (ROOT / "include" / "mst_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|size_t|void|const char\*|mst_profiler\*)\s+(mst_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no entry points parsed from include/mst_hip.h"
    lib = hip.load()                                  # raises if the .so is missing: no fallback
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in mst_hip.h but not exported"
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    assert lib.mst_version() == 300 == hip.ABI_VERSION


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os,sys; sys.path.insert(0, %r); os.environ['MST_HIP_LIB']=%r\n"
            "from mst import hip\n"
            "try:\n    hip.load()\nexcept RuntimeError as e:\n    print('RAISED', e)\n") % (
                str(ROOT / "new-vit_amd"), str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    assert "RAISED" in out and "no CPU" in out


def test_product_package_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+\S*oracle|mst_oracle|importlib[^\n]*oracle", re.M)
    for p in list((ROOT / "new-vit_amd").rglob("*.py")) + list((ROOT / "new-vit_amd" / "csrc").glob("*")):
        assert not pat.search(p.read_text()), f"{p} reaches into oracle/"


def test_module_surface_and_state_dict_layouts():
    from mst import synth
    from mst.models import DinoV2ClassifierSlice, ResNet, ResNetSliceTrans
    from mst.models.dino import DinoV3ClassifierSlice
    m = DinoV2ClassifierSlice(in_ch=3, out_ch=2, spatial_dims=2, pretrained=False)   # main_train.py:37
    sd = synth.synth_state_dict("s", 0)
    assert set(m.state_dict()) == set(sd)
    assert all(m.state_dict()[k].shape == v.shape for k, v in sd.items())
    m.load_state_dict(sd, strict=True)
    m.load_state_dict(synth.synth_state_dict("s", 1, chunked=False), strict=True)    # hub block naming
    for attr in ("forward", "get_attention_maps", "get_slice_attention", "get_plane_attention", "get_attention_cls",
                 "training_step", "validation_step", "test_step", "configure_optimizers", "save_best_checkpoint",
                 "load_best_checkpoint", "load_pretrained", "load_weights", "attention_maps", "attention_maps_slice"):
        assert hasattr(m, attr), attr
    opt = m.configure_optimizers()[0]
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 1e-6          # dino.py:41
    for kw in (dict(use_bottleneck=True, use_slice_pos_emb=True), dict(rotary_positional_encoding="RoPE"),
               dict(slice_fusion="average"), dict(slice_fusion="linear"), dict(model_size="b")):
        mm = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, **kw)
        ref = synth.synth_state_dict(kw.get("model_size", "s"), 0, use_bottleneck=kw.get("use_bottleneck", False),
                                     use_slice_pos_emb=kw.get("use_slice_pos_emb", False),
                                     slice_fusion=kw.get("slice_fusion", "transformer"),
                                     rotary=kw.get("rotary_positional_encoding"))
        assert set(mm.state_dict()) == set(ref), kw
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 2, 28, 28))
    with pytest.raises(NotImplementedError):
        DinoV3ClassifierSlice(in_ch=1, out_ch=2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rs = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False)                     # main_predict.py:138-139 isinstance dispatch
        rn = ResNet(in_ch=3, out_ch=2, spatial_dims=2, pretrained=False, model=34)     # reference tests/models/test_resnet.py
    assert isinstance(rs, ResNet)
    ref_keys = set(synth.synth_resnet_state_dict(0, 34, 2))
    assert {k for k in rs.state_dict() if not k.startswith(("auc_roc", "acc."))} == ref_keys
    assert {k for k in rn.state_dict() if not k.startswith(("auc_roc", "acc."))} == set(synth.synth_resnet_state_dict(0, 34, 2, slice_trans=False, fc_out=2))
    rs.load_state_dict(synth.synth_resnet_state_dict(1, 34, 2), strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        rs.eval()(torch.zeros(1, 1, 2, 32, 32))


def test_checkpoint_roundtrip(tmp_path):
    from mst import synth
    from mst.models import DinoV2ClassifierSlice
    m = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, use_bottleneck=True)
    m.load_state_dict(synth.synth_state_dict("s", 3, use_bottleneck=True))
    ck = tmp_path / "epoch=1.ckpt"
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": dict(m.hparams)}, ck)
    DinoV2ClassifierSlice.save_best_checkpoint(tmp_path, ck)
    m2 = DinoV2ClassifierSlice.load_best_checkpoint(tmp_path)
    assert hasattr(m2, "bottleneck")
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())


def test_shard_range_covers_all_slices():
    from mst.parallel import shard_range
    for D in (1, 7, 32, 64, 65, 96):
        for G in (1, 2, 3, 4, 8):
            got = []
            for r in range(G):
                d0, d1, dpad = shard_range(D, G, r)
                assert d1 - d0 <= dpad and dpad * G >= D
                got += list(range(d0, d1))
            assert got == list(range(D))


def _gloo_worker(rank, world, port, q, D=7):
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT / "new-vit_amd"))
    from mst.parallel import SliceSharding
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sh = SliceSharding()
        B, X = 3, 5                                            # D = 7: not divisible by the world size
        full = torch.arange(B * D * X, dtype=torch.float32).reshape(B, D, X)
        d0, d1, dpad = sh.shard_range(D)
        local = torch.zeros(B, dpad, X)
        local[:, : d1 - d0] = full[:, d0:d1]
        out = sh.all_gather_slices(local, D)
        tab = torch.tensor([1.0 + rank, 5.0 - rank, 0.0, float(rank == world - 1)])     # the fp8 scale table: max over the ranks
        sh.all_reduce_max(tab)
        ok_max = bool(torch.equal(tab, torch.tensor([float(world), 5.0, 0.0, 1.0])))
        q.put((rank, bool(torch.equal(out, full)) and ok_max))
    finally:
        dist.destroy_process_group()


def test_slice_sharding_all_gather_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


@pytest.mark.parametrize("D", [7, 3])
def test_slice_sharding_all_gather_gloo_world4_uneven_and_empty_shards(D):
    """Four ranks: D = 7 leaves the last rank one slice of its two-slice shard, D = 3 leaves it none (empty shard, all padding)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + D
    procs = [ctx.Process(target=_gloo_worker, args=(r, 4, port, q, D)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(4)]


def test_oracle_fp8_linear_properties():
    """oracle fp8_linear (BASELINE configs[4] arithmetic; no reference code exists for it): exact when both operands are
    representable in e4m3 at the absmax scale, bounded relative error otherwise, zero input handled."""
    import torch
    from oracle import mst_oracle as O
    g = torch.Generator().manual_seed(0)
    # integers up to 16 with max 448 -> scale 1: exactly representable
    x = torch.randint(-8, 9, (5, 64), generator=g).float()
    x[0, 0] = 448.0
    w = torch.randint(-4, 5, (7, 64), generator=g).float()
    w[0, 0] = 448.0
    b = torch.randn(7, generator=g)
    assert torch.allclose(O.fp8_linear(x, w, b), torch.nn.functional.linear(x, w, b), rtol=0, atol=1e-4)
    # random operands: e4m3 keeps 3 mantissa bits -> per-element relative error <= 2^-4, result error a few % of the scale
    x, w = torch.randn(64, 384, generator=g), torch.randn(128, 384, generator=g) / 384 ** 0.5
    y, r = O.fp8_linear(x, w, None), torch.nn.functional.linear(x, w)
    assert 1e-3 < float((y - r).norm() / r.norm()) < 6e-2
    q, s = O.fp8_e4m3(x)
    assert float(q.abs().max()) == 448.0 and abs(s * 448.0 - float(x.abs().max())) < 1e-6
    assert float((q * s - x).abs().max()) <= float(x.abs().max()) / 16 + 1e-6
    z, _ = O.fp8_e4m3(torch.zeros(4, 8))
    assert float(z.abs().max()) == 0.0


def test_fp8_weight_quantisation_host_side():
    """hip.quantize_weight_fp8 (host-side weight preparation of compute_dtype='fp8'): bytes are torch's e4m3fn encoding,
    scale = max|W|/448, the largest weight maps to +-448 exactly and dequantisation stays within one e4m3 step."""
    import torch
    from mst import hip
    g = torch.Generator().manual_seed(5)
    w = torch.randn(96, 384, generator=g) * 0.03
    q8, scale = hip.quantize_weight_fp8(w)
    assert q8.dtype == torch.uint8 and q8.shape == w.shape and q8.is_contiguous()
    assert abs(scale - float(w.abs().max()) / 448.0) < 1e-9
    deq = q8.view(torch.float8_e4m3fn).float() * scale
    assert float(deq.abs().max()) == pytest.approx(float(w.abs().max()), rel=1e-6)
    # normal range: relative step 2^-4 (half-step error after rounding); subnormals: absolute step 2^-9 * scale
    err = (deq - w).abs()
    assert bool((err <= w.abs() / 16 + 2.0 ** -10 * scale + 1e-12).all())
    z8, zs = hip.quantize_weight_fp8(torch.zeros(8, 16))
    assert int(z8.max()) == 0 and zs == 1.0
    assert "fp8" in hip.DT_NAMES and hip.DT_NAMES["fp8"] == hip.BF16 and "fp8" in hip.FP8_NAMES


def test_oracle_fp8_calibrated_table_reproduces_the_dynamic_run():
    """The oracle's two fp8 modes agree when the calibrated table holds exactly the per-call maxima: pins the order of the
    table's columns (inputs of qkv, proj, fc1, fc2) and its [depth][4] indexing."""
    import torch
    from mst import synth
    from oracle import mst_oracle as O
    sd = synth.synth_state_dict("s", 1)
    x = synth.synth_volume((1, 1, 2, 28, 42), 3).reshape(2, 28, 42)
    seen = []
    real = O.fp8_e4m3

    def spy(t, amax=None):
        seen.append(float(t.detach().abs().max()))
        return real(t, amax)

    O.fp8_e4m3 = spy
    try:
        with torch.no_grad():
            dyn, _ = O.vit_encode(sd, x, "s", linear="fp8")
    finally:
        O.fp8_e4m3 = real
    assert len(seen) == 12 * 4 * 2                          # activation then weight, four linear layers, twelve blocks
    table = torch.tensor(seen[0::2]).reshape(12, 4).tolist()
    with torch.no_grad():
        sta, _ = O.vit_encode(sd, x, "s", linear="fp8", act_amax=table)
        half, _ = O.vit_encode(sd, x, "s", linear="fp8", act_amax=(torch.tensor(table) * 0.5).tolist())
    assert torch.equal(sta, dyn)
    assert not torch.equal(half, dyn)                       # a too-tight table saturates the largest values


def test_bench_flop_model_matches_the_oracle_and_survey():
    """bench.py owns its FLOP model (SURVEY.md 8d formula); the oracle's restatement of it must agree."""
    sys.path.insert(0, str(ROOT))
    import bench
    from oracle import mst_oracle as O
    for D, H, W in ((16, 224, 224), (64, 518, 518), (96, 518, 518), (4, 504, 280)):
        assert abs(bench.flops_per_volume(D, H, W) - O.flops_per_volume(D, H, W)) < 1e-6 * O.flops_per_volume(D, H, W)
    assert abs(bench.flops_per_slice(518, 518) - 93.39e9) < 0.01e9
    assert abs(bench.flops_per_volume(64, 518, 518) - 5.977e12) < 0.001e12


def test_bench_gpus_n_without_devices_fails_loudly():
    """`python bench.py --gpus 2` must start ranks itself or exit non-zero -- never report n_gpus 1.  No GPU here."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MST_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 requested but only" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE 4" in r.stderr


def test_block_stream_packer_and_layout_helpers_cpu():
    """Host logic of the single-role block kernel (include/mst_hip.h, mst_block_fused_s): the weight stream holds every weight once,
    in consumption order and MFMA fragment order; the blocked / image layout helpers are inverse permutations."""
    import torch
    from mst import hip
    E, H = 384, 1536
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen)
    wp, bp, w1, b1, w2, b2, lw, lb, ls1, ls2 = r(E, E), r(E), r(H, E), r(H), r(E, H), r(E), r(E), r(E), r(E), r(E)
    seq, b1f, pbf, b2f = hip.pack_block_seq(wp, bp, ls1, w1, b1, w2, b2, lw, lb, ls2, torch.float32)
    assert seq.shape == (108, 12288) and b1f.shape == (H,) and pbf.shape == (E,) and b2f.shape == (E,)
    s = seq.view(108, 24, 64, 8)
    k8 = lambda p, h, e: 16 * p + 8 * (e >> 2) + 4 * h + (e & 3)
    # out-projection chunk 3, fragment (t 5, p 1), lane (m 7, h 1), element 2
    assert float(s[3, 11, 39, 2]) == float(ls1[167] * wp[167, 96 + 16 + 8 + 2])
    # W1 chunk 0 (element 12) and chunk 2 (element 13 + 2*(2-1)): fragment = k-step 2t+p
    assert float(s[12, 4, 3, 5]) == float(lw[64 + k8(0, 0, 5)] * w1[3, 64 + k8(0, 0, 5)])
    assert float(s[15, 4, 3, 5]) == float(lw[64 + k8(0, 0, 5)] * w1[64 + 3, 64 + k8(0, 0, 5)])
    # W2 chunk 1 (element 14 + 2): fragment (t 2, p 1), lane (m 3, h 1), element 6
    assert float(s[16, 5, 35, 6]) == float(ls2[67] * w2[67, 32 + k8(1, 1, 6)])
    assert float(s[107, 5, 35, 6]) == float(ls2[67] * w2[67, 47 * 32 + k8(1, 1, 6)])
    assert torch.allclose(b1f, b1 + w1 @ lb, atol=1e-5) and torch.equal(pbf, bp * ls1) and torch.equal(b2f, b2 * ls2)
    # every weight exactly once
    assert abs(float(seq.abs().sum()) - float((wp * ls1[:, None]).abs().sum() + (w1 * lw[None]).abs().sum() + (w2 * ls2[:, None]).abs().sum())) < 1.0
    a = r(96, E)
    assert torch.equal(hip.from_blocked16(hip.to_blocked16(a)), a) and torch.equal(hip.from_image32(hip.to_image32(a)), a)
    blk = hip.to_blocked16(a).view(3, 24, 64, 8)
    assert float(blk[1, 5, 7 + 32, 3]) == float(a[32 + 7, 16 * 5 + 8 + 3])          # row 7 of group 1, column 16*5 + 8*1 + 3
    img = hip.to_image32(a).view(3, 48, 64, 4)
    assert float(img[2, 9, 11 + 32, 1]) == float(a[64 + 11, 8 * 9 + 4 + 1])


def test_convolution_gradient_operand_layouts_cpu():
    """Host-side layouts of the implicit-GEMM gradients, checked with torch on the CPU (the kernels themselves: tests/test_resnet_gpu.py):
    mst_conv_dgrad computes a stride-1 convolution of the stride-dilated gradient, padded by k - 1 - pad, with the weight
    hip.conv_dgrad_weight builds ([Cin, (ky', kx', co)], both kernel axes flipped) -- that must be autograd's input gradient."""
    import torch.nn.functional as F
    from mst import hip
    g = torch.Generator().manual_seed(0)
    for cin, cout, k, stride, pad, hw in ((4, 6, 3, 1, 1, (7, 6)), (4, 6, 3, 2, 1, (9, 8)), (3, 5, 1, 2, 0, (7, 5)), (2, 3, 3, 2, 1, (6, 6))):
        x = torch.randn(2, cin, *hw, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64)
        y = F.conv2d(x, w, stride=stride, padding=pad)
        dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(dz)
        Ho, Wo = y.shape[2:]
        dil = torch.zeros(2, cout, (Ho - 1) * stride + 1, (Wo - 1) * stride + 1, dtype=torch.float64)
        dil[:, :, ::stride, ::stride] = dz                                        # the gradient rows on every stride-th position
        pt = k - 1 - pad
        # pad so that the stride-1 convolution has exactly H x W outputs (the kernel bounds-checks instead of padding)
        dil = F.pad(dil, (pt, hw[1] + k - 1 - pt - dil.shape[3], pt, hw[0] + k - 1 - pt - dil.shape[2]))
        wt = hip.conv_dgrad_weight(w, torch.float64)                              # [Cin, (ky', kx', co)]
        dx = F.conv2d(dil, wt.view(cin, k, k, cout).permute(0, 3, 1, 2))
        assert dx.shape == x.shape and torch.allclose(dx, x.grad, atol=1e-12)


def test_build_specific_precision_keywords_are_validated_cpu():
    import warnings
    from mst.models import DinoV2ClassifierSlice, ResNetSliceTrans
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, compute_dtype="bf16", train_precision="fp16")
        assert (m.compute_dtype_name, m.train_precision) == ("bf16", "fp16")
        assert ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False).compute_dtype_name == "fp32"
        for kw in (dict(compute_dtype="fp8"), dict(train_precision="int8")):
            with pytest.raises(ValueError):
                ResNetSliceTrans(in_ch=1, out_ch=2, pretrained=False, **kw)
    d = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, train_precision="bf16")
    assert d.train_precision == "bf16" and DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False).train_precision == "fp32"
    with pytest.raises(ValueError):
        DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, train_precision="fp8")
