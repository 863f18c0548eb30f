"""Per-kernel parity on a real MI355X (``-m gpu``): every call goes through the C ABI (mst.hip).

Floating-point path: the checker is a plain fp64/fp32 PyTorch-CPU reference of the same op on the
SAME (already rounded) operands, so tolerances reflect accumulation order and the output rounding
only:  f32 out 2e-5 relative-to-scale;  fp16 out 1e-3;  bf16 out 8e-3 (one bf16 ulp = 2^-8).
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from mst import synth

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
OUT_TOL = {torch.float32: 2e-5, torch.float16: 1.5e-3, torch.bfloat16: 8e-3}


@pytest.fixture(scope="module")
def hip():
    from mst import hip as h
    h.load()
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return h


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(synth.hash_normal(tuple(shape), seed, 77)) * scale


def scaled_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1, 384), (7, 384), (1030, 384), (5, 768), (9, 96), (3, 48), (4, 1024)])
@pytest.mark.parametrize("odt", [torch.float32, torch.float16, torch.bfloat16])
def test_layernorm(hip, rows, cols, odt):
    x = rnd((rows, cols), 1, 3.0) + 0.5
    g, b = rnd((cols,), 2) * 0.2 + 1, rnd((cols,), 3) * 0.2
    ref = torch.nn.functional.layer_norm(x.double(), (cols,), g.double(), b.double(), 1e-6)
    got = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, odt)
    assert got.dtype == odt
    assert scaled_err(got, ref) < OUT_TOL[odt]


def test_layernorm_strided_cls_rows(hip):
    """final norm reads only row 0 of every sequence (row stride N*E)."""
    n, N, E = 5, 9, 384
    x = rnd((n, N, E), 4).cuda()
    g, b = (rnd((E,), 5) + 1).cuda(), rnd((E,), 6).cuda()
    out = torch.empty(n, E, device="cuda")
    rc = hip.load().mst_layernorm(x.data_ptr(), N * E, g.data_ptr(), b.data_ptr(), out.data_ptr(), hip.F32, E, n, E,
                                  1e-6, hip.stream_of(x))
    assert rc == 0
    ref = torch.nn.functional.layer_norm(x[:, 0].double().cpu(), (E,), g.double().cpu(), b.double().cpu(), 1e-6)
    assert scaled_err(out, ref) < 2e-5


# ---------------------------------------------------------------------------------------------------
def _gemm_ref(a, w, bias, epi, gamma=None, resid=None, col_scale=1.0, scale_cols=0):
    y = a.double() @ w.double().t() + bias.double()
    if scale_cols:
        y[:, :scale_cols] *= col_scale
    if epi == 1:
        y = 0.5 * y * (1 + torch.erf(y / math.sqrt(2)))
    elif epi == 2:
        y = torch.relu(y)
    elif epi == 3:
        y = resid.double() + (gamma.double() if gamma is not None else 1.0) * y
    return y


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 384, 384), (1000, 1152, 384), (300, 384, 1536), (4112, 1536, 384)])
def test_gemm16_epilogues(hip, dt, M, N, K):
    tdt = DT[dt]
    a = (rnd((M, K), 10)).to(tdt)
    w = (rnd((N, K), 11) / math.sqrt(K)).to(tdt)
    bias = rnd((N,), 12) * 0.1
    ac, wc, bc = a.cuda(), w.cuda(), bias.cuda()
    for epi in (0, 1, 2):
        for odt in (tdt, torch.float32):
            got = hip.gemm(ac, wc, bc, epilogue=epi, out_dtype=odt)
            ref = _gemm_ref(a.float(), w.float(), bias, epi)
            assert scaled_err(got, ref) < OUT_TOL[odt], (epi, odt)
    # q-scale columns (QKV projection) and the residual epilogue with / without LayerScale
    got = hip.gemm(ac, wc, bc, epilogue=0, out_dtype=torch.float32, col_scale=0.125, scale_cols=128)
    assert scaled_err(got, _gemm_ref(a.float(), w.float(), bias, 0, col_scale=0.125, scale_cols=128)) < 2e-5
    for gamma in (None, rnd((N,), 13) * 0.3 + 1):
        resid = rnd((M, N), 14)
        out = resid.cuda().clone()
        hip.gemm(ac, wc, bc, epilogue=3, out=out, gamma=None if gamma is None else gamma.cuda())
        assert scaled_err(out, _gemm_ref(a.float(), w.float(), bias, 3, gamma, resid)) < 2e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(16500, 1152, 384), (49300, 384, 1536), (12400, 1536, 384), (65536 + 77, 384, 384)])
def test_gemm16_big_tile_path(hip, dt, M, N, K):
    """Shapes large enough for the persistent 256x384 kernel (>= 192 tiles), ragged last M tile, all epilogues."""
    tdt = DT[dt]
    a = rnd((M, K), 15).to(tdt)
    w = (rnd((N, K), 16) / math.sqrt(K)).to(tdt)
    bias = rnd((N,), 17) * 0.1
    ac, wc, bc = a.cuda(), w.cuda(), bias.cuda()
    ad, wd = a.double(), w.double()
    base = ad @ wd.t() + bias.double()
    got = hip.gemm(ac, wc, bc, epilogue=0, out_dtype=torch.float32, col_scale=0.125, scale_cols=N // 3)
    ref = base.clone()
    ref[:, : N // 3] *= 0.125
    assert scaled_err(got, ref) < 2e-5
    got = hip.gemm(ac, wc, bc, epilogue=1, out_dtype=tdt)
    assert scaled_err(got, 0.5 * base * (1 + torch.erf(base / math.sqrt(2)))) < OUT_TOL[tdt]
    got = hip.gemm(ac, wc, bc, epilogue=2, out_dtype=tdt)
    assert scaled_err(got, torch.relu(base)) < OUT_TOL[tdt]
    gamma = rnd((N,), 18) * 0.3 + 1
    resid = rnd((M, N), 19)
    out = resid.cuda().clone()
    hip.gemm(ac, wc, bc, epilogue=3, out=out, gamma=gamma.cuda())
    assert scaled_err(out, resid.double() + gamma.double() * base) < 2e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm16_big_exact_integers(hip, dt):
    tdt = DT[dt]
    M, N, K = 256 * 200, 384, 64
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[torch.arange(M), (torch.arange(M) * 7 + 3) % K] += 2.0
    w = (torch.arange(N)[:, None] % 13 - 6) * 1.0 + (torch.arange(K)[None, :] % 7) * 2.0
    got = hip.gemm(a.to(tdt).cuda(), w.to(tdt).cuda(), None, out_dtype=torch.float32)
    assert torch.equal(got.cpu(), a @ w.t())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm16_exact_integers_asymmetric(hip, dt):
    """A = [I | 0] against an asymmetric integer W: catches any row/col or k-order swap exactly."""
    tdt = DT[dt]
    M, N, K = 256, 256, 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    w = (torch.arange(N)[:, None] % 13 - 6) * 1.0 + (torch.arange(K)[None, :] % 7) * 2.0   # |w| <= 18: exact in bf16
    got = hip.gemm(a.to(tdt).cuda(), w.to(tdt).cuda(), None, out_dtype=torch.float32)
    assert torch.equal(got.cpu(), a @ w.t())


@pytest.mark.parametrize("M,N,K", [(1, 2, 384), (4, 2, 96), (65, 1152, 384), (130, 96, 384), (260, 384, 384), (33, 48, 48), (1, 2, 12288),
                                   (1024, 100, 64), (1025, 100, 64), (2080, 1152, 384)])
def test_gemm32(hip, M, N, K):
    """Exact-fp32 GEMM vs fp64: M <= 1024 runs the one-wave-per-tile kernel (k_gemm32s.hip), larger M the LDS-tiled one; both are
    fp32 FMA chains in different k orders, so the bound is fp32 accumulation noise (2e-5 of the output scale at K = 12288)."""
    a, w, bias = rnd((M, K), 20), rnd((N, K), 21) / math.sqrt(K), rnd((N,), 22) * 0.1
    for epi in (0, 1, 2):
        got = hip.gemm(a.cuda(), w.cuda(), bias.cuda(), epilogue=epi)
        assert scaled_err(got, _gemm_ref(a, w, bias, epi)) < 2e-5, epi
    resid = rnd((M, N), 23)
    out = resid.cuda().clone()
    hip.gemm(a.cuda(), w.cuda(), bias.cuda(), epilogue=3, out=out)
    assert scaled_err(out, _gemm_ref(a, w, bias, 3, None, resid)) < 2e-5


def test_gemm_falls_back_for_unaligned_operands(hip):
    """the weights-in-registers kernel (M >= 8192, K = 384, N % 384 == 0) moves 16-byte pieces on A, W and C; a row
    pitch that is not a multiple of 8 elements or a pointer off a 16-byte boundary must take the tiled kernels, same results."""
    M, K, N = 8448, 384, 384
    a = rnd((M, K), 31).to(torch.bfloat16).cuda()
    w = (rnd((N, K), 32) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = rnd((N,), 33).cuda()
    ref = hip.gemm(a, w, bias)
    # A shifted by 8 bytes (4 elements) inside a larger allocation: same values, misaligned base pointer, aligned pitch
    big = torch.zeros(M * K + 8, dtype=torch.bfloat16, device="cuda")
    a2 = big[4:4 + M * K].view(M, K)
    a2.copy_(a)
    assert a2.data_ptr() % 16 == 8
    out = hip.gemm(a2, w, bias)
    ref64 = _gemm_ref(a.cpu(), w.cpu(), bias.cpu(), 0, None, None)
    assert scaled_err(out, ref64) < 1e-2                 # the tiled kernels (different summation order than the wreg kernel)
    assert scaled_err(ref, ref64) < 1e-2


def test_colsum_more_row_blocks_than_grid_y(hip):
    """16.8 M rows (the stem BatchNorm sums of a 2 x 128 x 512^2 ResNet step) were 65,536 blocks of 256 rows -- one more
    than gridDim.y takes; the row blocks sit on gridDim.x (and their size now grows with the row count)."""
    rows, cols = 256 * 65536 + 3, 2
    a = torch.ones(rows, cols, device="cuda")
    a[:, 1] = 0.5
    out = torch.zeros(cols, device="cuda")
    hip.colsum(a, out)
    assert torch.allclose(out.cpu(), torch.tensor([float(rows), rows * 0.5]), rtol=1e-6)


@pytest.mark.parametrize("rows,cols", [(4112, 384), (1, 5), (17, 64), (1000, 130), (70, 4100), (300001, 64), (33, 2)])
@pytest.mark.parametrize("prod", [False, True])
def test_colsum_vector_and_scalar_column_blocks(hip, rows, cols, prod):
    """mst_colsum: 64-column blocks, float4 across the columns when cols % 4 == 0 (scalar otherwise), partial sums reduced in LDS; with and
    without the second operand, accumulating onto what `out` holds."""
    g = torch.Generator().manual_seed(rows + cols)
    a = torch.randn(rows, cols, generator=g)
    b = torch.randn(rows, cols, generator=g) if prod else None
    out0 = torch.randn(cols, generator=g)
    want = out0.double() + (a.double() * (b.double() if prod else 1.0)).sum(0)
    out = out0.cuda()
    hip.colsum(a.cuda(), out, b.cuda() if prod else None)
    assert float((out.cpu() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max())) * max(1.0, rows ** 0.5 / 30)


def test_gemm_rejects_bad_shapes(hip):
    a = torch.zeros(8, 100, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        hip.gemm(a, w, None)
    with pytest.raises(RuntimeError, match="HIP device"):
        hip.gemm(torch.zeros(8, 64), torch.zeros(128, 64), None)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,with_ls,with_xn", [(16, False, True), (300, True, True), (128 * 300 + 5, False, False), (41000, True, True)])
def test_mlp_fused(hip, dt, M, with_ls, with_xn):
    """LN2 -> fc1 -> GELU -> fc2 -> residual (+ next normalise) in one kernel vs an fp64 reference of the
    reference's block arithmetic (block.py:93-94; mlp.py:34-40).  Operand rounding is the kernel's own (weights,
    normalised rows and hidden activations in `dt`), so the tolerance is that of a 16-bit GEMM chain."""
    tdt = DT[dt]
    E, Hd = 384, 1536
    x = rnd((M, E), 50, 1.5) + 0.3
    w1, b1 = rnd((Hd, E), 51) / math.sqrt(E), rnd((Hd,), 52) * 0.1
    w2, b2 = rnd((E, Hd), 53) / math.sqrt(Hd), rnd((E,), 54) * 0.1
    g, be = rnd((E,), 55) * 0.2 + 1, rnd((E,), 56) * 0.2
    ls = (rnd((E,), 57) * 0.3 + 1) if with_ls else None
    wpack, b1p, b2p = hip.pack_mlp(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(), be.cuda(),
                                   None if ls is None else ls.cuda(), tdt)
    xc = x.cuda().clone()
    xn = torch.empty(M, E, dtype=tdt, device="cuda") if with_xn else None
    hip.mlp_fused(xc, wpack, b1p, b2p, xn, tdt)
    xd = x.double()
    h = torch.nn.functional.layer_norm(xd, (E,), g.double(), be.double(), 1e-6)
    h = h @ w1.double().t() + b1.double()
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2)))
    y = h @ w2.double().t() + b2.double()
    ref = xd + (ls.double() if ls is not None else 1.0) * y
    tol = {"bf16": 1.5e-2, "fp16": 2.5e-3}[dt]
    assert float((xc.double().cpu() - ref).abs().max() / y.abs().max()) < tol
    if with_xn:
        refn = torch.nn.functional.layer_norm(ref, (E,))
        assert float((xn.double().cpu() - refn).abs().max()) < tol * 4


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,with_ls,with_xn,alias", [(16, False, True, False), (300, True, True, True), (128 * 300 + 5, False, False, False),
                                                     (41000, True, True, True), (128 * 257, False, True, True)])
def test_block_fused(hip, dt, M, with_ls, with_xn, alias):
    """Out-projection + residual + LN2 -> fc1 -> GELU -> fc2 -> residual (+ next normalise) in ONE kernel (mst_block_fused)
    vs an fp64 reference of the reference's block arithmetic (attention.py:67-68; block.py:89-94,112-113; mlp.py:34-40), and
    against the two-launch path it replaces (mst_gemm residual epilogue + mst_mlp_fused).  `alias`: xn_out is the attention
    output buffer itself, as the encoder calls it."""
    tdt = DT[dt]
    E, Hd = 384, 1536
    x = rnd((M, E), 60, 1.5) + 0.3
    att = (rnd((M, E), 61, 1.0)).to(tdt)
    wp, bp = rnd((E, E), 62) / math.sqrt(E), rnd((E,), 63) * 0.1
    w1, b1 = rnd((Hd, E), 51) / math.sqrt(E), rnd((Hd,), 52) * 0.1
    w2, b2 = rnd((E, Hd), 53) / math.sqrt(Hd), rnd((E,), 54) * 0.1
    g, be = rnd((E,), 55) * 0.2 + 1, rnd((E,), 56) * 0.2
    ls1 = (rnd((E,), 64) * 0.3 + 1) if with_ls else None
    ls2 = (rnd((E,), 57) * 0.3 + 1) if with_ls else None
    cu = lambda v: None if v is None else v.cuda()
    wpack, b1p, b2p = hip.pack_mlp(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(), be.cuda(), cu(ls2), tdt)
    ppack, pbf = hip.pack_proj(wp.cuda(), bp.cuda(), cu(ls1), tdt)
    xc = x.cuda().clone()
    attc = att.cuda().clone()
    xn = (attc if alias else torch.empty(M, E, dtype=tdt, device="cuda")) if with_xn else None
    hip.block_fused(xc, attc, ppack, pbf, wpack, b1p, b2p, xn)
    # fp64 reference on the same 16-bit attention output
    xd = x.double()
    proj = att.double() @ wp.double().t() + bp.double()
    xmid = xd + (ls1.double() if ls1 is not None else 1.0) * proj
    h = torch.nn.functional.layer_norm(xmid, (E,), g.double(), be.double(), 1e-6)
    h = h @ w1.double().t() + b1.double()
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2)))
    y = h @ w2.double().t() + b2.double()
    ref = xmid + (ls2.double() if ls2 is not None else 1.0) * y
    tol = {"bf16": 1.5e-2, "fp16": 2.5e-3}[dt]
    scale = max(float(y.abs().max()), float(proj.abs().max()))
    assert float((xc.double().cpu() - ref).abs().max() / scale) < tol
    if with_xn:
        refn = torch.nn.functional.layer_norm(ref, (E,))
        assert float((xn.double().cpu() - refn).abs().max()) < tol * 4
    # the two launches it replaces, on the same operands: same arithmetic up to the rounding of ls1-scaled weights
    x2 = x.cuda().clone()
    hip.gemm(att.cuda(), wp.cuda().to(tdt), bp.cuda(), epilogue=hip.EPI_RESIDUAL, out=x2, gamma=cu(ls1))
    xn2 = torch.empty(M, E, dtype=tdt, device="cuda") if with_xn else None
    hip.mlp_fused(x2, wpack, b1p, b2p, xn2, tdt)
    assert float((xc - x2).abs().max() / scale) < tol
    if not with_ls:
        assert float((xc - x2).abs().max() / scale) < 2e-3     # identical weight rounding: only the summation order differs


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,with_ls,with_xn,alias", [(16, False, True, False), (300, True, True, True), (128 * 300 + 5, False, False, False),
                                                     (41000, True, True, True), (128 * 257, False, True, True)])
def test_block_fused_single_role(hip, dt, M, with_ls, with_xn, alias):
    """mst_block_fused_s (round 3: one wave per SIMD owns its rows end to end; weights as one stream in consumption order) vs
    the same fp64 reference of the reference's block arithmetic as test_block_fused (attention.py:67-68; block.py:89-94,112-113;
    mlp.py:34-40), and against the producer/consumer kernel on the same operands."""
    tdt = DT[dt]
    E, Hd = 384, 1536
    x = rnd((M, E), 60, 1.5) + 0.3
    att = (rnd((M, E), 61, 1.0)).to(tdt)
    wp, bp = rnd((E, E), 62) / math.sqrt(E), rnd((E,), 63) * 0.1
    w1, b1 = rnd((Hd, E), 51) / math.sqrt(E), rnd((Hd,), 52) * 0.1
    w2, b2 = rnd((E, Hd), 53) / math.sqrt(Hd), rnd((E,), 54) * 0.1
    g, be = rnd((E,), 55) * 0.2 + 1, rnd((E,), 56) * 0.2
    ls1 = (rnd((E,), 64) * 0.3 + 1) if with_ls else None
    ls2 = (rnd((E,), 57) * 0.3 + 1) if with_ls else None
    cu = lambda v: None if v is None else v.cuda()
    seq, b1f, pbf, b2f = hip.pack_block_seq(wp.cuda(), bp.cuda(), cu(ls1), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(),
                                            be.cuda(), cu(ls2), tdt)
    xc = x.cuda().clone()
    attc = att.cuda().clone()
    xn = (attc if alias else torch.empty(M, E, dtype=tdt, device="cuda")) if with_xn else None
    hip.block_fused_s(xc, attc, seq, b1f, pbf, b2f, xn)
    xd = x.double()
    proj = att.double() @ wp.double().t() + bp.double()
    xmid = xd + (ls1.double() if ls1 is not None else 1.0) * proj
    h = torch.nn.functional.layer_norm(xmid, (E,), g.double(), be.double(), 1e-6)
    h = h @ w1.double().t() + b1.double()
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2)))
    y = h @ w2.double().t() + b2.double()
    ref = xmid + (ls2.double() if ls2 is not None else 1.0) * y
    tol = {"bf16": 1.5e-2, "fp16": 2.5e-3}[dt]
    scale = max(float(y.abs().max()), float(proj.abs().max()))
    assert float((xc.double().cpu() - ref).abs().max() / scale) < tol
    if with_xn:
        refn = torch.nn.functional.layer_norm(ref, (E,))
        assert float((xn.double().cpu() - refn).abs().max()) < tol * 4
    # the producer/consumer kernel on the same operands: same operand rounding, different summation order only
    wpack, b1p, b2p = hip.pack_mlp(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(), be.cuda(), cu(ls2), tdt)
    ppack, pbf2 = hip.pack_proj(wp.cuda(), bp.cuda(), cu(ls1), tdt)
    x2 = x.cuda().clone()
    att2 = att.cuda().clone()
    hip.block_fused(x2, att2, ppack, pbf2, wpack, b1p, b2p, None)
    assert float((xc - x2).abs().max() / scale) < 2e-3
    # the layouts a block keeps between its neighbours inside one encoder (fp32 image / 16-bit blocked): bit-identical results
    Mp = (M + 31) // 32 * 32
    pad = lambda t: torch.cat([t, torch.zeros(Mp - M, E, dtype=t.dtype, device=t.device)])
    for layout in (hip.LAYOUT_X_IN_IMAGE, hip.LAYOUT_X_OUT_IMAGE, hip.LAYOUT_ACT_BLOCKED, 5, 6, 7):
        xi = pad(x.cuda())
        if layout & hip.LAYOUT_X_IN_IMAGE:
            xi = hip.to_image32(xi)
        ai = pad(att.cuda())
        if layout & hip.LAYOUT_ACT_BLOCKED:
            ai = hip.to_blocked16(ai)
        # (in place also when the two x flags differ: a 32-row group occupies the same 48 KiB in both layouts and one wave owns it)
        xo = (ai if alias else torch.empty(Mp, E, dtype=tdt, device="cuda")) if with_xn else None
        hip.block_fused_s(xi, ai, seq, b1f, pbf, b2f, xo, layout=layout)
        got_x = (hip.from_image32(xi) if layout & hip.LAYOUT_X_OUT_IMAGE else xi)[:M]
        assert torch.equal(got_x, xc), layout
        if with_xn:
            got_n = (hip.from_blocked16(xo) if layout & hip.LAYOUT_ACT_BLOCKED else xo)[:M]
            assert torch.equal(got_n, xn[:M]), layout


# ---------------------------------------------------------------------------------------------------
def _attn_ref(qkv, n, N, heads, hd):
    q, k, v = qkv.double().reshape(n, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-2, -1)).softmax(-1)
    return (p @ v).transpose(1, 2).reshape(n * N, heads * hd), p


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,N,heads", [(2, 64, 2), (3, 257, 6), (2, 300, 2), (1, 1370, 6), (2, 17, 1)])
def test_attention(hip, dt, n, N, heads):
    tdt = DT.get(dt, torch.float32)
    qkv = rnd((n * N, 3 * heads * 64), 30, 1.0)
    qkv[:, : heads * 64] *= 0.35         # q arrives pre-scaled; keep logits O(few)
    qkv = qkv.to(tdt)
    ref, p = _attn_ref(qkv.float(), n, N, heads, 64)
    got = hip.attention(qkv.cuda(), n, N, heads)
    tol = {"bf16": 1.2e-2, "fp16": 2e-3, "fp32": 2e-5}[dt]   # P is rounded to the operand type before P.V
    assert scaled_err(got, ref) < tol
    probs = hip.attention_cls_probs(qkv.cuda(), n, N, heads)
    assert scaled_err(probs, p[:, :, 0]) < 1e-5
    assert torch.allclose(probs.sum(-1).cpu(), torch.ones(n, heads), atol=1e-5)
    if N <= 300:
        full = hip.attention_probs_full(qkv.cuda(), n, N, heads)
        assert scaled_err(full, p) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
def test_attention_online_softmax_rescale_branch(hip, dt):
    """Force the running max to jump at a late KV tile (guide rule 26): spike one key per query block."""
    tdt = DT.get(dt, torch.float32)
    n, N, heads = 1, 400, 1
    qkv = rnd((N, 192), 31, 0.3)
    qkv[:, :64] = qkv[:, :64].abs() * 0.3 + 0.1
    qkv[333, 64:128] = 6.0             # key 333 (tile 5) dominates every query
    qkv[70, 64:128] = 3.0              # an earlier, smaller spike in tile 1
    qkv = qkv.to(tdt)
    ref, _ = _attn_ref(qkv.float(), n, N, heads, 64)
    got = hip.attention(qkv.cuda(), n, N, heads)
    assert scaled_err(got, ref) < {"bf16": 1.2e-2, "fp16": 2e-3, "fp32": 2e-5}[dt]


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["spike_first_tile", "spike_late_tile", "all_very_negative", "late_tile_above_first_by_20", "ragged_26_of_64"])
def test_attention_fast_mode_guard_and_fallback(hip, dt, case):
    """The 16-bit kernel drops the running maximum (and the reference MFMA) after a first tile with small maxima and guards the range
    with the tile's row sum (k_attn16.hip): scores that leave the range of exp2 / of the operand type in the first tile (classic mode
    from the start), in a late tile (that tile is recomputed the classic way), far below zero, or moderately above the first tile
    (fp16: guard trips, bf16: stays fast) must all give the softmax of the reference."""
    tdt = DT[dt]
    n, N, heads = 2, {"ragged_26_of_64": 64 * 3 + 26}.get(case, 400), 2
    qkv = rnd((n * N, 3 * heads * 64), 77, 0.3)
    E = heads * 64
    qkv[:, :E] = qkv[:, :E].abs() * 0.3 + 0.1                     # positive queries: a key's scale steers its score for every query
    if case == "spike_first_tile":
        qkv[5, E:2 * E] = 12.0                                     # ~190 in the log2 domain: beyond exp2's range without a reference
        qkv[N + 40, E:2 * E] = 12.0
    elif case == "spike_late_tile":
        qkv[333, E:2 * E] = 12.0
        qkv[N + 200, E:2 * E] = 12.0
    elif case == "all_very_negative":
        qkv[:, E:2 * E] = -(qkv[:, E:2 * E].abs() + 8.0)           # every score ~ -100 natural: exp2 underflows without a reference
    elif case == "late_tile_above_first_by_20":
        qkv[300, E:2 * E] = 1.6                                    # ~ +25 in the log2 domain
    qkv = qkv.to(tdt)
    ref, _ = _attn_ref(qkv.float(), n, N, heads, 64)
    got = hip.attention(qkv.cuda(), n, N, heads)
    assert torch.isfinite(got).all()
    assert scaled_err(got, ref) < {"bf16": 1.2e-2, "fp16": 2e-3}[dt]


# ---------------------------------------------------------------------------------------------------
def test_pos_embed_interp_matches_reference_fixture(hip):
    g = load_golden("ops")
    pe = (torch.from_numpy(synth.hash_normal((1, 257, 384), 9, 1)) * 0.2)[0]
    for tag, (gh, gw) in (("518", (37, 37)), ("504", (36, 36)), ("518x224", (37, 16))):
        got = hip.pos_embed_interp(pe[1:].contiguous().cuda(), 16, gh, gw, 0.1)
        ref = torch.from_numpy(g[f"pos224_to_{tag}"])[0, 1:]
        assert (got.cpu() - ref).abs().max() < 2e-6, tag


@pytest.mark.parametrize("dt,idt", [("fp32", torch.float32), ("fp16", torch.float32), ("bf16", torch.float32),
                                    ("bf16", torch.bfloat16), ("fp16", torch.float16)])
@pytest.mark.parametrize("n,H,W,R", [(3, 56, 84, 0), (2, 224, 224, 0), (5, 42, 28, 4)])
def test_patch_embed(hip, dt, idt, n, H, W, R):
    from oracle import mst_oracle as O
    tdt = DT.get(dt, torch.float32)
    E = 384
    vol = rnd((n, H, W), 40).to(idt)
    w = rnd((E, 3, 14, 14), 41) / math.sqrt(588)
    bias = rnd((E,), 42) * 0.1
    Np = (H // 14) * (W // 14)
    prefix, pos = rnd((1 + R, E), 43), rnd((Np, E), 44) * 0.2
    wp = torch.zeros(E, 14, 16)
    wp[:, :, :14] = w.sum(1)
    wp = wp.reshape(E, 224).to(tdt)
    got = hip.patch_embed(vol.cuda(), wp.cuda(), bias.cuda(), prefix.cuda(), pos.cuda())
    # reference on the rounded operands: conv as GEMM + pos, prefix rows verbatim
    cols = vol.float().reshape(n, H // 14, 14, W // 14, 14).permute(0, 1, 3, 2, 4).reshape(n, Np, 196).to(tdt).double()
    wk = wp.double().reshape(E, 14, 16)[:, :, :14].reshape(E, 196)
    ref = torch.cat([prefix.double().expand(n, -1, -1), cols @ wk.t() + bias.double() + pos.double()], dim=1)
    assert got.shape == (n, 1 + R + Np, E)
    assert scaled_err(got, ref) < 2e-5
    if dt == "fp32" and idt == torch.float32 and R == 0:
        # and against the oracle's reference-shaped patch embed (3-channel conv): fp32 noise only
        pe = O.patch_embed(vol, w, bias)
        assert scaled_err(got[:, 1:] - pos.cuda(), pe) < 2e-5


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["plain", "rope"])
def test_slice_fusion_matches_reference_layer_fixture(hip, tag):
    """d_model 48 / 12 heads transformer layer captured from the reference classes (ops.npz)."""
    g = load_golden("ops")
    pre = f"tel_{tag}_sd."
    sd = {k[len(pre):]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith(pre)}
    x = torch.from_numpy(g[f"tel_{tag}_in"])            # [2, 9, 48]: row 0 plays the CLS token
    B, L, E = x.shape
    D = L - 1
    fw = hip.FusionWeights()
    fw.emb_in = fw.emb = E
    fw.num_heads, fw.out_ch, fw.fusion_type = 12, 0, hip.FUSION_TRANSFORMER
    P = lambda k: sd[k].contiguous().data_ptr()
    fw.ln1_w, fw.ln1_b = P("layers.0.norm1.weight"), P("layers.0.norm1.bias")
    fw.in_proj_w, fw.in_proj_b = P("layers.0.self_attn.in_proj_weight"), P("layers.0.self_attn.in_proj_bias")
    fw.out_proj_w, fw.out_proj_b = P("layers.0.self_attn.out_proj.weight"), P("layers.0.self_attn.out_proj.bias")
    fw.ln2_w, fw.ln2_b = P("layers.0.norm2.weight"), P("layers.0.norm2.bias")
    fw.lin1_w, fw.lin1_b = P("layers.0.linear1.weight"), P("layers.0.linear1.bias")
    fw.lin2_w, fw.lin2_b = P("layers.0.linear2.weight"), P("layers.0.linear2.bias")
    fw.norm_w, fw.norm_b = P("norm.weight"), P("norm.bias")
    if tag == "rope":
        fw.rope_freqs = P("layers.0.self_attn.rotary_positional_encoding.freqs")
    ws = torch.empty(hip.fusion_workspace_bytes(fw, 1, D), dtype=torch.uint8, device="cuda")
    for masked in (False, True):
        mask_full = torch.from_numpy(g[f"tel_{tag}_mask"])
        for b in range(B):
            # the fixture treats token 0 as an ordinary token; feed it as this volume's "cls_token"
            cls = x[b, 0].contiguous().cuda()
            fw.cls_token = cls.data_ptr()
            emb = x[b, 1:].contiguous().cuda()
            feat = torch.empty(1, E, device="cuda")
            probs = torch.empty(1, 12, L, L, device="cuda")
            m = mask_full[b:b + 1, 1:].to(torch.uint8).contiguous().cuda() if masked else None
            hip.slice_fusion(fw, emb, 1, D, m, feat, None, probs, ws)
            ref_out = g[f"tel_{tag}_out_masked" if masked else f"tel_{tag}_out"][b, 0]
            ref_w = g[f"tel_{tag}_weights_masked" if masked else f"tel_{tag}_weights"][b]
            assert np.abs(feat.cpu().numpy()[0] - ref_out).max() < 2e-5
            assert np.abs(probs.cpu().numpy()[0] - ref_w).max() < 5e-6


def test_attention_readout(hip):
    from oracle import mst_oracle as O
    B, D, heads, Np, R = 2, 5, 6, 16, 0
    N = 1 + R + Np
    cls = torch.rand(B * D, heads, N).softmax(-1)
    sp = torch.rand(B, 12, D + 1, D + 1).softmax(-1)
    plane = torch.empty(B * D, heads, Np, device="cuda")
    maps = torch.empty_like(plane)
    sa = torch.empty(B * D, device="cuda")
    hip.attention_readout(cls.cuda(), sp.cuda(), B, D, heads, N, R, 12, plane, sa, maps)
    assert rel_l2(plane.cpu(), O.plane_attention(cls[:, :, None, :])) < 1e-6
    assert rel_l2(sa.cpu(), O.slice_attention(sp).reshape(-1)) < 1e-6
    assert rel_l2(maps.cpu(), O.attention_maps(cls[:, :, None, :], sp)) < 1e-6
    assert torch.allclose(plane.sum(-1).cpu(), torch.ones(B * D, heads), atol=1e-5)


@pytest.mark.parametrize("N,batch,layers", [(17, 5, 1), (37, 18, 2), (130, 7, 3), (257, 12, 12), (1370, 2, 3)])
def test_attention_rollout_chain(hip, N, batch, layers):
    """Batched exact-fp32 chain A_0 @ ... @ A_last on row-stochastic maps, ragged N, vs an fp64 product."""
    from oracle import mst_oracle as O
    g = torch.Generator().manual_seed(N + layers)
    maps = [torch.rand(batch, N, N, generator=g).mul(4).softmax(-1) for _ in range(layers)]
    out = hip.attention_rollout([m.cuda() for m in maps])
    ref = O.attention_rollout([m.double() for m in maps])
    assert out.shape == (batch, N, N)
    assert rel_l2(out.cpu().double(), ref) < 2e-6
    assert torch.allclose(out.sum(-1).cpu(), torch.ones(batch, N), atol=1e-4)   # products of stochastic maps stay stochastic
    with pytest.raises(ValueError):
        hip.attention_rollout([maps[0].cuda(), maps[0][:, :, :-1].cuda()])


@pytest.mark.parametrize("std", [0.02, 1.0])
def test_liere_rotation_matches_matrix_exp(hip, std):
    """R = block_diag(exp(A_blk)) (rotary_embedding_torch.py:319-372) vs an fp64 matrix_exp; std 1.0 is the
    reference's own init (generator norm in the hundreds: exercises the scaling-and-squaring)."""
    g = torch.Generator().manual_seed(3)
    vars_ = [torch.randn(120, 33, 1, generator=g) * std for _ in range(2)]
    R = hip.liere_rotation([v.cuda() for v in vars_]).cpu().double()
    pos = torch.arange(33, dtype=torch.float64)
    blocks = []
    for v in vars_:
        flat = v[:, :, 0].double() @ pos
        i, j = torch.tril_indices(16, 16, offset=-1)
        A = torch.zeros(16, 16, dtype=torch.float64)
        A[i, j] = flat
        A[j, i] = -flat
        blocks.append(torch.linalg.matrix_exp(A))
    ref = torch.block_diag(*blocks)
    assert R.shape == (32, 32)
    assert (R - ref).abs().max() < 1e-6
    assert (R @ R.T - torch.eye(32, dtype=torch.float64)).abs().max() < 1e-6   # a rotation


@pytest.mark.parametrize("D,g,size", [(5, 6, (5, 84, 84)), (4, 4, (4, 56, 56)), (3, 7, (6, 50, 37)), (64, 37, (64, 518, 518))])
def test_saliency_accumulate_and_upsample(hip, D, g, size):
    """Head mean + flip-back accumulation + trilinear up-sampling (main_predict.py:72-105, 147-165) vs the oracle / torch."""
    import torch.nn.functional as F
    from oracle import mst_oracle as O
    gen = torch.Generator().manual_seed(D * 100 + g)
    heads, Np = 6, g * g + (3 if g == 7 else 0)               # extra columns past the square grid are ignored (l.94-96)
    low = torch.empty(D, g, g, device="cuda")
    ws = torch.empty(D, device="cuda")
    ref_low = torch.zeros(1, 1, D, g, g)
    ref_ws = torch.zeros(D)
    for k, dims in enumerate([()] + O.TTA_FLIPS):
        maps = torch.rand(D, heads, Np, generator=gen)
        sa = torch.rand(D, generator=gen)
        fm = sum(1 << (a - 2) for a in dims)
        hip.saliency_accumulate(maps.cuda(), sa.cuda(), g, g, fm, low, ws, accumulate=k > 0)
        w_i = maps.mean(1)[:, :g * g].reshape(1, 1, D, g, g)
        ref_low += torch.flip(w_i, dims) if dims else w_i
        ref_ws += torch.flip(sa, (0,)) if 2 in dims else sa
    assert rel_l2(low.cpu(), ref_low[0, 0]) < 1e-6
    assert rel_l2(ws.cpu(), ref_ws) < 1e-6
    out = hip.saliency_upsample(low, size, scale=1.0 / 8)
    assert tuple(out.shape) == tuple(size)
    ref = F.interpolate(ref_low / 8, size=size, mode="trilinear")[0, 0]
    assert rel_l2(out.cpu(), ref) < 2e-6
    assert rel_l2(out.cpu(), O.trilinear_upsample(ref_low / 8, size)[0, 0]) < 2e-6


def test_pos_embed_interp_antialias_matches_hub_register_fixture(hip):
    """interpolate_antialias=True, interpolate_offset=0.0 (the hub's register models) vs the vendored class's own output."""
    import torch.nn.functional as F
    g = load_golden("hub_reg")
    sd = synth.synth_state_dict("s", int(g["seed"]), img_size=518, layerscale=True, chunked=False, num_register_tokens=4)
    pe = sd["encoder.pos_embed"][0]
    for key, gh, gw in (("pos_16x16", 16, 16), ("pos_8x10", 8, 10)):
        got = hip.pos_embed_interp(pe[1:].contiguous().cuda(), 37, gh, gw, 0.0, antialias=True)
        assert rel_l2(got.cpu(), g[key][0, 1:]) < 2e-6
    # up-sampling with antialias (support 2, A = -0.5) and a 1-row target, vs torch
    small = torch.rand(1, 8, 5, 5)
    for size in ((9, 7), (1, 3), (5, 5)):
        ref = F.interpolate(small, size=size, mode="bicubic", antialias=True)[0].permute(1, 2, 0).reshape(-1, 8)
        got = hip.pos_embed_interp(small[0].permute(1, 2, 0).reshape(25, 8).contiguous().cuda(), 5, size[0], size[1], 0.0, antialias=True)
        assert rel_l2(got.cpu(), ref) < 2e-6


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 1, 7, 28, 28), (1, 1, 9, 14, 42), (3, 1, 5, 3, 5), (1, 1, 64, 518, 518)])
def test_slices2rgb_matches_reference_arithmetic(hip, dt, shape):
    """dino.py:10-27 (dead code in the reference; a16 of the scope table): bit-exact index shuffle, incl. the wrap-around padding."""
    from mst.models.dino import slices2rgb
    from oracle import mst_oracle as O
    tdt = DT.get(dt, torch.float32)
    x = rnd(shape, 77).to(tdt)
    got = slices2rgb(x.cuda())
    ref = O.slices2rgb(x)
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref)
    with pytest.raises(AssertionError, match="More than one channel"):
        slices2rgb(torch.zeros(1, 2, 3, 4, 4, device="cuda"))
    if dt == "fp32" and shape[2] == 7:                   # what the reference's own function returned (tests/golden/slices2rgb.npz)
        from conftest import load_golden
        from mst import synth
        g = load_golden("slices2rgb")
        for i in range(3):
            v = synth.synth_volume(tuple(int(u) for u in g[f"shape{i}"]), int(g[f"seed{i}"]))
            assert np.array_equal(slices2rgb(v.cuda()).cpu().numpy(), g[f"out{i}"])


@pytest.mark.parametrize("mode", ["minimum", 0, -3.5])
@pytest.mark.parametrize("src,tgt", [((5, 7, 9), (9, 12, 16)), ((12, 10, 8), (7, 5, 4)), ((6, 20, 9), (11, 8, 9)), ((3, 4, 5), (3, 4, 5)),
                                     ((32, 150, 140), (32, 224, 224)), ((1, 9, 9), (4, 3, 12))])
def test_crop_or_pad_matches_numpy_pad_semantics(hip, mode, src, tgt):
    """8f-4: CropOrPad (augmentations_3d.py:144-195, centre mode) on the device, bit-exact against the oracle (numpy.pad itself)."""
    from mst import preprocess
    from oracle import mst_oracle as O
    x = rnd((2, *src), 90, 3.0)
    got = preprocess.crop_or_pad(x.cuda(), tgt, mode)
    ref = O.crop_or_pad(x, tgt, mode)
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("shape,pct", [((1, 6, 40, 50), (0.5, 99.5)), ((1, 3, 17, 19), (0, 100)), ((1, 32, 224, 224), (0.5, 99.5)),
                                       ((1, 4, 30, 30), (10, 60))])
def test_znormalize_matches_reference_arithmetic(hip, shape, pct):
    """8f-4: ZNormalization (augmentations_3d.py:40-86): the quantile cut-offs are exact order statistics (radix select), so they
    match torch.quantile to the last bit of the interpolation; mean / std differ by summation order only."""
    from mst import preprocess
    from oracle import mst_oracle as O
    x = rnd(shape, 91, 2.0) * 37.0 + 11.0
    x[0, 0, :3, :3] = x.min() - 5            # background plateau below everything, a few saturated voxels above
    x[0, -1, -2:, -2:] = x.max() + 9
    got, st = preprocess.znormalize(x.cuda(), pct, return_stats=True)
    ref = O.znormalize(x, pct)
    mask = (x > x.min()) & (x < x.max())
    cut = torch.quantile(x.masked_select(mask), torch.tensor(pct) / 100.0)
    assert st["count"] == int(mask.sum())
    assert abs(st["cut_lo"] - float(cut[0])) <= 1e-6 * abs(float(cut[0])) and abs(st["cut_hi"] - float(cut[1])) <= 1e-6 * abs(float(cut[1]))
    assert float((got.cpu() - ref).abs().max()) < 2e-5
    with pytest.raises(RuntimeError, match="Standard deviation is 0"):
        preprocess.znormalize(torch.tensor([0.0, 1.0, 1.0, 1.0, 2.0]).reshape(1, 1, 1, 5).cuda())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,lda,scale_cols", [(65536 + 77, 384, 384), (70001, 392, 0), (65536, 384, 1152)])
def test_gemm_weights_in_registers_qkv_shape(dt, M, lda, scale_cols):
    """k_gemm16_wreg.hip (K = 384, N = 3 x 384, M >= 65536: the QKV projection of the bench): ragged last chunk, strided rows,
    scaled column ranges aligned to the 384-column tiles; against the fp64 product of the same 16-bit operands, within the
    rounding of the output type."""
    from mst import hip
    N, K = 1152, 384
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, lda, generator=g).to(dt).cuda()[:, :K]
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    full = torch.zeros(M + 40, N, dtype=dt, device="cuda")          # 40 guard rows behind the ragged chunk
    out = full[:M]
    lib = hip.load()
    hip._check(lib.mst_gemm(a.data_ptr(), hip._DT[dt], lda, hip.ptr(w), K, hip.ptr(b), hip.ptr(out), hip._DT[dt], N, M, N, K,
                            hip.EPI_BIAS, None, 0.125, scale_cols, hip.stream_of(out)), "mst_gemm")
    ref = a.double() @ w.double().t() + b.double()
    ref[:, :scale_cols] *= 0.125
    err = (out.double() - ref).abs()
    tol = (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * ref.abs().clamp_min(1.0) + 1e-3
    assert bool((err <= tol).all()), float((err / tol).max())
    assert float(full[M:].float().abs().sum()) == 0                  # rows beyond M are never written


@pytest.mark.parametrize("N", [384, 768, 1152, 1536])
@pytest.mark.parametrize("M", [1, 31, 33, 256 * 32, 256 * 32 * 2 + 7, 256 * 32 * 3 - 1, 11 * 8 * 32 * 4 + 5, 10 * 8 * 32 * 5])
def test_gemm_weights_in_registers_few_chunks_per_cu(monkeypatch, M, N):
    """The same kernel with the size threshold lowered (MST_GEMM_WREG_MIN_M): CUs with 0, 1, 2, 3, 4, 5 chunks (the counted-vmcnt
    table of the prologue iterations and the vmcnt(0) tail), 1-4 column tiles, ragged last chunk; must equal the mid-tile /
    128 x 128 kernels' result up to the rounding of the output type."""
    from mst import hip
    monkeypatch.setenv("MST_GEMM_WREG_MIN_M", "1")
    K = 384
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    full = torch.zeros(M + 33, N, dtype=torch.bfloat16, device="cuda")
    out = full[:M]
    hip.gemm(a, w, b, out=out, col_scale=0.5, scale_cols=384)
    ref = a.double() @ w.double().t() + b.double()
    ref[:, :384] *= 0.5
    err = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs().clamp_min(1.0) + 1e-3
    assert bool((err <= tol).all()), float((err / tol).max())
    assert float(full[M:].float().abs().sum()) == 0
