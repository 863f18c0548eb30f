"""Training step of DinoV2ClassifierSlice on the HIP path.

The reference trains through torch.autograd (mst/models/base_model.py:148-181 `_step`: ``pred = self(**batch)``, CE loss;
scripts/main_train.py:110-126 Trainer.fit).  Here ``forward`` under ``torch.enable_grad()`` returns logits that carry ONE
autograd node (`_MSTFunction`): its forward runs the model op by op through the C ABI keeping what the backward needs, its
backward produces the gradient of every parameter with the kernels of csrc/k_train.hip (`mst_gemm_ex`, `mst_layernorm_bwd`,
`mst_softmax_rows_bwd`, `mst_act_bwd`, ...).  The loss itself stays the reference's own ``torch.nn.CrossEntropyLoss`` call on the
[B, out_ch] logits (host code).  torch is used for memory only (allocation, views, concatenation / copies of whole tensors).

Default: everything in exact fp32 (fp32 MFMA), whatever ``compute_dtype`` the inference path uses -- the mode the gradient parity bar
(1e-3 against autograd of the CPU oracle on every parameter, tests/test_train_gpu.py) is on.  ``train_precision='fp16' | 'bf16'`` (the
reference's Trainer(precision='16-mixed'), scripts/main_train.py:110-123) runs the blocks' nn.Linear products -- forward, d input, d weight
-- on 16-bit MFMA operands with fp32 accumulation; every other op and everything stored stays fp32.  The attention probabilities of every
block are kept ([n, heads, N, N] fp32: 2.9 GB per block at 64 x 518^2 -- sized for 288 GB of HBM).  RoPE slice transformers and
register-token encoders (at their stored position grid) train; raise: the LieRE variant, ``save_attn`` inside a training forward.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import hip

PATCH = 14


_MP = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}


def _mp(model) -> Optional[torch.dtype]:
    """MFMA operand type of the per-slice encoder's nn.Linear products in the training step: None = exact fp32 MFMA (default),
    bf16 / fp16 = mixed precision -- what the reference's Trainer(precision='16-mixed') (scripts/main_train.py:110-123) does to
    F.linear: 16-bit operands, fp32 accumulation, fp32 everywhere else (LayerNorm, softmax, GELU, residuals, gradients in memory)."""
    return _MP[getattr(model, "train_precision", "fp32")]


def _lin_fwd(x: torch.Tensor, lin, mp: Optional[torch.dtype] = None, keep: Optional[dict] = None, **kw) -> torch.Tensor:
    if mp is None:
        return hip.gemm(x, lin.weight.detach(), lin.bias.detach(), **kw)
    x16 = hip.cvt16(x, mp)
    if keep is not None and x.shape[0] <= 12288:         # the weight gradient of this product reads the same image (mst_conv_wgrad16 path)
        keep[id(lin)] = x16
    return hip.gemm(x16, hip.cvt16(lin.weight.detach(), mp), lin.bias.detach(), out_dtype=torch.float32, **kw)


class _Grads:
    def __init__(self, mp: Optional[torch.dtype] = None):
        self.by_param: Dict[int, torch.Tensor] = {}
        self.mp = mp

    def put(self, param, g: torch.Tensor):
        g = g.reshape(param.shape)
        if id(param) in self.by_param:
            hip.axpby_cols(g.reshape(1, -1), self.by_param[id(param)].reshape(1, -1))
        else:
            self.by_param[id(param)] = g

    def lin_bwd(self, dY: torch.Tensor, X: torch.Tensor, lin, need_dx: bool = True, X16: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """nn.Linear backward: d weight = dY^T . X, d bias = column sums of dY, returns dX = dY . W."""
        M, N = dY.shape
        K = X.shape[1]
        dev = dY.device
        if self.mp is not None and N % 128 == 0 and K % 128 == 0 and M >= 64:
            return self._lin_bwd_16(dY, X, lin, need_dx, X16)
        # d weight: an [N, K] output reduced over M rows is 36-144 tiles walking thousands of rows each; split the rows into up to
        # 16 slabs (more workgroups than CUs), partial products reduced by mst_colsum
        sp = next((d for d in (16, 8, 4, 2) if M % d == 0 and M // d >= 64), 1)
        if sp > 1:
            part = torch.empty((sp, N * K), dtype=torch.float32, device=dev)
            ch = M // sp
            hip.gemm_ex(dY, X, part, N, K, ch, sa=(1, N), sb=(K, 1), sc=(K, 1), nb=(sp, 1), ba=(ch * N, 0), bb=(ch * K, 0), bc=(N * K, 0))
            dW = hip.colsum(part, torch.zeros(N * K, dtype=torch.float32, device=dev)).view(N, K)
        else:
            dW = torch.empty((N, K), dtype=torch.float32, device=dev)
            hip.gemm_ex(dY, X, dW, N, K, M, sa=(1, N), sb=(K, 1), sc=(K, 1))
        self.put(lin.weight, dW)
        if getattr(lin, "bias", None) is not None:
            self.put(lin.bias, hip.colsum(dY, torch.zeros(N, dtype=torch.float32, device=dev)))
        if not need_dx:
            return None
        dX = torch.empty((M, K), dtype=torch.float32, device=dev)
        hip.gemm_ex(dY, lin.weight.detach(), dX, M, K, N, sa=(N, 1), sb=(K, 1), sc=(K, 1))
        return dX

    def _lin_bwd_16(self, dY: torch.Tensor, X: torch.Tensor, lin, need_dx: bool, X16: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """The same three results on 16-bit MFMA operands (fp32 accumulation, fp32 results): operands are rounded into scratch images right
        before each product."""
        M, N = dY.shape
        K = X.shape[1]
        dev = dY.device
        mp = self.mp
        dY16 = hip.cvt16(dY, mp)
        if M <= 12288:
            # d weight = dY^T . X is the weight gradient of a 1 x 1 "convolution" over M one-pixel images: mst_conv_wgrad16 reads both operands
            # row-major (token-major) and transposes the fragments in the LDS read -- no transposed operand images, token-split partial
            # products (1 x 16 x 224^2: 14.4 -> 10.9 ms per step against the form below)
            x16 = X16 if (X16 is not None and X16.dtype == mp and X16.shape == X.shape) else hip.cvt16(X, mp)   # the forward's image, if it was kept
            self.put(lin.weight, hip.conv_wgrad(dY16, x16.view(M, 1, 1, K), 1, 1, 0))
        else:
            # many tokens: TRANSPOSED operand images (both operands contiguous along the token index) through the 128 x 128 x 64 GEMM with
            # 16-byte fragment reads, split over the tokens (mst_gemm16_splitk): 3 % faster at 16,448 tokens than the transposing reads
            tiles = (N // 128) * (K // 128)
            sp = max(1, min(64, 512 // tiles))
            kc = -(-M // (sp * 64)) * 64                                 # token rows per split, a multiple of the K-step
            part = hip.gemm16_splitk(hip.cvt16(dY, mp, transpose=True, rows_pad=kc * sp), hip.cvt16(X, mp, transpose=True, rows_pad=kc * sp), sp)
            self.put(lin.weight, hip.colsum(part.view(sp, N * K), torch.zeros(N * K, dtype=torch.float32, device=dev)).view(N, K))
        if getattr(lin, "bias", None) is not None:
            self.put(lin.bias, hip.colsum(dY, torch.zeros(N, dtype=torch.float32, device=dev)))
        if not need_dx:
            return None
        # dX = dY . W: W^T [K, N] plays nn.Linear's weight
        return hip.gemm(dY16, hip.cvt16(lin.weight.detach(), mp, transpose=True), None, out_dtype=torch.float32)

    def ln_bwd(self, x, x_stride, ln, dy, dy_stride, dres, dres_stride, dx, dx_stride, rows, cols, eps):
        dev = dy.device
        dg = torch.zeros(cols, dtype=torch.float32, device=dev)
        db = torch.zeros(cols, dtype=torch.float32, device=dev)
        hip.layernorm_bwd(x, x_stride, ln.weight.detach(), dy, dy_stride, dres, dres_stride, dx, dx_stride, dg, db, rows, cols, eps)
        self.put(ln.weight, dg)
        self.put(ln.bias, db)


def _attention_fwd(qkv: torch.Tensor, nb: int, L: int, heads: int, hd: int, alpha: float, mask: Optional[torch.Tensor]):
    """softmax(alpha * q k^T + mask) v on packed rows [nb*L, 3*heads*hd] (q | k | v, head-major).  Returns (out [nb*L, heads*hd], P)."""
    e = heads * hd
    dev = qkv.device
    P = torch.empty((nb, heads, L, L), dtype=torch.float32, device=dev)
    hip.gemm_ex(qkv, qkv, P, L, L, hd, sa=(3 * e, 1), sb=(1, 3 * e), sc=(L, 1), nb=(nb, heads), ba=(L * 3 * e, hd),
                bb=(L * 3 * e, hd), bc=(heads * L * L, L * L), alpha=alpha, offs=(0, e, 0))
    hip.softmax_rows(P, mask, heads * L)
    out = torch.empty((nb * L, e), dtype=torch.float32, device=dev)
    hip.gemm_ex(P, qkv, out, L, hd, L, sa=(L, 1), sb=(3 * e, 1), sc=(e, 1), nb=(nb, heads), ba=(heads * L * L, L * L),
                bb=(L * 3 * e, hd), bc=(L * e, hd), offs=(0, 2 * e, 0))
    return out, P


def _attention_bwd(dout: torch.Tensor, qkv: torch.Tensor, P: torch.Tensor, nb: int, L: int, heads: int, hd: int, alpha: float,
                   q_scale: float) -> torch.Tensor:
    """Gradient w.r.t. the packed qkv rows.  `alpha` multiplied the scores inside the attention; `q_scale` is a factor the stored
    q already carries from the projection's epilogue (its gradient flows to the un-scaled projection output)."""
    e = heads * hd
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    bP, bq, bo = (heads * L * L, L * L), (L * 3 * e, hd), (L * e, hd)
    # dV = P^T . dO
    hip.gemm_ex(P, dout, dqkv, L, hd, L, sa=(1, L), sb=(e, 1), sc=(3 * e, 1), nb=(nb, heads), ba=bP, bb=bo, bc=bq, offs=(0, 0, 2 * e))
    # dP = dO . V^T
    dP = torch.empty_like(P)
    hip.gemm_ex(dout, qkv, dP, L, L, hd, sa=(e, 1), sb=(1, 3 * e), sc=(L, 1), nb=(nb, heads), ba=bo, bb=bq, bc=bP, offs=(0, 2 * e, 0))
    hip.softmax_rows_bwd(P, dP, alpha)                  # dS (scores before the softmax), alpha folded in
    # dQ = dS . K  (times the epilogue factor of the stored q);  dK = dS^T . Q
    hip.gemm_ex(dP, qkv, dqkv, L, hd, L, sa=(L, 1), sb=(3 * e, 1), sc=(3 * e, 1), nb=(nb, heads), ba=bP, bb=bq, bc=bq,
                alpha=q_scale, offs=(0, e, 0))
    hip.gemm_ex(dP, qkv, dqkv, L, hd, L, sa=(1, L), sb=(3 * e, 1), sc=(3 * e, 1), nb=(nb, heads), ba=bP, bb=bq, bc=bq, offs=(0, 0, e))
    return dqkv


def fusion_fwd(model, tok: torch.Tensor, B: int, D: int, e: int, hs: int, mask: Optional[torch.Tensor]):
    """Slice Transformer (one pre-norm nn.TransformerEncoderLayer with ReLU + final LayerNorm; dino.py:134-153, resnet.py:180-190)
    over [CLS | D slice tokens] per volume; `model` carries slice_fusion.layers[0], slice_fusion.norm and cls_token.  Returns the
    CLS features [B, e] and what the backward needs."""
    dev = tok.device
    lay = model.slice_fusion.layers[0]
    L = D + 1
    hd = e // hs
    xs = torch.cat([model.cls_token.detach().expand(B, 1, e), tok.view(B, D, e)], dim=1).contiguous().view(B * L, e)
    mk = None
    if mask is not None:
        mk = torch.cat([torch.zeros((B, 1), dtype=torch.uint8, device=dev), mask.to(dev).to(torch.uint8)], dim=1).contiguous()
    t = {"xs": xs, "mask": mk}
    t["y1"] = hip.layernorm(xs, lay.norm1.weight.detach(), lay.norm1.bias.detach(), 1e-5)
    t["qkv"] = hip.gemm(t["y1"], lay.self_attn.in_proj_weight.detach(), lay.self_attn.in_proj_bias.detach())
    rot = getattr(lay.self_attn, "rotary_positional_encoding", None)
    t["rope"] = None
    if rot is not None:                                  # RoPE on q and k (transformer_blocks.py:262-264); the rotated rows are what the backward needs
        if not hasattr(rot, "freqs"):
            raise NotImplementedError("training step: the LieRE variant of the slice transformer is not on the HIP backward (its rotation "
                                      "generators are learned: the adjoint of the matrix exponential is not built)")
        t["rope"] = rot.freqs.detach().to(dev, torch.float32).contiguous()
        hip.rope_rows(t["qkv"], L, hs, hd, t["rope"], 1.0)
    t["ao"], t["P"] = _attention_fwd(t["qkv"], B, L, hs, hd, 1.0 / math.sqrt(hd), mk)
    xs1 = xs.clone()
    hip.axpby_cols(_lin_fwd(t["ao"], lay.self_attn.out_proj), xs1)
    t["xs1"] = xs1
    t["y2"] = hip.layernorm(xs1, lay.norm2.weight.detach(), lay.norm2.bias.detach(), 1e-5)
    t["f1"] = _lin_fwd(t["y2"], lay.linear1)
    t["r"] = hip.act_fwd(t["f1"], 1)
    xs2 = xs1.clone()
    hip.axpby_cols(_lin_fwd(t["r"], lay.linear2), xs2)
    t["xs2"] = xs2
    feat = hip.layernorm_rows(xs2, L * e, B, e, model.slice_fusion.norm.weight.detach(), model.slice_fusion.norm.bias.detach(), 1e-5)
    return feat, t


def fusion_bwd(G: "_Grads", model, t, dfeat: torch.Tensor, B: int, D: int, e: int, hs: int) -> torch.Tensor:
    """Backward of `fusion_fwd`: parameter gradients into G, returns the gradient of the slice tokens [B*D, e]."""
    dev = dfeat.device
    lay = model.slice_fusion.layers[0]
    L = D + 1
    hd = e // hs
    dxs2 = torch.zeros((B * L, e), dtype=torch.float32, device=dev)
    G.ln_bwd(t["xs2"], L * e, model.slice_fusion.norm, dfeat, e, None, 0, dxs2, L * e, B, e, 1e-5)      # row 0 of every volume
    dr = G.lin_bwd(dxs2, t["r"], lay.linear2)
    hip.act_bwd(t["f1"], dr, 1)
    dy2 = G.lin_bwd(dr, t["y2"], lay.linear1)
    dxs1 = torch.empty_like(dxs2)
    G.ln_bwd(t["xs1"], e, lay.norm2, dy2, e, dxs2, e, dxs1, e, B * L, e, 1e-5)
    dao = G.lin_bwd(dxs1, t["ao"], lay.self_attn.out_proj)
    dqkv = _attention_bwd(dao, t["qkv"], t["P"], B, L, hs, hd, 1.0 / math.sqrt(hd), 1.0)
    if t["rope"] is not None:
        hip.rope_rows(dqkv, L, hs, hd, t["rope"], -1.0)    # adjoint of the rotation: gradients of the un-rotated projections
    sa = lay.self_attn
    dW = torch.empty_like(sa.in_proj_weight)
    hip.gemm_ex(dqkv, t["y1"], dW, 3 * e, e, B * L, sa=(1, 3 * e), sb=(e, 1), sc=(e, 1))
    G.put(sa.in_proj_weight, dW)
    G.put(sa.in_proj_bias, hip.colsum(dqkv, torch.zeros(3 * e, dtype=torch.float32, device=dev)))
    dy1 = torch.empty((B * L, e), dtype=torch.float32, device=dev)
    hip.gemm_ex(dqkv, sa.in_proj_weight.detach(), dy1, B * L, e, 3 * e, sa=(3 * e, 1), sb=(e, 1), sc=(e, 1))
    dxs = torch.empty_like(dxs2)
    G.ln_bwd(t["xs"], e, lay.norm1, dy1, e, dxs1, e, dxs, e, B * L, e, 1e-5)
    dcls = torch.zeros(e, dtype=torch.float32, device=dev)
    hip._check(hip.load().mst_colsum(hip.ptr(dxs), L * e, None, 0, B, e, hip.ptr(dcls), hip.stream_of(dxs)), "mst_colsum")
    G.put(model.cls_token, dcls)
    return dxs.view(B, L, e)[:, 1:].contiguous().view(B * D, e)


def forward_train(model, source: torch.Tensor, mask: Optional[torch.Tensor], without_linear: bool):
    import torch.nn as nn
    if model.rotary is not None and model.rotary != "RoPE":
        raise NotImplementedError("training step: the LieRE variant of the slice transformer is not on the HIP backward (its rotation "
                                  "generators are learned: the adjoint of the matrix exponential is not built); RoPE is")
    enc = model.encoder
    R = int(enc.num_register_tokens)                     # register tokens (vision_transformer.py:222-230): extra prefix rows without a position
    dev = model.device
    x = source.to(dev)
    B, C, D0, H, W = x.shape
    if C != 1:
        x = x.permute(0, 2, 1, 3, 4)
    D = D0 * C
    vol = x.reshape(B * D, H, W).float().contiguous()
    assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
    assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
    sv = {"B": B, "D": D, "H": H, "W": W, "vol": vol, "without_linear": without_linear}
    E, heads = enc.embed_dim, enc.num_heads
    n = B * D
    gh, gw = H // PATCH, W // PATCH
    Np = gh * gw
    N = 1 + R + Np
    M = n * N
    sv["R"] = R
    # ---- tokens (patch_embed.py:68-81; vision_transformer.py:213-232)
    pos = enc.pos_embed.detach()[0]
    n_stored = pos.shape[0] - 1
    Mg = int(math.isqrt(n_stored))
    sv["interp"] = not (Np == n_stored and H == W)
    if sv["interp"] and R:
        # register encoders are the hub's dinov2_vit*14_reg (anti-aliased, size-based resampling: models/dino.py::_pos_patch); the
        # adjoint of that filter is not built -- train them at the stored grid (518 x 518 for the hub weights)
        raise NotImplementedError("training step: register-token encoders train at their stored position grid only (the adjoint of the "
                                  "anti-aliased position resampling is not on the HIP backward)")
    pos_patch = hip.pos_embed_interp(pos[1:].contiguous(), Mg, gh, gw, 0.1) if sv["interp"] else pos[1:].contiguous()
    prefix = torch.zeros((1 + R, E), dtype=torch.float32, device=dev)
    prefix[:1].copy_(pos[:1])
    hip.axpby_cols(enc.cls_token.detach().reshape(1, E), prefix[:1])
    if R:
        prefix[1:].copy_(enc.register_tokens.detach()[0])
    wsum = torch.zeros((E, PATCH * 16), dtype=torch.float32, device=dev)            # three identical input channels: one summed kernel
    w = enc.patch_embed.proj.weight.detach()
    for c in range(3):
        hip.axpby_cols(w[:, c].contiguous().view(E * PATCH, PATCH), wsum, x_stride=PATCH, y_stride=16, rows=E * PATCH, cols=PATCH)
    xt = hip.patch_embed(vol, wsum, enc.patch_embed.proj.bias.detach(), prefix, pos_patch).view(M, E)
    # ---- blocks (block.py:89-114)
    blocks = []
    mp = _mp(model)
    for blk in enc.block_list():
        s = {"x0": xt, "x16": {}}
        k16 = s["x16"] if mp is not None else None
        s["xn1"] = hip.layernorm(xt, blk.norm1.weight.detach(), blk.norm1.bias.detach(), 1e-6)
        s["qkv"] = _lin_fwd(s["xn1"], blk.attn.qkv, mp, k16, col_scale=0.125, scale_cols=E)            # q * head_dim^-0.5 (attention.py:60)
        s["a"], s["P"] = _attention_fwd(s["qkv"], n, N, heads, 64, 1.0, None)
        s["br1"] = _lin_fwd(s["a"], blk.attn.proj, mp, k16)
        x1 = xt.clone()
        hip.axpby_cols(s["br1"], x1, g=blk.ls1.gamma.detach() if hasattr(blk, "ls1") else None)
        s["x1"] = x1
        s["xn2"] = hip.layernorm(x1, blk.norm2.weight.detach(), blk.norm2.bias.detach(), 1e-6)
        s["hpre"] = _lin_fwd(s["xn2"], blk.mlp.fc1, mp, k16)
        s["hact"] = hip.act_fwd(s["hpre"], 0)
        s["br2"] = _lin_fwd(s["hact"], blk.mlp.fc2, mp, k16)
        x2 = x1.clone()
        hip.axpby_cols(s["br2"], x2, g=blk.ls2.gamma.detach() if hasattr(blk, "ls2") else None)
        blocks.append(s)
        xt = x2
    sv["blocks"], sv["xL"] = blocks, xt
    emb = hip.layernorm_rows(xt, N * E, n, E, enc.norm.weight.detach(), enc.norm.bias.detach(), 1e-6)     # CLS rows
    sv["emb"] = emb
    # ---- across-slice stage (dino.py:134-166)
    e = model.emb_ch
    tok = _lin_fwd(emb, model.bottleneck) if hasattr(model, "bottleneck") else emb
    if hasattr(model, "slice_pos_emb"):
        tok = tok.clone()
        p = model.slice_pos_emb.weight.detach()[:D].contiguous()
        for b in range(B):
            hip.axpby_cols(p, tok[b * D:(b + 1) * D])
    ft = model.slice_fusion_type
    if ft == "transformer":
        feat, sv["fusion"] = fusion_fwd(model, tok, B, D, e, 12, mask)
    elif ft == "linear":
        feat = tok.reshape(B, D * e)
    else:                                                                        # 'average' (dino.py:156-157)
        feat = torch.zeros((B, e), dtype=torch.float32, device=dev)
        for b in range(B):
            hip.colsum(tok[b * D:(b + 1) * D], feat[b])
        hip.axpby_cols(feat, feat, alpha=1.0 / D, beta=0.0)
    sv["tok"], sv["feat"] = tok, feat
    if without_linear or isinstance(model.linear, nn.Identity):
        sv["head"] = False
        return feat, sv
    if ft == "linear" and feat.shape[1] != model.linear.weight.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{feat.shape[1]} and "
                           f"{model.linear.weight.shape[1]}x{model.linear.weight.shape[0]})")
    sv["head"] = True
    return _lin_fwd(feat, model.linear), sv


def backward_train(model, sv, dout: torch.Tensor) -> Dict[int, torch.Tensor]:
    G = _Grads()
    enc = model.encoder
    dev = dout.device
    B, D, H, W = sv["B"], sv["D"], sv["H"], sv["W"]
    e = model.emb_ch
    dfeat = G.lin_bwd(dout, sv["feat"], model.linear) if sv["head"] else dout
    ft = model.slice_fusion_type
    if ft == "transformer":
        dtok = fusion_bwd(G, model, sv["fusion"], dfeat, B, D, e, 12)
    elif ft == "linear":
        dtok = dfeat.reshape(B * D, e).contiguous()
    else:
        dtok = dfeat[:, None, :].expand(B, D, e).contiguous().view(B * D, e)
        hip.axpby_cols(dtok, dtok, alpha=1.0 / D, beta=0.0)
    if hasattr(model, "slice_pos_emb"):
        dp = torch.zeros_like(model.slice_pos_emb.weight)
        acc = torch.zeros(D * e, dtype=torch.float32, device=dev)
        hip.colsum(dtok.view(B, D * e), acc)
        dp[:D].copy_(acc.view(D, e))
        G.put(model.slice_pos_emb.weight, dp)
    demb = G.lin_bwd(dtok, sv["emb"], model.bottleneck) if hasattr(model, "bottleneck") else dtok
    if not any(p.requires_grad for p in enc.parameters()):
        return G.by_param                                                        # frozen encoder (dino.py:65-67)
    # ---- encoder
    G.mp = _mp(model)                                    # the blocks' nn.Linear products on 16-bit operands, if asked for
    E, heads = enc.embed_dim, enc.num_heads
    n = B * D
    gh, gw = H // PATCH, W // PATCH
    Np = gh * gw
    R = sv["R"]
    N = 1 + R + Np
    M = n * N
    dx = torch.zeros((M, E), dtype=torch.float32, device=dev)
    G.ln_bwd(sv["xL"], N * E, enc.norm, demb, E, None, 0, dx, N * E, n, E, 1e-6)                         # CLS rows only
    for blk, s in zip(reversed(enc.block_list()), reversed(sv["blocks"])):
        # x2 = x1 + ls2 * fc2(gelu(fc1(norm2 x1)))
        dbr = dx
        if hasattr(blk, "ls2"):
            G.put(blk.ls2.gamma, hip.colsum(dx, torch.zeros(E, dtype=torch.float32, device=dev), b=s["br2"]))
            dbr = torch.empty_like(dx)
            hip.axpby_cols(dx, dbr, g=blk.ls2.gamma.detach(), beta=0.0)
        x16 = s.get("x16", {})
        dh = G.lin_bwd(dbr, s["hact"], blk.mlp.fc2, X16=x16.get(id(blk.mlp.fc2)))
        hip.act_bwd(s["hpre"], dh, 0)
        dxn2 = G.lin_bwd(dh, s["xn2"], blk.mlp.fc1, X16=x16.get(id(blk.mlp.fc1)))
        dx1 = torch.empty_like(dx)
        G.ln_bwd(s["x1"], E, blk.norm2, dxn2, E, dx, E, dx1, E, M, E, 1e-6)
        # x1 = x0 + ls1 * proj(attn(qkv(norm1 x0)))
        dbr = dx1
        if hasattr(blk, "ls1"):
            G.put(blk.ls1.gamma, hip.colsum(dx1, torch.zeros(E, dtype=torch.float32, device=dev), b=s["br1"]))
            dbr = torch.empty_like(dx1)
            hip.axpby_cols(dx1, dbr, g=blk.ls1.gamma.detach(), beta=0.0)
        da = G.lin_bwd(dbr, s["a"], blk.attn.proj, X16=x16.get(id(blk.attn.proj)))
        dqkv = _attention_bwd(da, s["qkv"], s["P"], n, N, heads, 64, 1.0, 0.125)
        dxn1 = G.lin_bwd(dqkv, s["xn1"], blk.attn.qkv, X16=x16.get(id(blk.attn.qkv)))
        dx0 = torch.empty_like(dx)
        G.ln_bwd(s["x0"], E, blk.norm1, dxn1, E, dx1, E, dx0, E, M, E, 1e-6)
        dx = dx0
    # ---- tokens
    dcls = torch.zeros(E, dtype=torch.float32, device=dev)
    hip._check(hip.load().mst_colsum(hip.ptr(dx), N * E, None, 0, n, E, hip.ptr(dcls), hip.stream_of(dx)), "mst_colsum")
    G.put(enc.cls_token, dcls.clone())
    if R:                                                # d register_tokens[r] = sum over slices of row 1 + r
        dreg = torch.zeros((R, E), dtype=torch.float32, device=dev)
        for r in range(R):
            hip._check(hip.load().mst_colsum(hip.ptr(dx) + (1 + r) * E * 4, N * E, None, 0, n, E, hip.ptr(dreg) + r * E * 4, hip.stream_of(dx)), "mst_colsum")
        G.put(enc.register_tokens, dreg.view(1, R, E))
    dpatch = dx.view(n, N, E)[:, 1 + R:].contiguous().view(n * Np, E)
    dposp = torch.zeros(Np * E, dtype=torch.float32, device=dev)
    hip.colsum(dpatch.view(n, Np * E), dposp)
    dpos = torch.zeros_like(enc.pos_embed)
    dpos[0, 0].copy_(dcls)
    if sv["interp"]:
        Mg = int(math.isqrt(enc.pos_embed.shape[1] - 1))
        hip.pos_embed_interp_bwd(dposp.view(Np, E), Mg, gh, gw, 0.1, dpos[0, 1:])
    else:
        dpos[0, 1:].copy_(dposp.view(Np, E))
    G.put(enc.pos_embed, dpos)
    G.put(enc.patch_embed.proj.bias, hip.colsum(dpatch, torch.zeros(E, dtype=torch.float32, device=dev)))
    col = hip.im2col14(sv["vol"])
    dW = torch.empty((E, 196), dtype=torch.float32, device=dev)
    hip.gemm_ex(dpatch, col, dW, E, 196, n * Np, sa=(1, E), sb=(196, 1), sc=(196, 1))
    G.put(enc.patch_embed.proj.weight, dW.view(E, 1, PATCH, PATCH).expand(E, 3, PATCH, PATCH).contiguous())
    return G.by_param


class _MSTFunction(torch.autograd.Function):
    """One autograd node for the whole model: inputs are the parameters (so that autograd, DDP hooks and optimisers see
    ordinary ``.grad`` accumulation), output the logits (or features)."""

    @staticmethod
    def forward(ctx, model, source, mask, without_linear, *params):
        with torch.no_grad():
            out, saved = forward_train(model, source, mask, without_linear)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            grads = backward_train(ctx.model, ctx.saved, dout.contiguous().float())
        ctx.saved = None
        out: List[Optional[torch.Tensor]] = []
        for p, need in zip(ctx.params, ctx.needs_input_grad[4:]):
            out.append(grads.get(id(p)) if need else None)
        return (None, None, None, None, *out)


def forward_with_grad(model, source, mask, without_linear: bool):
    params = [p for p in model.parameters()]
    for p in params:
        if p.dtype != torch.float32 or p.device.type != "cuda":
            raise RuntimeError("training step: parameters must be fp32 on the MI355X (model.float().cuda()); there is no CPU fallback")
    return _MSTFunction.apply(model, source, mask, without_linear, *params)
