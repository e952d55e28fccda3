"""Swin Transformer (timm `swin_*_patch4_window7_224`) on the HIP kernels -- the DEFAULT backbone of both shipped configs of the reference
(`timm-swin_base_patch4_window7_224`: configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by `timm.create_model` in
models/classifier/classify_model.py:49-54 and models/faceX/backbone/timm_wrapper.py:16-21).

`create_model(name, num_classes=C)` mirrors the timm factory call; the module has timm's state_dict names (oracle/swin_ref.py restates the architecture and is pinned
against transformers.SwinModel), so reference checkpoints load.  First form of this family here: the model is a torch Module whose arithmetic runs in autograd nodes over
the library's kernels -- LayerNorm (`vdk_layernorm_fwd/bwd`), every Linear on the bf16 MFMA GEMM with its bias / GELU / residual epilogues (`vdk_gemm_bf16_nt`, weight
gradients in the TN form), the 49-token window attention with relative-position bias and shifted-window masks (`vdk_window_attention_fwd/bwd`), the pooled head on the fp32
MFMA.  The window partition and the cyclic shift are a row index the attention kernels follow (`rowidx`: the token rows stay in image order, no gather copies); patch
merging is an index permutation.  Arithmetic = the reference's autocast path: bf16 operands,
fp32 accumulation, fp32 residual stream and master weights.

Round 4: `SwinTransformer` is now a thin module over the NATIVE engine (csrc/swin_engine.hip: `vdk_swin_forward` / `vdk_swin_backward`, one C call each over a flat
parameter space, like the ViT engine), so `vit.FusedTrainStep` (fused clip + SGD + EMA, bucketed gradient all-reduce) and `face.FaceTrainStep` drive it.  The first form --
a torch Module whose arithmetic runs in ~35 autograd nodes per block over the same kernels -- stays as `SwinTransformerAutograd` (`create_model(..., native=False)`): the
tests require both to produce the same logits and gradients, it is the second implementation the engine is checked against.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, ops
from ._abi import ACT_DGELU, ACT_GELU

WS = 7
N = WS * WS


@dataclass(frozen=True)
class SwinSpec:
    img_size: int = 224
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 128
    depths: Sequence[int] = (2, 2, 18, 2)
    heads: Sequence[int] = (4, 8, 16, 32)
    ln_eps: float = 1e-5


TIMM_SWINS = {
    "swin_tiny_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24)),
    "swin_small_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 18, 2), heads=(3, 6, 12, 24)),
    "swin_base_patch4_window7_224": dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32)),
    "swin_large_patch4_window7_224": dict(embed_dim=192, depths=(2, 2, 18, 2), heads=(6, 12, 24, 48)),
}


# ---- index helpers (no arithmetic) ------------------------------------------------------------------------------------------------------------
def _partition(x: torch.Tensor, ws: int) -> torch.Tensor:          # [B, H, W, C] -> [B * nW * ws * ws, C]
    B, H, W, Cc = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, Cc).permute(0, 1, 3, 2, 4, 5).reshape(-1, Cc)


def _rel_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = _partition(img, ws).view(-1, ws * ws)
    m = mw[:, None, :] - mw[:, :, None]
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0).contiguous()


def _wgrad(be, dyb: torch.Tensor, xb: torch.Tensor) -> torch.Tensor:
    """dW f32 [N, K] = dY^T X for bf16 dY [T, N], X [T, K]: the TN GEMM straight from both tensors when T % 64 == 0, else through zero-padded transposes"""
    T, Nn = dyb.shape
    K = xb.shape[1]
    if T % 64 == 0 and Nn % 8 == 0 and K % 8 == 0:
        tiles = ((Nn + 255) // 256) * ((K + 255) // 256)
        sk = max(1, min(256 // tiles, (T // 64) // 4, 64))
        return ops.gemm_nt(dyb, xb, out_dtype=torch.float32, trans=True, splitk=sk, backend=be)
    dyt = ops.transpose_pad(dyb, backend=be)                    # [N, Tp]
    xt = ops.transpose_pad(xb, backend=be)                      # [K, Tp]
    return ops.gemm_nt(dyt, xt, out_dtype=torch.float32, backend=be)


def _bf(be, t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.bfloat16 else ops.cast_bf16(t.contiguous(), backend=be)


# ---- autograd nodes over the kernels ----------------------------------------------------------------------------------------------------------
class _LN(torch.autograd.Function):
    """LayerNorm over the last dimension of f32 rows -> bf16 (a GEMM operand) or f32 (the residual stream)"""

    @staticmethod
    def forward(ctx, x, w, b, eps, out_dtype, be):
        x2 = x.contiguous().view(-1, w.numel())
        y, mean, rstd = ops.layernorm_fwd(x2, w.detach(), b.detach(), eps, out_dtype, backend=be)
        ctx.save_for_backward(x2, mean, rstd, w)
        ctx.be, ctx.shape = be, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        dy2 = dy.contiguous().view(x2.shape)
        dx, _, dg, db = ops.layernorm_bwd(dy2, x2, mean, rstd, w.detach(), want_bf16=False, backend=ctx.be)
        return dx.view(ctx.shape), dg, db, None, None, None


class _LNRes(torch.autograd.Function):
    """(LayerNorm(x) as bf16, x): the block's pre-norm with the shortcut routed through the same node, so that the backward adds the shortcut's gradient inside the
    LayerNorm backward kernel (its `dres` input) instead of an autograd accumulation pass over the f32 stream.  The f32 dx carries its bf16 copy (written by the same kernel)
    as `_vdk_bf16`: the next consumer is a GEMM and takes it instead of casting (see _grad_bf16)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, be):
        x2 = x.contiguous().view(-1, w.numel())
        y, mean, rstd = ops.layernorm_fwd(x2, w.detach(), b.detach(), eps, torch.bfloat16, backend=be)
        ctx.save_for_backward(x2, mean, rstd, w)
        ctx.be, ctx.shape = be, x.shape
        return y.view(x.shape), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x2, mean, rstd, w = ctx.saved_tensors
        dy2 = dy.contiguous().view(x2.shape)
        dr = dres.contiguous().view(x2.shape) if dres is not None else None
        dx, dxb, dg, db = ops.layernorm_bwd(dy2, x2, mean, rstd, w.detach(), dres=dr, want_bf16=True, backend=ctx.be)
        dx = dx.view(ctx.shape)
        dx._vdk_bf16 = (dxb, dx._version, dx.data_ptr())      # valid only while this very tensor reaches the consumer unmodified (see _grad_bf16)
        return dx, dg, db, None, None


def _grad_bf16(be, dy: torch.Tensor, rows: int) -> torch.Tensor:
    """the bf16 [rows, C] operand of a gradient: the copy its producer attached (_LNRes.backward), else a cast"""
    c = getattr(dy, "_vdk_bf16", None)
    # the attached copy is trusted only if the gradient is still the producer's tensor, untouched: an in-place accumulation (a second consumer of the activation, a
    # tensor hook) bumps _version, a re-materialised sum has another data pointer -- then the copy is stale and the gradient is cast again
    if c is not None and c[0].numel() == dy.numel() and c[1] == dy._version and c[2] == dy.data_ptr():
        return c[0].view(rows, -1)
    return _bf(be, dy.contiguous().view(rows, -1))


class _Lin(torch.autograd.Function):
    """y = x W^T (+ b) (+ residual): bf16 operands on the MFMA GEMM; y bf16, or f32 when it joins the residual stream"""

    @staticmethod
    def forward(ctx, x, w, b, residual, out_dtype, be):
        xb = _bf(be, x.contiguous().view(-1, x.shape[-1]))
        wb = ops.cast_bf16(w.detach().contiguous(), backend=be)
        res = residual.contiguous().view(-1, w.shape[0]) if residual is not None else None
        y = ops.gemm_nt(xb, wb, out_dtype=out_dtype, bias=b.detach() if b is not None else None, residual=res, backend=be)
        ctx.save_for_backward(xb, w)
        ctx.be, ctx.has_b, ctx.has_r, ctx.xshape, ctx.xdtype = be, b is not None, residual is not None, x.shape, x.dtype
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        be = ctx.be
        dyb = _grad_bf16(be, dy, dy.numel() // w.shape[0])
        wt = ops.transpose_cast(w.detach().contiguous(), backend=be)        # bf16 [in, out]: B operand of dX = dY W
        dx = ops.gemm_nt(dyb, wt[:, :w.shape[0]] if wt.shape[1] != w.shape[0] else wt, out_dtype=torch.bfloat16 if ctx.xdtype == torch.bfloat16 else torch.float32, backend=be)
        dw = _wgrad(be, dyb, xb)
        db = ops.colsum_bf16(dyb, backend=be) if ctx.has_b else None
        dres = dy if ctx.has_r else None
        return dx.view(ctx.xshape), dw, db, dres, None, None


class _Mlp(torch.autograd.Function):
    """x + fc2(gelu(fc1(h))): fc1 with the GELU epilogue (pre-activation kept), fc2 with the residual epilogue; backward with the dGELU epilogue on the fc2 input-gradient GEMM"""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, x, be):
        hb = h.contiguous().view(-1, h.shape[-1])
        w1b = ops.cast_bf16(w1.detach().contiguous(), backend=be); w2b = ops.cast_bf16(w2.detach().contiguous(), backend=be)
        u = torch.empty((hb.shape[0], w1.shape[0]), dtype=torch.bfloat16, device=h.device)
        g = ops.gemm_nt(hb, w1b, bias=b1.detach(), act=ACT_GELU, aux=u, backend=be)
        y = ops.gemm_nt(g, w2b, out_dtype=torch.float32, bias=b2.detach(), residual=x.contiguous().view(-1, w2.shape[0]), backend=be)
        ctx.save_for_backward(hb, u, g, w1, w2)
        ctx.be, ctx.shape = be, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        hb, u, g, w1, w2 = ctx.saved_tensors
        be = ctx.be
        dyb = _grad_bf16(be, dy, dy.numel() // w2.shape[0])
        w2t = ops.transpose_cast(w2.detach().contiguous(), backend=be)       # [4C, C]
        du = ops.gemm_nt(dyb, w2t, act=ACT_DGELU, aux=u, backend=be)
        dw2 = _wgrad(be, dyb, g); db2 = ops.colsum_bf16(dyb, backend=be)
        w1t = ops.transpose_cast(w1.detach().contiguous(), backend=be)       # [C, 4C]
        dh = ops.gemm_nt(du, w1t, backend=be)
        dw1 = _wgrad(be, du, hb); db1 = ops.colsum_bf16(du, backend=be)
        return dh.view(*ctx.shape[:-1], w1.shape[1]), dw1, db1, dw2, db2, dy, None


class _WinAttn(torch.autograd.Function):
    """the attention step of timm's WindowAttention on bf16 qkv rows; rowidx (int32 [T], a permutation) = the tensor row of every (window, token): the cyclic shift and the
    window partition / reverse of the block as an index inside the kernels, so qkv and o stay in image order (None: rows already in (window, token) order)"""

    @staticmethod
    def forward(ctx, qkv, bias, mask, heads, be, rowidx=None):
        T, C3 = qkv.shape
        Cc = C3 // 3
        windows = T // N
        o = torch.empty((T, Cc), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((windows, heads, N), dtype=torch.float32, device=qkv.device)
        biasc = bias.detach().contiguous()
        scale = (Cc // heads) ** -0.5
        nW = 0 if mask is None else mask.shape[0]
        need = C.c_size_t(0)
        be.check(be.lib.vdk_window_attention_fwd_workspace_bytes(nW, heads, C.byref(need)), "vdk_window_attention_fwd_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=qkv.device)
        be.check(be.lib.vdk_window_attention_fwd(be.ptr(qkv), C3, be.ptr(o), Cc, be.ptr(lse), be.ptr(biasc), be.ptr(mask), nW, windows, heads, N, Cc // heads, scale,
                                                 be.ptr(rowidx), be.ptr(ws), ws.numel(), be.stream()), "vdk_window_attention_fwd")
        ctx.save_for_backward(qkv, o, lse, biasc)
        ctx.mask, ctx.heads, ctx.be, ctx.scale, ctx.rowidx = mask, heads, be, scale, rowidx
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, biasc = ctx.saved_tensors
        be, heads, mask = ctx.be, ctx.heads, ctx.mask
        T, C3 = qkv.shape
        Cc = C3 // 3
        windows = T // N
        do = _bf(be, do.contiguous())
        dqkv = torch.empty_like(qkv)
        dbias = torch.empty((heads, N, N), dtype=torch.float32, device=qkv.device)
        need = C.c_size_t(0)
        be.check(be.lib.vdk_window_attention_bwd_workspace_bytes(windows, 0 if mask is None else mask.shape[0], heads, C.byref(need)),
                 "vdk_window_attention_bwd_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=qkv.device)
        be.check(be.lib.vdk_window_attention_bwd(be.ptr(qkv), C3, be.ptr(o), be.ptr(do), Cc, be.ptr(lse), be.ptr(biasc), be.ptr(mask), 0 if mask is None else mask.shape[0],
                                                 windows, heads, N, Cc // heads, ctx.scale, be.ptr(ctx.rowidx), be.ptr(dqkv), C3, be.ptr(dbias), be.ptr(ws), ws.numel(),
                                                 be.stream()), "vdk_window_attention_bwd")
        return dqkv, dbias, None, None, None, None


class _BiasGather(torch.autograd.Function):
    """relative_position_bias_table [169, H] -> bias [H, 49, 49] (an index gather); the backward sums each table entry's <= 49 uses in a fixed order with one gather kernel
    (timm's index backward is an index_add with atomics)"""

    @staticmethod
    def forward(ctx, table, index, uses, be):
        ctx.be, ctx.uses, ctx.shape = be, uses, table.shape
        return table.detach()[index.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()

    @staticmethod
    def backward(ctx, dbias):
        be, uses = ctx.be, ctx.uses
        R, H = ctx.shape
        dbias = dbias.contiguous()
        dtable = torch.empty((R, H), dtype=torch.float32, device=dbias.device)
        be.check(be.lib.vdk_relpos_bias_table_grad(be.ptr(dbias), be.ptr(uses), R, uses.shape[1], H, N * N, be.ptr(dtable), be.stream()), "vdk_relpos_bias_table_grad")
        return dtable, None, None, None


class _Pool(torch.autograd.Function):
    """global average pool over the tokens of f32 NHWC rows"""

    @staticmethod
    def forward(ctx, x, be):
        B, H, W, Cc = x.shape
        out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        be.check(be.lib.vdk_avgpool_rows_f32_fwd(be.ptr(x.contiguous()), be.ptr(out), B, H * W, Cc, be.stream()), "vdk_avgpool_rows_f32_fwd")
        ctx.be, ctx.shape = be, x.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        ctx.be.check(ctx.be.lib.vdk_avgpool_rows_f32_bwd(ctx.be.ptr(dy.contiguous()), ctx.be.ptr(dx), None, B, H * W, Cc, ctx.be.stream()), "vdk_avgpool_rows_f32_bwd")
        return dx, None


# ---- modules with timm's names -------------------------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    pass


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, be, dev):
        super().__init__()
        self.heads, self.be = heads, be
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WS - 1) ** 2, heads, device=dev))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        idx = _rel_index(WS)
        self.register_buffer("relative_position_index", idx.to(dev), persistent=False)
        uses = torch.full(((2 * WS - 1) ** 2, N), -1, dtype=torch.int32)            # for the gather's backward: where each table entry is used (-1 padded)
        flat = idx.view(-1)
        for r in range((2 * WS - 1) ** 2):
            pos = (flat == r).nonzero().view(-1)
            uses[r, :pos.numel()] = pos.to(torch.int32)
        self.register_buffer("_uses", uses.to(dev), persistent=False)
        self.qkv = nn.Linear(dim, 3 * dim, device=dev)
        self.proj = nn.Linear(dim, dim, device=dev)


class Mlp(nn.Module):
    def __init__(self, dim: int, dev):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim, device=dev)
        self.fc2 = nn.Linear(4 * dim, dim, device=dev)


class SwinBlock(nn.Module):
    def __init__(self, dim: int, res: int, heads: int, shift: int, eps: float, be, dev):
        super().__init__()
        if res < WS or res % WS:
            raise ValueError("feature maps must be multiples of the 7 x 7 window (img_size % 224 == 0 for the patch4_window7 family)")
        if res == WS:
            shift = 0
        self.res, self.shift, self.eps, self.be, self.heads = res, shift, eps, be, heads
        self.norm1 = nn.LayerNorm(dim, eps=eps, device=dev)
        self.attn = WindowAttention(dim, heads, be, dev)
        self.norm2 = nn.LayerNorm(dim, eps=eps, device=dev)
        self.mlp = Mlp(dim, dev)
        self.register_buffer("attn_mask", _shift_mask(res, res, WS, shift).to(dev) if shift else None, persistent=False)
        self._perm = {}

    def _perms(self, B: int, dev):
        """the image-order row of every (window, token) of the (cyclically shifted) window partition of a batch of B maps"""
        if B not in self._perm:
            idx = torch.arange(B * self.res * self.res).view(B, self.res, self.res, 1)
            if self.shift:
                idx = torch.roll(idx, (-self.shift, -self.shift), (1, 2))
            self._perm = {B: _partition(idx, WS).view(-1).to(torch.int32).to(dev)}
        return self._perm[B]

    def forward(self, x):                                       # f32 [B, H, W, C]
        B, H, W, Cc = x.shape
        return self.rows(x.reshape(-1, Cc), B).view(B, H, W, Cc)

    def rows(self, x, B: int):
        """the block on the f32 token rows [B * H * W, C] in image order.  A stage keeps this 2-D form from block to block: no view nodes between the blocks' autograd nodes,
        so a gradient reaches its consumer as the very tensor its producer made (with its bf16 copy attached, _LNRes)"""
        Cc = x.shape[1]
        be, a = self.be, self.attn
        h, xs = _LNRes.apply(x, self.norm1.weight, self.norm1.bias, self.eps, be)                        # xs = x: the shortcut, routed through the norm's node
        qkv = _Lin.apply(h, a.qkv.weight, a.qkv.bias, None, torch.bfloat16, be)                          # rows stay in image order: the kernels follow the window index
        bias = _BiasGather.apply(a.relative_position_bias_table, a.relative_position_index, a._uses, be)
        o = _WinAttn.apply(qkv, bias, self.attn_mask, self.heads, be, self._perms(B, x.device))
        y = _Lin.apply(o, a.proj.weight, a.proj.bias, xs, torch.float32, be)                             # shortcut added in the GEMM epilogue
        h2, ys = _LNRes.apply(y, self.norm2.weight, self.norm2.bias, self.eps, be)
        return _Mlp.apply(h2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, ys, be)


class PatchMerging(nn.Module):
    def __init__(self, dim: int, eps: float, be, dev):
        super().__init__()
        self.eps, self.be = eps, be
        self.norm = nn.LayerNorm(4 * dim, eps=eps, device=dev)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False, device=dev)

    def forward(self, x):
        B, H, W, Cc = x.shape
        x = x.reshape(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 4, 2, 5).flatten(3)
        h = _LN.apply(x, self.norm.weight, self.norm.bias, self.eps, torch.bfloat16, self.be)
        return _Lin.apply(h, self.reduction.weight, None, None, torch.float32, self.be)


class SwinStage(nn.Module):
    def __init__(self, dim_in: int, dim: int, res: int, depth: int, heads: int, downsample: bool, eps: float, be, dev):
        super().__init__()
        self.downsample = PatchMerging(dim_in, eps, be, dev) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, 0 if j % 2 == 0 else WS // 2, eps, be, dev) for j in range(depth)])

    def forward(self, x):
        x = self.downsample(x)
        B, H, W, Cc = x.shape
        t = x.reshape(-1, Cc)
        for blk in self.blocks:
            t = blk.rows(t, B)
        return t.view(B, H, W, Cc)


class SwinTransformerAutograd(nn.Module):
    """(first form, kept as the engine's cross-check) timm.create_model('swin_*_patch4_window7_224', num_classes=C) as autograd nodes over the kernels: forward(x [B, 3, 224,
    224]) -> logits [B, C]; num_classes = 0: forward_features, the normed NHWC map [B, 7, 7, C_last] (timm's global_pool='' feature output of this family is NHWC)"""

    def __init__(self, spec: SwinSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None):
        super().__init__()
        self.spec = spec
        self.be = backend or _lib.load()
        dev = device if device is not None else ("cuda" if self.be.device_only else "cpu")
        if spec.img_size % 224:
            raise ValueError("the patch4_window7 family needs img_size % 224 == 0 (56 x 56 ... 7 x 7 maps of 7 x 7 windows)")
        if seed is not None:
            torch.manual_seed(seed)
        self.num_classes = spec.num_classes
        self.patch_embed = _Holder()
        self.patch_embed.proj = nn.Conv2d(spec.in_chans, spec.embed_dim, 4, 4, device=dev)
        self.patch_embed.norm = nn.LayerNorm(spec.embed_dim, eps=spec.ln_eps, device=dev)
        res = spec.img_size // 4
        layers, dim_in = [], spec.embed_dim
        for i, (d, h) in enumerate(zip(spec.depths, spec.heads)):
            dim = spec.embed_dim * 2 ** i
            if dim // h != 32:
                raise ValueError("window attention kernels are built for head dim 32 (every timm swin_*_window7_224)")
            if i > 0:
                res //= 2
            layers.append(SwinStage(dim_in, dim, res, d, h, i > 0, spec.ln_eps, self.be, dev))
            dim_in = dim
        self.layers = nn.Sequential(*layers)
        self.norm = nn.LayerNorm(dim_in, eps=spec.ln_eps, device=dev)
        self.num_features = dim_in
        self.head = _Holder()
        self.head.fc = nn.Linear(dim_in, spec.num_classes, device=dev) if spec.num_classes > 0 else nn.Identity()
        self.reset_parameters()

    def reset_parameters(self):
        """the reference's override after timm's init (classify_model.py:70-81): N(0, 0.02) for every Conv2d / Linear weight, zeros for Linear bias"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.02)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        s, be = self.spec, self.be
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (s.in_chans, s.img_size, s.img_size):
            raise ValueError(f"expected float32 [B, {s.in_chans}, {s.img_size}, {s.img_size}], got {tuple(x.shape)} {x.dtype}")
        B = x.shape[0]
        g = s.img_size // 4
        # 4 x 4 / stride 4 convolution = a Linear over (c, ky, kx) patches (the patch operand is an index permutation of the image)
        pt = x.reshape(B, s.in_chans, g, 4, g, 4).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, s.in_chans * 16)
        pe = self.patch_embed
        y = _Lin.apply(pt, pe.proj.weight.view(s.embed_dim, -1), pe.proj.bias, None, torch.float32, be)
        y = _LN.apply(y.view(B, g, g, s.embed_dim), pe.norm.weight, pe.norm.bias, s.ln_eps, torch.float32, be)
        y = self.layers(y)
        return _LN.apply(y, self.norm.weight, self.norm.bias, s.ln_eps, torch.float32, be)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        f = self.forward_features(x)
        if self.num_classes == 0:
            return f
        from .vit import _LinearFn
        return _LinearFn.apply(_Pool.apply(f, self.be), self.head.fc.weight, self.head.fc.bias, self.be)


# =====================================================================================================================================================
# the native engine
class SwinEngine:
    """owns the flat HBM buffers (fp32 master params, bf16 operand copies, grads, workspace) and calls vdk_swin_forward / vdk_swin_backward; the interface of vit.VitEngine
    (forward -> padded logits, backward(dlogits bf16), refresh_weights, params / grads / wb16 / n_floats / cp), so the fused train steps take either"""

    fp8 = 0

    def __init__(self, spec: SwinSpec, device=None, backend: Optional[_lib.Backend] = None, operand: str = "bf16"):
        from . import _abi
        from .vit import OPERANDS
        assert operand in OPERANDS, operand
        # 16-bit format of the GEMM / window-attention operands and the saved activations: "bf16", or "fp16" = what the reference's `torch.autocast(device_type=...)`
        # (engine/procedure/train.py:118, no dtype => float16 on a GPU) computes in; vit.FusedTrainStep then runs the GradScaler protocol of train.py:203-215 around the step
        self.operand = operand
        self.spec = spec
        # Stochastic depth (timm DropPath).  timm builds swin_* with drop_path_rate = 0.1 (block k of n gets rate 0.1 * k / (n - 1)); in training every block's two branches are
        # multiplied per sample by mask / keep_prob before the shortcut is added.  forward(training=True) draws the factors [2 * blocks, B] on the device from torch's generator
        # and hands them to the engine (VdkSwinConfig.drop_path); the backward of that forward reads the same buffer.  0.0: no factors, the epilogues are untouched.
        self.drop_path_rate = 0.0
        self._dp: Optional[torch.Tensor] = None
        self.be = backend or _lib.load()
        self.device = torch.device(device if device is not None else ("cuda" if self.be.device_only else "cpu"))
        cfg = self._cfg(1)
        nf, nt, ntr = _abi.I64(0), _abi.I32(0), _abi.I64(0)
        self.be.check(self.be.lib.vdk_swin_param_count(C.byref(cfg), C.byref(nf), C.byref(nt), C.byref(ntr)), "vdk_swin_param_count")
        self.n_floats, self.n_tensors, self.n_transposed = nf.value, nt.value, ntr.value
        self.entries = []
        name = C.create_string_buffer(96)
        off, numel, ndim = _abi.I64(0), _abi.I64(0), _abi.I32(0)
        shape = (_abi.I64 * 4)()
        for i in range(self.n_tensors):
            self.be.check(self.be.lib.vdk_swin_param_info(C.byref(cfg), i, name, 96, C.byref(off), C.byref(numel), shape, C.byref(ndim)), "vdk_swin_param_info")
            self.entries.append((name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
        dev = self.device
        self.params = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.wb16 = torch.zeros(self.n_floats, dtype=self.op_dtype, device=dev)
        self.wt16 = torch.zeros(self.n_transposed, dtype=self.op_dtype, device=dev)
        self.cp = (spec.num_classes + 7) // 8 * 8
        nst = len(spec.depths)
        self.features = spec.embed_dim * 2 ** (nst - 1)
        self.map_rows = (spec.img_size // 4 // 2 ** (nst - 1)) ** 2      # rows of the final map per image (49 for the 4-stage family at 224)
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = -1
        self._out: Optional[torch.Tensor] = None
        self._weights_version = None

    @property
    def op_dtype(self) -> torch.dtype:
        from .vit import OPERANDS
        return OPERANDS[self.operand]

    def set_operand(self, operand: str) -> None:
        """switch between bf16 and fp16 operands (the fp32 master weights stay; the 16-bit copies are rebuilt on the next forward)"""
        from .vit import OPERANDS
        assert operand in OPERANDS, operand
        if operand == self.operand:
            return
        self.operand = operand
        self.wb16 = torch.zeros(self.n_floats, dtype=self.op_dtype, device=self.device)
        self.wt16 = torch.zeros(self.n_transposed, dtype=self.op_dtype, device=self.device)
        self._weights_version = None

    def __deepcopy__(self, memo):
        import copy
        new = object.__new__(SwinEngine)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, v if k == "be" else copy.deepcopy(v, memo))
        return new

    def _cfg(self, batch: int, pass_dp: bool = False):
        from . import _abi
        s = self.spec
        I4 = _abi.I32 * 4
        pad = lambda t: I4(*(tuple(t) + (0,) * (4 - len(t))))      # shallower members of the family (tests): trailing zeros
        if pass_dp and self._dp is not None and self._dp.shape[1] != batch:
            raise RuntimeError(f"stochastic-depth factors of a forward at batch {self._dp.shape[1]} under a call at batch {batch}: a backward must follow ITS forward "
                               "(the engine keeps one forward's workspace and factors)")
        dp = self._dp if pass_dp else None      # (only forward / backward read the factors: layout, workspace and weight-refresh calls pass none)
        return _abi.SwinConfig(batch, s.img_size, s.in_chans, s.embed_dim, pad(s.depths), pad(s.heads), s.num_classes, s.ln_eps, _abi.F16_ if self.operand == "fp16" else _abi.BF16,
                               self.be.ptr(dp))

    def draw_drop_path(self, batch: int) -> Optional[torch.Tensor]:
        """factors f32 [2 * blocks, batch] = Bernoulli(keep_prob) / keep_prob per (branch, sample) (timm's drop_path(): `x.new_empty(shape).bernoulli_(keep_prob).div_(keep_prob)`);
        block k's keep_prob = 1 - drop_path_rate * k / (blocks - 1), the same for its two branches.  None when drop_path_rate == 0."""
        if self.drop_path_rate <= 0.0:
            return None
        n = sum(self.spec.depths)
        if getattr(self, "_dp_const", None) is None or self._dp_const[0] != self.drop_path_rate:      # the constants once: three launches per draw (rand, <, where)
            rates = torch.linspace(0, self.drop_path_rate, n, device=self.device).repeat_interleave(2)  # timm: [x.tolist() for x in torch.linspace(0, drop_path_rate, sum(depths)).split(depths)]
            keep = (1.0 - rates).unsqueeze(1)
            self._dp_const = (self.drop_path_rate, keep, 1.0 / keep, torch.zeros((), device=self.device))
        _, keep, inv_keep, zero = self._dp_const
        u = torch.rand((2 * n, batch), device=self.device)
        return torch.where(u < keep, inv_keep, zero).contiguous()

    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch != batch:
            need = C.c_size_t(0)
            cfg = self._cfg(batch)
            self.be.check(self.be.lib.vdk_swin_workspace_bytes(C.byref(cfg), C.byref(need)), "vdk_swin_workspace_bytes")
            if self._ws is None or self._ws.numel() < need.value:
                self._ws = None
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
            shape = (batch, self.cp) if self.cp else (batch * self.map_rows, self.features)
            self._out = torch.empty(shape, dtype=torch.float32, device=self.device)
        return self._ws

    def refresh_weights(self, skip_wb16: bool = False) -> None:
        cfg = self._cfg(1)
        be = self.be
        be.check(be.lib.vdk_swin_refresh_weights(C.byref(cfg), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wt16), int(skip_wb16), be.stream()), "vdk_swin_refresh_weights")
        self._weights_version = self.params._version

    def _ensure_fresh(self) -> None:
        if self._weights_version != self.params._version:
            self.refresh_weights()

    def fp8_update(self) -> None:      # (protocol of the fused train step: this family has no fp8 mode)
        pass

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """the tensor `name` (timm key) inside a flat buffer laid out like params (grads, a momentum or EMA buffer of the fused step)"""
        for n, off, numel, shape in self.entries:
            if n == name:
                return flat[off:off + numel].view(shape)
        raise KeyError(name)

    def forward(self, x: torch.Tensor, training: bool = True, drop_path: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x f32 [B, C, H, W] -> logits f32 [B, Cp] (padded columns beyond num_classes) or, in feature mode, the normed map rows f32 [B * 49, 8 E].
        training: stochastic depth is active (drop_path_rate > 0): factors are drawn per call, or taken from `drop_path` ([2 * blocks, B], tests hand the oracle the same ones)."""
        s = self.spec
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (s.in_chans, s.img_size, s.img_size):
            raise ValueError(f"expected float32 [B, {s.in_chans}, {s.img_size}, {s.img_size}], got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B = x.shape[0]
        ws = self._workspace(B)
        self._ensure_fresh()
        self._dp = None
        self._fwd_serial = getattr(self, "_fwd_serial", 0) + 1      # (the workspace and the factors below belong to THIS forward: _SwinFunction.backward checks it is still the last one)
        if training:
            self._dp = drop_path.to(self.device, torch.float32).contiguous() if drop_path is not None else self.draw_drop_path(B)
            assert self._dp is None or tuple(self._dp.shape) == (2 * sum(s.depths), B)
        cfg = self._cfg(B, pass_dp=True)
        be = self.be
        be.check(be.lib.vdk_swin_forward(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.wb16), be.ptr(ws), ws.numel(), be.ptr(self._out), be.stream()), "vdk_swin_forward")
        return self._out

    def backward(self, dout: torch.Tensor, on_ready=None) -> torch.Tensor:
        """dlogits in the operand format [B, Cp] (feature mode: f32 [B * 49, 8 E]) -> self.grads (flat fp32, overwritten); needs the workspace of the matching forward"""
        from . import _abi
        B = dout.shape[0] if self.cp else dout.shape[0] // self.map_rows
        assert B == self._ws_batch and dout.is_contiguous()
        assert (dout.dtype == self.op_dtype and dout.shape[1] == self.cp) if self.cp else (dout.dtype == torch.float32 and dout.shape[1] == self.features)
        cfg = self._cfg(B, pass_dp=True)
        be = self.be
        cb = _abi.GRAD_READY_FN(lambda user, off, n: on_ready(off, n)) if on_ready is not None else _abi.GRAD_READY_FN(0)
        be.check(be.lib.vdk_swin_backward(C.byref(cfg), be.ptr(dout), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wt16), be.ptr(self._ws), self._ws.numel(),
                                          be.ptr(self.grads), cb, None, be.stream()), "vdk_swin_backward")
        return self.grads


class _SwinFunction(torch.autograd.Function):
    """model(x) as ONE autograd node: forward = vdk_swin_forward, backward = vdk_swin_backward"""

    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module.engine
        module._sync_flat()
        out = eng.forward(x, training=module.training)
        ctx.module = module
        ctx.serial = eng._fwd_serial
        if eng.cp == 0:
            r = int(round(eng.map_rows ** 0.5))
            return out.view(x.shape[0], r, r, eng.features).clone()
        return out[:, :eng.spec.num_classes].clone()

    @staticmethod
    def backward(ctx, dout):
        module = ctx.module
        eng = module.engine
        be = eng.be
        if eng._fwd_serial != ctx.serial:
            raise RuntimeError("another forward ran through this Swin engine since the one being differentiated: its saved activations and stochastic-depth factors are gone "
                               "(one engine keeps ONE forward's workspace; run backward before the next forward)")
        if eng.cp == 0:
            g = eng.backward(dout.contiguous().view(-1, eng.features))
        else:
            B, Cn = dout.shape
            stage = torch.zeros((B, eng.cp), dtype=torch.float32, device=dout.device)
            stage[:, :Cn].copy_(dout)
            dl = torch.empty((B, eng.cp), dtype=eng.op_dtype, device=dout.device)
            cast = be.lib.vdk_cast_f32_f16 if eng.operand == "fp16" else be.lib.vdk_cast_f32_bf16
            be.check(cast(be.ptr(stage), be.ptr(dl), stage.numel(), be.stream()), "vdk_cast_f32_16")
            g = eng.backward(dl)
        grads = tuple(g[off:off + numel].view(shape) for (_, off, numel, shape) in eng.entries)
        return (None, None) + grads


class SwinTransformer(nn.Module):
    """Drop-in for the object `timm.create_model('swin_*_patch4_window7_224', pretrained=False, num_classes=C)` hands the reference, on the native engine.  The module tree
    mirrors timm's (patch_embed.proj / norm, layers.i.downsample.{norm, reduction}, layers.i.blocks.j.{norm1, attn.{relative_position_bias_table, qkv, proj}, norm2,
    mlp.{fc1, fc2}}, norm, head.fc) with parameter-only holders, so named_parameters() / state_dict() / load_state_dict() carry timm's key names; every Parameter is a view
    into the engine's flat fp32 buffer.  forward(x [B, 3, 224, 224]) -> logits [B, C]; num_classes = 0: the normed NHWC map [B, 7, 7, C_last] (timm's forward_features)."""

    def __init__(self, spec: SwinSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None, operand: str = "bf16", drop_path_rate: float = 0.0):
        super().__init__()
        if not 0.0 <= drop_path_rate < 1.0:
            raise ValueError("0 <= drop_path_rate < 1")
        self.spec = spec
        self.engine = SwinEngine(spec, device=device, backend=backend, operand=operand)
        self.engine.drop_path_rate = float(drop_path_rate)      # stochastic depth in train() mode (SwinEngine.draw_drop_path); create_model's default is timm's 0.1
        self.be = self.engine.be
        self.num_classes = spec.num_classes
        self.num_features = self.engine.features
        self._plist = []
        for name, off, numel, shape in self.engine.entries:
            p = nn.Parameter(self.engine.params[off:off + numel].view(shape))
            parts = name.split(".")
            m = self
            for part in parts[:-1]:
                if part not in m._modules:
                    m.add_module(part, _Holder())
                m = m._modules[part]
            m.register_parameter(parts[-1], p)
            self._plist.append((name, p))
        self.reset_parameters(seed)

    def reset_parameters(self, seed: Optional[int] = None) -> None:
        """timm's defaults + the reference's override after them (classify_model.py:70-81): N(0, 0.02) for every Conv2d / Linear weight, zeros for Linear bias (the
        convolution's bias keeps its default), LayerNorm (1, 0), relative_position_bias_table trunc_normal(0.02)"""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        s = self.spec
        with torch.no_grad():
            for name, p in self._plist:
                if name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight"):
                    v = torch.ones(p.shape)
                elif name.endswith("relative_position_bias_table"):
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen).clamp_(-2.0, 2.0)
                elif name == "patch_embed.proj.bias":
                    bound = 1.0 / (s.in_chans * 16) ** 0.5
                    v = (torch.rand(p.shape, generator=gen) * 2 - 1) * bound
                elif name.endswith(".bias"):
                    v = torch.zeros(p.shape)
                else:
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen)
                p.copy_(v.to(p.device))

    def _sync_flat(self) -> None:
        eng = self.engine
        base = eng.params.data_ptr()
        for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
            if p.data_ptr() != base + off * 4:
                with torch.no_grad():
                    eng.params[off:off + numel].view(shape).copy_(p.detach().to(eng.device))
                    p.data = eng.params[off:off + numel].view(shape)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self.engine.device))
        if probe.device != self.engine.device or probe.dtype != torch.float32:
            if probe.dtype != torch.float32:
                raise RuntimeError("visiondk_amd Swin keeps fp32 master weights; bf16 copies are internal")
            if self.engine.be.device_only and probe.device.type != "cuda":
                raise RuntimeError("visiondk_amd Swin lives on the GPU (no CPU fallback)")
            eng = self.engine
            eng.device = probe.device
            for attr in ("params", "grads", "wb16", "wt16"):
                setattr(eng, attr, getattr(eng, attr).to(probe.device))
            eng._ws, eng._ws_batch, eng._weights_version = None, -1, None
            for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
                p.data = eng.params[off:off + numel].view(shape)
        return self

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        if self.num_classes != 0:
            raise RuntimeError("forward_features: build the model with num_classes=0 (the feature model TimmWrapper asks for)")
        return self.forward(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _SwinFunction.apply(x, self, *[p for _, p in self._plist])


def create_model(name: str, pretrained: bool = False, num_classes: int = 1000, img_size: int = 224, device=None, backend=None, seed: Optional[int] = None,
                 native: bool = True, operand: str = "bf16", drop_path_rate: float = 0.1, global_pool: Optional[str] = None, **kw):
    """timm.create_model(name, pretrained=..., num_classes=...) for the swin_*_patch4_window7_224 family (models/classifier/classify_model.py:49-54); native=False: the
    autograd-node form of round 3 (the engine's cross-check).  drop_path_rate: timm's own default for this family is 0.1 (stochastic depth in train mode); the yaml's
    `kwargs` reach it as they reach timm.  Keyword arguments timm would act on and this engine does not build are refused, not swallowed."""
    if kw:
        raise TypeError(f"swin.create_model: unsupported keyword arguments {sorted(kw)} (built: drop_path_rate, img_size, operand)")
    if global_pool not in (None, "", "avg"):
        raise NotImplementedError("global_pool: timm's default average pool (classifier) or '' (feature map)")
    name = name[5:] if name.startswith("timm-") else name
    if name not in TIMM_SWINS:
        raise KeyError(f"unknown Swin id {name!r}: {sorted(TIMM_SWINS)}")
    if pretrained:
        raise RuntimeError("there is no network here: load a checkpoint with load_state_dict (timm names)")
    spec = SwinSpec(img_size=img_size, num_classes=num_classes, **TIMM_SWINS[name])
    if native:
        return SwinTransformer(spec, device=device, backend=backend, seed=seed, operand=operand, drop_path_rate=drop_path_rate)
    if drop_path_rate not in (0.0, 0.1):      # (0.1 = the default nobody asked for: the cross-check form never drops)
        raise NotImplementedError("stochastic depth is built into the native engine (native=True)")
    if operand != "bf16":
        raise NotImplementedError("the autograd-node form runs on bf16 operands; fp16 is the native engine's (native=True)")
    return SwinTransformerAutograd(spec, device=device, backend=backend, seed=seed)
