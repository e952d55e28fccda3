"""Margin-softmax heads of the faceX / CBIR training path on the HIP kernels (csrc/margin_head.hip + the MFMA GEMMs).

Mirrors models/faceX/head/{arcface,circleloss,mv_softmax}.py and `HeadFactory` (head_def.py:14-56): same constructor
arguments, same `weight` Parameter ([feat_dim, num_class], unit-norm columns init, arcface.py:11-12), `forward(feats, labels)
-> logits`.  `margin_ce()` is the fused form (head + CrossEntropy, no B x C logits through torch)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _abi, _lib, ops


def _up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


class _HeadState:
    """buffers of one forward (kept for the backward)"""
    __slots__ = ("cos", "fh", "fb", "fbt", "finv", "winv", "wb", "B", "Bp", "C", "Cp", "D")


def _forward_cos(be, feats: torch.Tensor, weight: torch.Tensor, planes: int = 3, with_cos: bool = True, dtype=torch.bfloat16) -> _HeadState:
    """planes = 3: cos from split-bf16 planes (hi*hi + lo*hi + hi*lo: fp32-class, what the reference's CPU path computes); planes = 1: one bf16 plane per operand --
    exactly the reference's GPU path, where train.py:118 runs the head under autocast and torch.mm(feats, kernel_norm) rounds both operands to bf16.
    dtype: format of the planes (torch.bfloat16 | torch.float16: the head of an fp16 train step, vdk_colnorm_fwd_dt)"""
    st = _HeadState()
    dtc = _abi.F16_ if dtype == torch.float16 else _abi.BF16
    B, D = feats.shape
    Cn = weight.shape[1]
    st.B, st.D, st.C, st.Bp, st.Cp = B, D, Cn, _up(B, 64), _up(Cn, 8)
    dev = feats.device
    st.winv = torch.empty(Cn, dtype=torch.float32, device=dev)
    st.wb = torch.empty((planes * D, st.Cp), dtype=dtype, device=dev)   # split planes (hi, hi, lo) along K
    be.check(be.lib.vdk_colnorm_fwd_dt(be.ptr(weight), Cn, D, Cn, st.Cp, 1e-12, be.ptr(st.winv), be.ptr(st.wb), st.Cp, planes, dtc, be.stream()), "vdk_colnorm_fwd")
    st.fh = torch.empty((B, D), dtype=torch.float32, device=dev)
    st.fb = torch.empty((st.Bp, D), dtype=dtype, device=dev)
    st.fbt = torch.empty((planes * D, st.Bp), dtype=dtype, device=dev)  # split planes (hi, lo, hi)
    st.finv = torch.empty(B, dtype=torch.float32, device=dev)
    be.check(be.lib.vdk_rownorm_fwd_dt(be.ptr(feats), B, st.Bp, D, 1e-12, be.ptr(st.fh), be.ptr(st.fb), be.ptr(st.fbt), be.ptr(st.finv), planes, dtc, be.stream()),
             "vdk_rownorm_fwd")
    # cos[Bp, Cp] = f^ . W^ : TN kernel over K = 3D split planes (hi*hi + lo*hi + hi*lo), A = fbt [3D, Bp], B = wb [3D, Cp]
    st.cos = ops.gemm_nt(st.fbt, st.wb, out_dtype=torch.float32, trans=True, backend=be) if with_cos else None
    return st


def _backward_from_dcos(be, st: _HeadState, weight: torch.Tensor, dcos: torch.Tensor):
    """dcos bf16 [Bp, Cp] -> (dfeats f32 [B, D], dweight f32 [D, C])"""
    dev = weight.device
    # dW^ [D, Cp] = f^T dcos : TN, A = f^ [Bp, D], B = dcos [Bp, Cp]
    dwh = ops.gemm_nt(st.fb, dcos, out_dtype=torch.float32, trans=True, backend=be)
    dW = torch.empty((st.D, st.C), dtype=torch.float32, device=dev)
    be.check(be.lib.vdk_colnorm_bwd(be.ptr(weight), st.C, be.ptr(st.winv), be.ptr(dwh), st.Cp, st.D, st.C, be.ptr(dW), st.C, be.stream()), "vdk_colnorm_bwd")
    # df^ [Bp, D] = dcos W^T : NT, A = dcos [Bp, Cp], B = W^ [D, Cp]; the contraction is the class dim -> split-K
    tiles = ((st.Bp + 255) // 256) * ((st.D + 255) // 256)
    splitk = max(1, min(64, 256 // tiles, st.Cp // 4096)) if st.Cp >= 8192 else 1
    dfh = ops.gemm_nt(dcos, st.wb[:st.D], out_dtype=torch.float32, splitk=splitk, backend=be)   # hi plane
    df = torch.empty((st.B, st.D), dtype=torch.float32, device=dev)
    be.check(be.lib.vdk_rownorm_bwd(be.ptr(st.fh), be.ptr(st.finv), be.ptr(dfh), st.D, st.B, st.D, be.ptr(df), be.stream()), "vdk_rownorm_bwd")
    return df, dW


class _HeadFn(torch.autograd.Function):
    """head(feats, labels) -> logits, differentiable (the reference form: the loss is applied by the caller)"""

    @staticmethod
    def forward(ctx, feats, weight, labels, head):
        be = head.be
        st = _forward_cos(be, feats.contiguous(), weight)
        logits = torch.empty((st.B, st.C), dtype=torch.float32, device=feats.device)
        cfg = head.cfg
        be.check(be.lib.vdk_margin_ce(C.byref(cfg), be.ptr(st.cos), st.Cp, st.B, st.C, be.ptr(labels), 0.0, 1.0, be.ptr(logits), st.C, None, None, 0,
                                      be.stream()), "vdk_margin_ce")
        ctx.st, ctx.head, ctx.labels = st, head, labels
        ctx.save_for_backward(weight)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        st, head = ctx.st, ctx.head
        be = head.be
        (weight,) = ctx.saved_tensors
        dcos = torch.zeros((st.Bp, st.Cp), dtype=torch.bfloat16, device=dlogits.device)
        dlogits = dlogits.contiguous()
        be.check(be.lib.vdk_margin_bwd(C.byref(head.cfg), be.ptr(st.cos), st.Cp, st.B, st.C, be.ptr(ctx.labels), be.ptr(dlogits), st.C, be.ptr(dcos), st.Cp,
                                       be.stream()), "vdk_margin_bwd")
        df, dW = _backward_from_dcos(be, st, weight, dcos)
        return df, dW, None, None


class _MarginHead(nn.Module):
    mode = -1

    def __init__(self, feat_dim: int, num_class: int, backend: Optional[_lib.Backend] = None, device=None):
        super().__init__()
        self.be = backend or _lib.load()
        dev = device if device is not None else ("cuda" if self.be.device_only else "cpu")
        w = torch.empty(feat_dim, num_class)
        w.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)     # arcface.py:12 — unit-norm columns
        self.weight = nn.Parameter(w.to(dev))
        self.cfg = _abi.MarginHead(self.mode, 1.0, 0.0, 0.0, 0.0)

    def forward(self, feats: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        return _HeadFn.apply(feats, self.weight, labels, self)

    def _margin_ce_f32(self, feats: torch.Tensor, labels: torch.Tensor, label_smoothing: float, grad_scale: Optional[float]):
        """margin_ce in fp32-class arithmetic throughout (FaceTrainStep(precision="fp32")): the cosines, d(loss)/d(cos) and both gradient products on the fp32 MFMA from
        fp32 operands -- what the reference's head computes in its no-autocast face / CBIR loop (engine/procedure/train.py:217-227)."""
        be = self.be
        w = self.weight.detach()
        st = _forward_cos(be, feats.contiguous(), w, 1, with_cos=False)      # the fp32 normalised features (fh), 1 / |f| and 1 / |w_c|; the bf16 planes are not used
        dev = feats.device
        wn = torch.zeros((st.D, st.Cp), dtype=torch.float32, device=dev)
        wn[:, :st.C] = w * st.winv[None, :]                                  # unit-norm columns (arcface.py:16 F.normalize(self.weight, dim=0))
        fh = st.fh
        cos = ops.gemm_f32(fh, wn, b_kmajor=True, backend=be)                # [B, Cp]
        loss = torch.empty(st.B, dtype=torch.float32, device=dev)
        dcos = torch.empty((st.B, st.Cp), dtype=torch.float32, device=dev)
        gs = 1.0 / st.B if grad_scale is None else grad_scale
        be.check(be.lib.vdk_margin_ce_f32(C.byref(self.cfg), be.ptr(cos), st.Cp, st.B, st.C, be.ptr(labels), label_smoothing, gs, be.ptr(loss), be.ptr(dcos), st.Cp,
                                          be.stream()), "vdk_margin_ce_f32")
        dwh = ops.gemm_f32(fh, dcos, a_kmajor=True, b_kmajor=True, backend=be)                       # f^T dcos: [D, Cp]
        dW = torch.empty((st.D, st.C), dtype=torch.float32, device=dev)
        be.check(be.lib.vdk_colnorm_bwd(be.ptr(w), st.C, be.ptr(st.winv), be.ptr(dwh), st.Cp, st.D, st.C, be.ptr(dW), st.C, be.stream()), "vdk_colnorm_bwd")
        tiles = ((st.B + 127) // 128) * ((st.D + 127) // 128)
        splits = max(1, min(64, 1024 // tiles, st.Cp // 2048))
        dfh = ops.gemm_f32(dcos, wn, k_splits=splits, backend=be)                                    # dcos W^T: [B, D], the contraction over the classes in slabs
        df = torch.empty((st.B, st.D), dtype=torch.float32, device=dev)
        be.check(be.lib.vdk_rownorm_bwd(be.ptr(st.fh), be.ptr(st.finv), be.ptr(dfh), st.D, st.B, st.D, be.ptr(df), be.stream()), "vdk_rownorm_bwd")
        return loss, df, dW

    def margin_ce(self, feats: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.0, grad_scale: Optional[float] = None, cos_planes: int = 3,
                  fused: Optional[bool] = None, precise: bool = False, operand: str = "bf16", loss_scale: Optional[torch.Tensor] = None):
        """Fused head + CrossEntropy (mean): returns (loss_rows [B], dfeats [B, D], dweight [D, C]); no autograd, no B x C logits.
        fused=True: the epilogue-fused form (SURVEY K11 to the letter: cos never exists as an fp32 [B, C] tensor; the cos GEMM runs twice with the head applied to its tiles in
        registers).  Measured on the MI355X at B 512, C 10^6: 6.7 ms against 5.8 ms for the default form that writes cos once in fp32 -- the head's ~18 lane-ops per logit run in a
        GEMM epilogue that nothing overlaps (one workgroup per CU), whereas the row kernel runs them at full occupancy -- so it is opt-in (profiles/r02_margin_head.json).
        cos_planes = 1: the cosines from single bf16 operands, as the reference's autocast path computes them (see _forward_cos)
        operand = "fp16": fp16 planes and an fp16 d(loss)/d(cos) -- 8x less operand rounding in the two gradient products than bf16; loss_scale (a device scalar, GradScaler's
        scale: engine/procedure/train.py:205) is multiplied into d(loss)/d(cos), so dfeats and dweight come back SCALED and the caller un-scales in its optimizer pass
        (FaceTrainStep over an fp16 backbone).  Without it the softmax gradient of a wide head underflows fp16."""
        be = self.be
        if precise:
            if loss_scale is not None:
                raise ValueError("precise=True computes unscaled fp32 gradients")
            return self._margin_ce_f32(feats, labels, label_smoothing, grad_scale)
        if operand not in ("bf16", "fp16"):
            raise ValueError("operand must be 'bf16' or 'fp16'")
        dt16 = torch.float16 if operand == "fp16" else torch.bfloat16
        if operand == "fp16" or loss_scale is not None:
            if fused:
                raise NotImplementedError("the epilogue-fused form is built for bf16 operands without a loss scale")
            fused = False
        gs_ = None
        if fused:
            # the fused form: the cos GEMM runs twice with the head applied to its tiles in registers (statistics, then the gradient); cos never exists as an fp32 [B, C] tensor
            st = _forward_cos(be, feats.contiguous(), self.weight.detach(), cos_planes, with_cos=False)
            dev = feats.device
            K = st.fbt.shape[0]
            nslice = (st.Cp + 63) // 64
            stats = torch.empty((st.B, nslice, 4), dtype=torch.float32, device=dev)
            tlogit = torch.zeros(st.B, dtype=torch.float32, device=dev)
            gt = None
            if self.mode in (_abi.HEAD_MV_AM, _abi.HEAD_MV_ARC):      # MV-Softmax: every column's margin depends on the row's target cosine
                gt = torch.empty(st.B, dtype=torch.float32, device=dev)
                be.check(be.lib.vdk_margin_target_cos_direct(be.ptr(st.fbt), st.Bp, be.ptr(st.wb), st.Cp, K, st.B, be.ptr(labels), be.ptr(gt), be.stream()),
                         "vdk_margin_target_cos_direct")
            gs_ = 1.0 / st.B if grad_scale is None else grad_scale
            rc = be.lib.vdk_margin_cos_pass(C.byref(self.cfg), 1, be.ptr(st.fbt), st.Bp, be.ptr(st.wb), st.Cp, st.B, st.Bp, st.C, st.Cp, K, be.ptr(labels), be.ptr(gt), be.ptr(stats),
                                            be.ptr(tlogit), None, label_smoothing, gs_, None, 0, be.stream())
            if rc == 0:
                loss = torch.empty(st.B, dtype=torch.float32, device=dev)
                rowstat = torch.empty((st.B, 2), dtype=torch.float32, device=dev)
                be.check(be.lib.vdk_margin_rowstat(be.ptr(stats), nslice, be.ptr(tlogit), st.B, st.C, label_smoothing, be.ptr(rowstat), be.ptr(loss), be.stream()), "vdk_margin_rowstat")
                dcos = torch.empty((st.Bp, st.Cp), dtype=torch.bfloat16, device=dev)
                be.check(be.lib.vdk_margin_cos_pass(C.byref(self.cfg), 2, be.ptr(st.fbt), st.Bp, be.ptr(st.wb), st.Cp, st.B, st.Bp, st.C, st.Cp, K, be.ptr(labels), be.ptr(gt), None, None,
                                                    be.ptr(rowstat), label_smoothing, gs_, be.ptr(dcos), st.Cp, be.stream()), "vdk_margin_cos_pass")
                df, dW = _backward_from_dcos(be, st, self.weight.detach(), dcos)
                return loss, df, dW
            if rc != _abi.EUNSUPPORTED:      # every other error is an error; VDK_EUNSUPPORTED = the 256x256 TN kernel does not serve this shape -> the materialised form below
                be.check(rc, "vdk_margin_cos_pass")
            st.cos = ops.gemm_nt(st.fbt, st.wb, out_dtype=torch.float32, trans=True, backend=be)      # shapes the 256x256 TN kernel does not serve: the materialised form
        else:
            st = _forward_cos(be, feats.contiguous(), self.weight.detach(), cos_planes, dtype=dt16)
        loss = torch.empty(st.B, dtype=torch.float32, device=feats.device)
        dcos = torch.empty((st.Bp, st.Cp), dtype=dt16, device=feats.device)    # rows < B are written whole (padding columns zeroed) by the kernel
        if st.Bp > st.B:
            dcos[st.B:].zero_()
        gs = 1.0 / st.B if grad_scale is None else grad_scale
        be.check(be.lib.vdk_margin_ce_amp(C.byref(self.cfg), be.ptr(st.cos), st.Cp, st.B, st.C, be.ptr(labels), label_smoothing, gs, be.ptr(loss_scale), None, 0, be.ptr(loss),
                                          be.ptr(dcos), st.Cp, _abi.F16_ if operand == "fp16" else _abi.BF16, be.stream()), "vdk_margin_ce")
        df, dW = _backward_from_dcos(be, st, self.weight.detach(), dcos)
        return loss, df, dW


class ArcFace(_MarginHead):
    """models/faceX/head/arcface.py"""
    mode = _abi.HEAD_ARCFACE

    def __init__(self, feat_dim, num_class, margin_arc=0.35, margin_am=0.0, scale=32, **kw):
        super().__init__(feat_dim, num_class, **kw)
        self.cfg = _abi.MarginHead(self.mode, float(scale), float(margin_arc), float(margin_am), 0.0)


class CircleLoss(_MarginHead):
    """models/faceX/head/circleloss.py"""
    mode = _abi.HEAD_CIRCLE

    def __init__(self, feat_dim, num_class, margin=0.25, gamma=256, **kw):
        super().__init__(feat_dim, num_class, **kw)
        self.cfg = _abi.MarginHead(self.mode, float(gamma), float(margin), 0.0, 0.0)


class MV_Softmax(_MarginHead):
    """models/faceX/head/mv_softmax.py"""

    def __init__(self, feat_dim, num_class, is_am, margin=0.35, mv_weight=1.12, scale=32, **kw):
        self.mode = _abi.HEAD_MV_AM if is_am else _abi.HEAD_MV_ARC
        super().__init__(feat_dim, num_class, **kw)
        self.cfg = _abi.MarginHead(self.mode, float(scale), float(margin), 0.0, float(mv_weight))


class _MagFn(torch.autograd.Function):
    """ArcFace logits with a per-row margin tensor that itself carries gradient (MagFace): d logit[b, y_b] / d m_b = -s sin(theta + m_b) where the margin branch is active"""

    @staticmethod
    def forward(ctx, feats, weight, labels, margins, head):
        be = head.be
        st = _forward_cos(be, feats.contiguous(), weight)
        logits = torch.empty((st.B, st.C), dtype=torch.float32, device=feats.device)
        margins = margins.detach().contiguous().float()
        cfg = _abi.MarginHead(_abi.HEAD_ARCFACE, head.cfg.scale, 0.0, head.cfg.margin_am, 0.0, be.ptr(margins))
        be.check(be.lib.vdk_margin_ce(C.byref(cfg), be.ptr(st.cos), st.Cp, st.B, st.C, be.ptr(labels), 0.0, 1.0, be.ptr(logits), st.C, None, None, 0,
                                      be.stream()), "vdk_margin_ce")
        ctx.st, ctx.head, ctx.labels, ctx.margins, ctx.cfg = st, head, labels, margins, cfg
        ctx.save_for_backward(weight)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        st, head, m = ctx.st, ctx.head, ctx.margins
        be = head.be
        (weight,) = ctx.saved_tensors
        dcos = torch.zeros((st.Bp, st.Cp), dtype=torch.bfloat16, device=dlogits.device)
        dlogits = dlogits.contiguous()
        be.check(be.lib.vdk_margin_bwd(C.byref(ctx.cfg), be.ptr(st.cos), st.Cp, st.B, st.C, be.ptr(ctx.labels), be.ptr(dlogits), st.C, be.ptr(dcos), st.Cp,
                                       be.stream()), "vdk_margin_bwd")
        df, dW = _backward_from_dcos(be, st, weight, dcos)
        # the margin's own gradient: B target entries, host-side tensor arithmetic on [B] vectors
        rows = torch.arange(st.B, device=dlogits.device)
        c = st.cos[:st.B].gather(1, ctx.labels.view(-1, 1)).squeeze(1).clamp(-1.0, 1.0)
        active = c > torch.cos(math.pi - m)
        dm = dlogits[rows, ctx.labels] * (-head.cfg.scale) * (torch.sqrt(1.0 - c * c) * torch.cos(m) + c * torch.sin(m)) * active
        return df, dW, None, dm, None


class MagFace(_MarginHead):
    """models/faceX/head/magface.py: ArcFace with the margin growing with the feature magnitude, `forward -> (logits, lamda * loss_g)` exactly like the reference
    module (a tuple the reference's Trainer cannot consume, train.py:196: dead code upstream, kept for head_def.py:26-36's sake)."""
    mode = _abi.HEAD_ARCFACE

    def __init__(self, feat_dim, num_class, margin_am=0.0, scale=32, l_a=10, u_a=110, l_margin=0.45, u_margin=0.8, lamda=20, **kw):
        super().__init__(feat_dim, num_class, **kw)
        self.cfg = _abi.MarginHead(self.mode, float(scale), 0.0, float(margin_am), 0.0)
        self.l_a, self.u_a, self.l_margin, self.u_margin, self.lamda = l_a, u_a, l_margin, u_margin, lamda

    def calc_margin(self, x):
        return (self.u_margin - self.l_margin) / (self.u_a - self.l_a) * (x - self.l_a) + self.l_margin

    def forward(self, feats: torch.Tensor, labels: torch.Tensor):
        x_norm = torch.norm(feats, dim=1, keepdim=True).clamp(self.l_a, self.u_a)
        ada_margin = self.calc_margin(x_norm)
        loss_g = 1 / (self.u_a ** 2) * x_norm + 1 / x_norm
        logits = _MagFn.apply(feats, self.weight, labels, ada_margin.squeeze(1), self)
        return logits, self.lamda * loss_g

    def margin_ce(self, *a, **k):
        raise NotImplementedError("MagFace returns (logits, regulariser) like the reference module; use forward()")


class HeadFactory:
    """models/faceX/head/head_def.py:14-56 — head_type in {'arcface', 'circleloss' (yaml alias 'circle'), 'mv-softmax'}."""

    def __init__(self, head_type: str, head_conf: dict, backend=None, device=None):
        self.head_type, self.head_param, self.kw = head_type, head_conf, dict(backend=backend, device=device)

    def get_head(self):
        p = self.head_param
        t = self.head_type.lower()
        if t == "arcface":
            return ArcFace(p["feat_dim"], p["num_class"], p.get("margin_arc", 0.35), p.get("margin_am", 0.0), p.get("scale", 32), **self.kw)
        if t in ("circleloss", "circle"):
            return CircleLoss(p["feat_dim"], p["num_class"], p.get("margin", 0.25), p.get("gamma", 256), **self.kw)
        if t in ("mv-softmax", "mv_softmax"):
            return MV_Softmax(p["feat_dim"], p["num_class"], p["is_am"], p.get("margin", 0.35), p.get("mv_weight", 1.12), p.get("scale", 32), **self.kw)
        if t == "magface":
            return MagFace(p["feat_dim"], p["num_class"], p.get("margin_am", 0.0), p.get("scale", 32), p.get("l_a", 10), p.get("u_a", 110), p.get("l_margin", 0.45),
                           p.get("u_margin", 0.8), p.get("lamda", 20), **self.kw)
        raise KeyError(f"unknown head type {self.head_type}")


def sharded_margin_ce(head: _MarginHead, feats: torch.Tensor, labels: torch.Tensor, weight_shard: torch.Tensor, c_base: int, num_class: int, group=None,
                      label_smoothing: float = 0.0, grad_scale: Optional[float] = None):
    """Class-sharded form of `head.margin_ce` for data-parallel training with many identities: this rank holds the weight columns
    [c_base, c_base + weight_shard.shape[1]) of the [D, num_class] head.  Features and labels of ALL ranks are all-gathered (B_total x D floats), every rank
    scores them against its shard, three small all-reduces ([B_total] target cosine, [B_total] max, [B_total, 3] sums) give the global softmax statistics,
    and the feature gradient is summed over the shards ([B_total, D]).  The [D, C] head gradient never crosses a link: each rank gets the gradient of its own
    shard.  Returns (loss_rows of the LOCAL samples, dfeats of the local samples, dweight of the shard); grad_scale defaults to 1 / B_local, the scale
    `margin_ce` uses, so that the optimizer's 1 / world factor applies to both forms alike."""
    import torch.distributed as dist
    be = head.be
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, D = feats.shape
    if world > 1:
        fl = [torch.empty_like(feats) for _ in range(world)]; ll = [torch.empty_like(labels) for _ in range(world)]
        dist.all_gather(fl, feats.contiguous(), group=group); dist.all_gather(ll, labels.contiguous(), group=group)   # equal per-rank batches (DistributedSampler, drop_last)
        fall, lall = torch.cat(fl, 0), torch.cat(ll, 0)
    else:
        fall, lall = feats.contiguous(), labels.contiguous()
    Bt, Cloc = fall.shape[0], weight_shard.shape[1]
    st = _forward_cos(be, fall, weight_shard.detach())
    dev = feats.device
    gt = torch.empty(Bt, dtype=torch.float32, device=dev)
    be.check(be.lib.vdk_margin_target_cos(be.ptr(st.cos), st.Cp, Bt, Cloc, c_base, be.ptr(lall), be.ptr(gt), be.stream()), "vdk_margin_target_cos")
    if world > 1:
        dist.all_reduce(gt, op=dist.ReduceOp.SUM, group=group)
    stats = torch.empty((Bt, 4), dtype=torch.float32, device=dev)
    be.check(be.lib.vdk_margin_stats(C.byref(head.cfg), be.ptr(st.cos), st.Cp, Bt, Cloc, c_base, be.ptr(lall), be.ptr(gt), be.ptr(stats), be.stream()), "vdk_margin_stats")
    gmax = stats[:, 0].contiguous()
    if world > 1:
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    sums = torch.stack([stats[:, 1] * torch.exp(stats[:, 0] - gmax), stats[:, 2], stats[:, 3]], 1).contiguous()
    if world > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    gsum = sums[:, 0].contiguous()
    loss_all = gmax + torch.log(gsum) - (1.0 - label_smoothing) * sums[:, 2] - label_smoothing * sums[:, 1] / num_class
    dcos = torch.zeros((st.Bp, st.Cp), dtype=torch.bfloat16, device=dev)
    gs = 1.0 / B if grad_scale is None else grad_scale
    be.check(be.lib.vdk_margin_grad(C.byref(head.cfg), be.ptr(st.cos), st.Cp, Bt, Cloc, c_base, num_class, be.ptr(lall), be.ptr(gt), be.ptr(gmax), be.ptr(gsum),
                                    label_smoothing, gs, be.ptr(dcos), st.Cp, be.stream()), "vdk_margin_grad")
    df_all, dW = _backward_from_dcos(be, st, weight_shard.detach(), dcos)      # df of every sample w.r.t. THIS shard's columns
    if world > 1:
        dist.all_reduce(df_all, op=dist.ReduceOp.SUM, group=group)
    lo = rank * B
    return loss_all[lo:lo + B].contiguous(), df_all[lo:lo + B].contiguous(), dW
