"""faceX / CBIR model path on the HIP kernels: the reference's `TimmWrapper` (backbone + embedding neck),
`FaceTrainingModel` / `FaceTrainingWrapper`, `FeatureExtractor`, and the `get_model` factory seam.

Reference: models/faceX/backbone/timm_wrapper.py:5-54, models/faceX/face_model.py:10-54,88-143, models/smartmodel.py:5-10.
Transformer backbones (timm ViT ids in visiondk_amd.vit.TIMM_VITS): neck LayerNorm(C) -> Flatten -> Linear(N*C, feat_dim) ->
BatchNorm1d(feat_dim) (timm_wrapper.py:39-47).  CNN backbones (timm ConvNeXt ids in visiondk_amd.convnext.TIMM_CONVNEXTS): neck
BatchNorm2d(C) -> Flatten -> Linear(C*H*W, feat_dim) -> BatchNorm1d(feat_dim) (timm_wrapper.py:30-37).  state_dict keys equal the
reference's: `model.<timm keys>`, `output_layer.0.*` (LayerNorm | BatchNorm2d), `output_layer.2.*` (Linear), `output_layer.3.*`."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _abi, _lib, convnext, heads, ops, resnet, swin, vit


def _up(x, a):
    return (x + a - 1) // a * a


_BN_TALL_ROWS = 4096      # from this many rows on a per-rank BatchNorm runs as the slab-parallel kernel sequence (partials, combine, apply) that SyncBatchNorm uses


def _bn_sync_cb(ws: torch.Tensor, group):
    """vdk_stat_sync_fn of the neck's SyncBatchNorm: SUM all-reduce of the statistics vector the kernel sequence hands over (it lives inside `ws`)"""
    import torch.distributed as dist

    def cb(user, ptr, n):
        off = ptr - ws.data_ptr()
        dist.all_reduce(ws[off:off + 4 * n].view(torch.float32), op=dist.ReduceOp.SUM, group=group)
    return _abi.STAT_SYNC_FN(cb)


def _bn_rows_fwd(be, x, R, Cc, w, b, bn, training, y, sm, si, sync_group):
    """BatchNorm over R rows of x f32 [R, Cc] -> y f32.  sync_group False: the fused per-rank kernel; None / a process group (training): SyncBatchNorm -- statistics
    kernel, all-reduce of (sum x, sum x^2, count), apply (the reference converts every BatchNorm when `sync_bn` is set, engine/vision_engine.py:224-225)."""
    tall = R >= _BN_TALL_ROWS and Cc % 4 == 0          # BatchNorm2d over the map's rows (25 088 x 1024 at cfg3): the slab-parallel kernel sequence; the one-kernel form walks each column with 16 row groups
    if (sync_group is False or not training) and not tall:
        be.check(be.lib.vdk_batchnorm1d_fwd(be.ptr(x), Cc, R, Cc, be.ptr(w), be.ptr(b), bn.eps, bn.momentum, int(training), be.ptr(bn.running_mean),
                                            be.ptr(bn.running_var), be.ptr(y), Cc, be.ptr(sm), be.ptr(si), be.stream()), "vdk_batchnorm1d_fwd")
        return
    ws = ops._bn_ws(be, R, Cc, x.device)
    cb = _bn_sync_cb(ws, sync_group) if (sync_group is not False and training) else None
    be.check(be.lib.vdk_bn_act_fwd(be.ptr(x), R, Cc, be.ptr(w), be.ptr(b), bn.eps, bn.momentum, int(training), be.ptr(bn.running_mean), be.ptr(bn.running_var), None, None, 0,
                                   None, be.ptr(y), be.ptr(sm), be.ptr(si), be.ptr(ws), ws.numel(), cb, None, be.stream()), "vdk_bn_act_fwd")


def _bn_rows_bwd(be, dy, x, R, Cc, w, sm, si, dx, dw, db, sync_group):
    if sync_group is False and not (R >= _BN_TALL_ROWS and Cc % 4 == 0):
        be.check(be.lib.vdk_batchnorm1d_bwd(be.ptr(dy), Cc, be.ptr(x), Cc, R, Cc, be.ptr(w), be.ptr(sm), be.ptr(si), be.ptr(dx), Cc, be.ptr(dw), be.ptr(db), be.stream()),
                 "vdk_batchnorm1d_bwd")
        return
    ws = ops._bn_ws(be, R, Cc, x.device)
    cb = _bn_sync_cb(ws, sync_group) if sync_group is not False else None
    be.check(be.lib.vdk_bn_rows_bwd(be.ptr(x), be.ptr(dy), R, Cc, be.ptr(w), be.ptr(sm), be.ptr(si), be.ptr(dx), be.ptr(dw), be.ptr(db), be.ptr(ws), ws.numel(), cb, None,
                                    be.stream()), "vdk_bn_rows_bwd")


class _NeckFn(torch.autograd.Function):
    """LayerNorm(C) -> Flatten -> Linear(N*C, F) -> BatchNorm1d(F) as one autograd node over the HIP kernels"""

    @staticmethod
    def forward(ctx, tokens, ln_w, ln_b, lin_w, lin_b, bn_w, bn_b, wrapper):
        be = wrapper.be
        B, N, D = tokens.shape
        Fd = lin_w.shape[0]
        K = N * D
        Bp = _up(B, 64)
        dev = tokens.device
        tokens = tokens.contiguous()
        bn = wrapper.output_layer[3]
        # LayerNorm over C (eps 1e-5: nn.LayerNorm default, timm_wrapper.py:42) -> bf16 rows, viewed as [Bp, N*D] (pad rows zero)
        dt16 = getattr(wrapper, "dt16", torch.bfloat16)      # the backbone's operand format (bf16 | fp16)
        h = torch.zeros((Bp, K), dtype=dt16, device=dev)
        mean = torch.empty(B * N, dtype=torch.float32, device=dev)
        rstd = torch.empty(B * N, dtype=torch.float32, device=dev)
        be.check(be.lib.vdk_layernorm_fwd(be.ptr(tokens), D, B * N, D, be.ptr(ln_w), be.ptr(ln_b), 1e-5, be.ptr(h), D, _abi.F16_ if dt16 == torch.float16 else _abi.BF16,
                                          be.ptr(mean), be.ptr(rstd), be.stream()), "vdk_layernorm_fwd")
        wb = ops.cast_16(lin_w.detach().contiguous(), dt16, backend=be)                  # [F, N*D] 16-bit
        z = ops.gemm_nt(h[:B], wb, out_dtype=torch.float32, bias=lin_b.detach(), backend=be)     # [B, F]
        y = torch.empty_like(z)
        training = bool(wrapper.training)
        sm = torch.empty(Fd, dtype=torch.float32, device=dev)
        si = torch.empty(Fd, dtype=torch.float32, device=dev)
        sg = getattr(wrapper, "sync_group", False)
        _bn_rows_fwd(be, z, B, Fd, bn_w, bn_b, bn, training, y, sm, si, sg)
        if training:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(tokens, ln_w, bn_w, mean, rstd, h, wb, z, sm, si)
        ctx.wrapper, ctx.dims, ctx.sg = wrapper, (B, N, D, Fd, K, Bp), sg
        return y

    @staticmethod
    def backward(ctx, dy):
        tokens, ln_w, bn_w, mean, rstd, h, wb, z, sm, si = ctx.saved_tensors
        wrapper = ctx.wrapper
        be = wrapper.be
        B, N, D, Fd, K, Bp = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        dz = torch.empty((B, Fd), dtype=torch.float32, device=dev)
        dbn_w = torch.empty(Fd, dtype=torch.float32, device=dev); dbn_b = torch.empty(Fd, dtype=torch.float32, device=dev)
        _bn_rows_bwd(be, dy, z, B, Fd, bn_w, sm, si, dz, dbn_w, dbn_b, ctx.sg)
        stage = torch.zeros((Bp, _up(Fd, 8)), dtype=torch.float32, device=dev)
        stage[:B, :Fd].copy_(dz)
        dzb = ops.cast_16(stage, h.dtype, backend=be)
        dlin_b = ops.reduce_rows(dz, backend=be)
        # dW [F, N*D] = dz^T h : TN kernel straight from dz [Bp, F] and h [Bp, N*D]
        dlin_w = ops.gemm_nt(dzb, h, out_dtype=torch.float32, trans=True, backend=be)[:Fd]
        # dh [Bp, N*D] = dz W : TN kernel with A = dz^T [F, Bp] (tiny transpose), B = W [F, N*D] as it lies
        dzt = ops.transpose_pad(dzb, rpad=Bp, backend=be)                    # [Fp, Bp]
        fpad = dzt.shape[0]
        if fpad % 64 == 0 and fpad == Fd:
            dh = ops.gemm_nt(dzt, wb, out_dtype=h.dtype, trans=True, backend=be)                   # [Bp, N*D]
        else:   # feature dims that are not multiples of 64: NT kernel against an explicit W^T copy
            wbt = ops.transpose_pad(wb, rpad=_up(Fd, 8), backend=be)          # [N*D, Fp]
            dh = ops.gemm_nt(dzb, wbt, out_dtype=h.dtype, backend=be)
        dtok, _, dln_w, dln_b = ops.layernorm_bwd(dh[:B].reshape(B * N, D), tokens.view(B * N, D), mean, rstd, ln_w, want_bf16=False, backend=be)
        return dtok.view(B, N, D), dln_w, dln_b, dlin_w.contiguous(), dlin_b, dbn_w, dbn_b, None


class _NeckCNNFn(torch.autograd.Function):
    """BatchNorm2d(C) -> Flatten -> Linear(C*H*W, F) -> BatchNorm1d(F) (timm_wrapper.py:30-37) as one autograd node.  The feature map
    lives as NHWC rows [B*H*W, C], so BatchNorm2d is the row BatchNorm kernel over B*H*W samples, and the Linear runs on the NHWC
    flattening against a column-permuted view of its weight (W'[f, p*C + c] = W[f, c*HW + p]: a layout change, no arithmetic)."""

    @staticmethod
    def forward(ctx, fmap, bn2_w, bn2_b, lin_w, lin_b, bn_w, bn_b, wrapper):
        be = wrapper.be
        B, Cc, Hh, Ww = fmap.shape
        HW, K, Fd = Hh * Ww, Cc * Hh * Ww, lin_w.shape[0]
        Bp = _up(B, 64)
        dev = fmap.device
        rows = fmap.permute(0, 2, 3, 1).contiguous().view(B * HW, Cc)          # no copy when the backbone handed an NHWC-backed view
        bn2, bn = wrapper.output_layer[0], wrapper.output_layer[3]
        training = bool(wrapper.training)
        y2 = torch.zeros((Bp * HW, Cc), dtype=torch.float32, device=dev)
        sm2 = torch.empty(Cc, dtype=torch.float32, device=dev); si2 = torch.empty(Cc, dtype=torch.float32, device=dev)
        sg = getattr(wrapper, "sync_group", False)
        _bn_rows_fwd(be, rows, B * HW, Cc, bn2_w, bn2_b, bn2, training, y2, sm2, si2, sg)
        wperm = lin_w.detach().view(Fd, Cc, HW).permute(0, 2, 1).reshape(Fd, K).contiguous()
        fp32 = getattr(wrapper, "precision", "bf16") == "fp32"      # FaceTrainStep(precision="fp32"): the Linear on the fp32 MFMA, operands as they are
        if fp32:
            h, wb = y2.view(Bp, K), wperm
            z = ops.gemm_f32(h[:B], wb, bias=lin_b.detach(), backend=be)
        else:
            dt16 = getattr(wrapper, "dt16", torch.bfloat16)                    # the backbone's operand format (bf16 | fp16)
            h = ops.cast_16(y2, dt16, backend=be).view(Bp, K)                  # 16-bit [Bp, HW*C], pad rows zero
            wb = ops.cast_16(wperm, dt16, backend=be)
            # [B, F] from K = HW * C = 50 176: 4 output tiles of 256 x 256 would leave the contraction to 4 CUs (826 us at B = F = 512); split over K instead
            tiles = ((B + 255) // 256) * ((Fd + 255) // 256)
            sk = max(1, min(256 // tiles, K // 64 // 8))
            if sk > 1 and K % 64 == 0:
                z = ops.gemm_nt(h[:B], wb, out_dtype=torch.float32, splitk=sk, backend=be)
                z.add_(lin_b.detach())
            else:
                z = ops.gemm_nt(h[:B], wb, out_dtype=torch.float32, bias=lin_b.detach(), backend=be)
        y = torch.empty_like(z)
        sm = torch.empty(Fd, dtype=torch.float32, device=dev); si = torch.empty(Fd, dtype=torch.float32, device=dev)
        _bn_rows_fwd(be, z, B, Fd, bn_w, bn_b, bn, training, y, sm, si, sg)
        if training:
            bn.num_batches_tracked += 1
            bn2.num_batches_tracked += 1
        ctx.save_for_backward(rows, bn2_w, bn_w, h, wb, z, sm, si, sm2, si2)
        ctx.wrapper, ctx.dims, ctx.sg = wrapper, (B, Cc, Hh, Ww, Fd, K, Bp), sg
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, bn2_w, bn_w, h, wb, z, sm, si, sm2, si2 = ctx.saved_tensors
        be = ctx.wrapper.be
        B, Cc, Hh, Ww, Fd, K, Bp = ctx.dims
        HW = Hh * Ww
        dev = dy.device
        dy = dy.contiguous()
        dz = torch.empty((B, Fd), dtype=torch.float32, device=dev)
        dbn_w = torch.empty(Fd, dtype=torch.float32, device=dev); dbn_b = torch.empty(Fd, dtype=torch.float32, device=dev)
        _bn_rows_bwd(be, dy, z, B, Fd, bn_w, sm, si, dz, dbn_w, dbn_b, ctx.sg)
        dlin_b = ops.reduce_rows(dz, backend=be)
        if h.dtype == torch.float32:                                                                           # precision="fp32": both gradients on the fp32 MFMA, operands read as they lie
            dwp = ops.gemm_f32(dz, h[:B], a_kmajor=True, b_kmajor=True, backend=be)                            # dz^T h: [F, HW*C]
            dlin_w = dwp.view(Fd, HW, Cc).permute(0, 2, 1).reshape(Fd, K)
            dh = ops.gemm_f32(dz, wb, b_kmajor=True, backend=be).reshape(B * HW, Cc)                           # dz W': [B, HW*C]
        else:
            stage = torch.zeros((Bp, _up(Fd, 8)), dtype=torch.float32, device=dev)
            stage[:B, :Fd].copy_(dz)
            dzb = ops.cast_16(stage, h.dtype, backend=be)
            dwp = ops.gemm_nt(dzb, h, out_dtype=torch.float32, trans=True, backend=be)[:Fd]                      # [F, HW*C]
            dlin_w = dwp.view(Fd, HW, Cc).permute(0, 2, 1).reshape(Fd, K)                                          # back to the NCHW column order
            dzt = ops.transpose_pad(dzb, rpad=Bp, backend=be)
            fpad = dzt.shape[0]
            if fpad % 64 == 0 and fpad == Fd:
                dh = ops.gemm_nt(dzt, wb, out_dtype=torch.float32, trans=True, backend=be)                         # [Bp, HW*C]
            else:
                wbt = ops.transpose_pad(wb, rpad=_up(Fd, 8), backend=be)
                dh = ops.gemm_nt(dzb, wbt, out_dtype=torch.float32, backend=be)
            dh = dh[:B].reshape(B * HW, Cc)
        dx = torch.empty((B * HW, Cc), dtype=torch.float32, device=dev)
        dbn2_w = torch.empty(Cc, dtype=torch.float32, device=dev); dbn2_b = torch.empty(Cc, dtype=torch.float32, device=dev)
        _bn_rows_bwd(be, dh.contiguous(), rows, B * HW, Cc, bn2_w, sm2, si2, dx, dbn2_w, dbn2_b, ctx.sg)
        return dx.view(B, Hh, Ww, Cc).permute(0, 3, 1, 2), dbn2_w, dbn2_b, dlin_w.contiguous(), dlin_b, dbn_w, dbn_b, None


class TimmWrapper(nn.Module):
    """models/faceX/backbone/timm_wrapper.py: timm backbone without head/pool + embedding neck -> [B, feat_dim]."""

    def __init__(self, model_name: str, feat_dim: int, image_size: int, pretrained: bool = True, backend: Optional[_lib.Backend] = None,
                 device=None, operand: str = "bf16", **kwargs):
        """operand: "bf16" | "fp16", the 16-bit format of the backbone's and the neck's GEMM operands.  The reference runs this path in fp32 (no autocast in the face / CBIR
        loop, engine/procedure/train.py:217-227): fp16 operands keep the embeddings within 1e-3 and the gradients within 5e-3 of that arithmetic at the bf16 speed
        (tests/test_parity_fullsize_gpu.py), with FaceTrainStep running the GradScaler protocol the reference's loop also runs (train.py:205-211)."""
        super().__init__()
        if operand not in ("bf16", "fp16"):
            raise ValueError("operand must be 'bf16' or 'fp16'")
        self.be = backend or _lib.load()
        dev = device if device is not None else ("cuda" if self.be.device_only else "cpu")
        self.is_cnn = model_name in convnext.TIMM_CONVNEXTS
        self.operand = operand
        self.dt16 = torch.float16 if operand == "fp16" else torch.bfloat16
        if self.is_cnn:
            self.model = convnext.create_model(model_name, pretrained=False, num_classes=0, global_pool="", img_size=image_size, device=dev, backend=self.be,
                                               operand=operand)
            channels, hw = self.model.engine.out_ch, self.model.engine.out_hw
            self.output_layer = nn.Sequential(nn.BatchNorm2d(channels), nn.Flatten(1), nn.Linear(channels * hw * hw, feat_dim), nn.BatchNorm1d(feat_dim)).to(dev)
        elif model_name in vit.TIMM_VITS:
            self.model = vit.create_model(model_name, pretrained=False, num_classes=0, global_pool="", img_size=image_size, device=dev, backend=self.be, operand=operand)
            tokens, channels = self.model.engine.tokens, self.model.spec.dim
            self.output_layer = nn.Sequential(nn.LayerNorm(channels), nn.Flatten(1), nn.Linear(tokens * channels, feat_dim), nn.BatchNorm1d(feat_dim)).to(dev)
        elif model_name in swin.TIMM_SWINS:
            # timm's Swin returns an NHWC map [B, 7, 7, C] for global_pool='', and the reference's wrapper reads ANY 4-D output as [B, channels, h, w]
            # (timm_wrapper.py:28-37): with this backbone -- the default of cbir.yaml:26 -- its neck is BatchNorm2d(7) over the map's ROW index, Flatten, Linear(7 * 7 * C, feat_dim),
            # BatchNorm1d.  Reproduced as it is: the NHWC tensor goes into the CNN neck as if it were NCHW.
            self.model = swin.create_model(model_name, pretrained=False, num_classes=0, img_size=image_size, device=dev, backend=self.be, operand=operand)
            self.is_cnn = True
            hw, channels = image_size // 32, self.model.num_features
            self.output_layer = nn.Sequential(nn.BatchNorm2d(hw), nn.Flatten(1), nn.Linear(hw * hw * channels, feat_dim), nn.BatchNorm1d(feat_dim)).to(dev)
        else:
            raise NotImplementedError(f"backbone '{model_name}': the HIP engines cover {sorted(vit.TIMM_VITS)}, {sorted(convnext.TIMM_CONVNEXTS)} and {sorted(swin.TIMM_SWINS)}")

    @torch.no_grad()
    def forward_precise(self, x):
        """Evaluation embedding with fp32 activations and fp32-MFMA contractions end to end (backbone + neck in eval mode): agrees with the reference's
        PyTorch-CPU fp32 embedding to ~1e-6, so downstream cosine top-k lists match the reference's.  Used by FeatureExtractor(precise=True)."""
        be, ol = self.be, self.output_layer
        if not hasattr(self.model, "forward_precise"):
            raise NotImplementedError("forward_precise (fp32-MFMA evaluation) is built for the ViT and ConvNeXt engines; the Swin backbone evaluates on bf16 operands")
        feat = self.model.forward_precise(x)
        B = x.shape[0]
        if self.is_cnn:
            Cc, HW = feat.shape[1], feat.shape[2] * feat.shape[3]
            rows = feat.permute(0, 2, 3, 1).contiguous().view(B * HW, Cc)
            y, _, _ = ops.batchnorm_fwd(rows, ol[0].weight.detach(), ol[0].bias.detach(), ol[0].running_mean, ol[0].running_var, training=False, eps=ol[0].eps,
                                        backend=be)
            a = y.view(B, HW * Cc)
            w = ol[2].weight.detach().view(-1, Cc, HW).permute(0, 2, 1).reshape(-1, HW * Cc).contiguous()     # NCHW-flatten columns -> NHWC-flatten columns
        else:
            N, D = feat.shape[1], feat.shape[2]
            y, _, _ = ops.layernorm_fwd(feat.contiguous().view(B * N, D), ol[0].weight.detach(), ol[0].bias.detach(), eps=ol[0].eps, out_dtype=torch.float32, backend=be)
            a = y.view(B, N * D)
            w = ol[2].weight.detach().contiguous()
        z = ops.gemm_f32(a, w, bias=ol[2].bias.detach(), backend=be)
        out, _, _ = ops.batchnorm_fwd(z, ol[3].weight.detach(), ol[3].bias.detach(), ol[3].running_mean, ol[3].running_var, training=False, eps=ol[3].eps, backend=be)
        return out

    def forward(self, x):
        feat = self.model(x)
        ol = self.output_layer
        fn = _NeckCNNFn if self.is_cnn else _NeckFn
        return fn.apply(feat, ol[0].weight, ol[0].bias, ol[2].weight, ol[2].bias, ol[3].weight, ol[3].bias, self)


class BackboneFactory:
    """models/faceX/backbone/backbone_def.py:5-26 — config {'timm-<id>': {pretrained, image_size, feat_dim}}"""

    def __init__(self, backbone_conf: dict, backend=None, device=None):
        (self.name, self.param), = backbone_conf.items()
        self.kw = dict(backend=backend, device=device)

    def get_backbone(self):
        assert self.name.startswith("timm-"), "backbone id must look like timm-<timm model id>"
        model_id = self.name[5:].split(".")[0]          # 'timm-vit_base_patch16_224.augreg2_in21k_ft_in1k' -> architecture id
        # `operand` is this library's own key next to the reference's three (backbone_def.py:17-24): "fp16" (default) | "bf16" (TimmWrapper).  The reference runs the face /
        # CBIR loop in fp32 (no autocast, engine/procedure/train.py:217-227): fp16 operands are the fast mode inside north_star's tolerance of that arithmetic
        # (tests/test_parity_fullsize_gpu.py: embeddings <= 1e-3, gradients <= 5e-3 at ConvNeXt-B + ArcFace(10^6)); bf16 (5.8e-3 / 4e-2 there) is the opt-out.
        return TimmWrapper(model_id, feat_dim=self.param["feat_dim"], image_size=self.param["image_size"],
                           pretrained=False, operand=self.param.get("operand", "fp16"), **self.kw)


class FaceTrainingModel(nn.Module):
    """models/faceX/face_model.py:28-54: `forward(data, label) = head(backbone(data), label)`"""

    def __init__(self, model_cfg: dict, backend=None, device=None):
        super().__init__()
        backbone = BackboneFactory(model_cfg["backbone"], backend=backend, device=device).get_backbone()
        (htype, hconf), = model_cfg["head"].items()
        head = heads.HeadFactory(htype, hconf, backend=backend, device=device).get_head()
        self.trainingwrapper = nn.ModuleDict({"backbone": backbone, "head": head})

    def forward(self, data, label):
        feat = self.trainingwrapper["backbone"](data)
        return self.trainingwrapper["head"](feat, label)


class FaceTrainingWrapper:
    """models/faceX/face_model.py:10-26"""

    def __init__(self, model_cfg, logger=None, backend=None, device=None):
        self.model = FaceTrainingModel(model_cfg, backend=backend, device=device)
        self.logger = logger

    def reset_parameters(self):   # defined but never called by the reference; kept for surface parity
        for m in self.model.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.02)
                nn.init.constant_(m.bias, 0)


class FaceModelLoader:
    """models/faceX/face_model.py:56-86 — build the backbone from the config and load a reference checkpoint: `ckpt['state_dict']` (the training
    backbone) or `ckpt['ema']` (a state_dict for the face / CBIR tasks, train.py:266-278); keys are the reference's (`model.<timm keys>`,
    `output_layer.*`), strict."""

    def __init__(self, model_cfg: dict, backend=None, device=None):
        self.model = BackboneFactory(model_cfg["backbone"], backend=backend, device=device).get_backbone()

    def load_weight_default(self, model_path):
        self.model.load_state_dict(torch.load(model_path, weights_only=False, map_location="cpu")["state_dict"], strict=True)
        return self.model

    def load_weight(self, model_path, ema: bool = False):
        ckpt = torch.load(model_path, weights_only=False, map_location="cpu")
        self.model.load_state_dict(ckpt["ema"] if ema else ckpt["state_dict"], strict=True)
        return self.model


class FeatureExtractor:
    """models/faceX/face_model.py:88-143 — eval forward -> F.normalize -> host numpy (order = loader order)"""

    def __init__(self, model, precise: bool = False):
        """precise=True: embeddings from the fp32-MFMA forward (TimmWrapper.forward_precise), ~1e-6 from the reference's CPU embeddings"""
        self.model = model
        self.precise = precise

    def extract_face(self, dataloader, device) -> dict:
        """face_model.py:93-117 — {`<parent dir>/<file name>`: unit embedding} for a loader yielding (_, tensors, file_realpaths)"""
        import os
        from . import cbir
        model = self.model
        model.eval()
        out = {}
        with torch.no_grad():
            for _, tensors, paths in dataloader:
                tensors = tensors.to(device)
                f = model.forward_precise(tensors) if self.precise else model(tensors)
                f = cbir.l2_normalize(f.contiguous(), backend=getattr(model, "be", None)).cpu().numpy()
                for path, feat in zip(paths, f):
                    out[os.path.join(os.path.basename(os.path.dirname(path)), os.path.basename(path))] = feat
        return out

    def extract_cbir(self, dataloader, device) -> np.ndarray:
        from . import cbir
        model = self.model
        model.eval()
        feats = []
        with torch.no_grad():
            for tensors in dataloader:
                tensors = tensors.to(device)
                feature = model.forward_precise(tensors) if self.precise else model(tensors)
                feature = cbir.l2_normalize(feature.contiguous(), backend=getattr(model, "be", None))
                feats.append(feature.cpu().numpy())
        return np.concatenate(feats, axis=0)


def get_model(model_cfg: dict, logger=None, rank: int = 0, backend=None, device=None):
    """models/smartmodel.py:5-10 — task 'face' | 'cbir' -> FaceTrainingWrapper, 'classification' -> VisionWrapper."""
    assert "task" in model_cfg, "Task is not specified"
    task = model_cfg["task"]
    if device is None and (backend is None or backend.device_only):
        device = f"cuda:{rank}"
    if task in ("face", "cbir"):
        return FaceTrainingWrapper(model_cfg, logger, backend=backend, device=device)
    if task == "classification":
        return VisionWrapper(model_cfg, logger, rank, backend=backend, device=device)
    raise ValueError(f"unknown task {task}")


class VisionWrapper:
    """models/classifier/classify_model.py:10-68 — `name: timm-<id>`, `.model`, `.reset_parameters()`, `.load_weight()`"""

    def __init__(self, model_cfg: dict, logger=None, rank: int = 0, backend=None, device=None):
        self.logger = logger
        name = model_cfg["name"]
        assert name.startswith("timm-"), "classifier id must look like timm-<timm model id>"
        kwargs = model_cfg.get("kwargs") or {}
        # classify_model.py:16-33,60-66: accepted keys.  In the reference backbone_freeze / bn_freeze call self.freeze_*() from inside create_model, before
        # self.model exists (AttributeError), so only their defaults are live configurations.  attention_pool=True is a no-op there (atten_pool_replace
        # returns the model unchanged, built/attention_based_pooler.py:30-47) and is accepted as such here.
        for opt in ("backbone_freeze", "bn_freeze", "bn_freeze_affine"):
            if model_cfg.get(opt, False):
                raise NotImplementedError(f"model.{opt}=True is not built on the HIP engines (it raises AttributeError in the reference as well)")
        arch = name[5:].split(".")[0]
        # timm-resnet18 | timm-convnext_* (pet.yaml:21-22) | timm-vit_*
        factory = (resnet.create_model if arch in resnet.TIMM_RESNETS else convnext.create_model if arch in convnext.TIMM_CONVNEXTS else
                   swin.create_model if arch in swin.TIMM_SWINS else vit.create_model)      # timm-swin_base_patch4_window7_224 is pet.yaml:25's default
        self.model = factory(arch, pretrained=False, num_classes=model_cfg["num_classes"], img_size=model_cfg.get("image_size") or 224, device=device,
                             backend=backend, **kwargs)
        if not model_cfg.get("pretrained", False):
            self.reset_parameters()

    def reset_parameters(self):
        self.model.reset_parameters()

    def load_weight(self, load_path: str, ema: bool = False, device="cpu"):
        ckpt = torch.load(load_path, map_location=device, weights_only=False)
        sd = ckpt["ema"].state_dict() if ema and hasattr(ckpt.get("ema"), "state_dict") else ckpt.get("ema" if ema else "model", ckpt)
        self.model.load_state_dict(sd)


class FaceTrainStep:
    """Trainer.compute_loss(face=True) + Trainer.update for the face / CBIR task (engine/procedure/train.py:192-215,217-226) as a fixed
    kernel sequence: backbone forward (native engine) -> neck -> margin head fused with CrossEntropy (no B x C logits through torch) ->
    backward -> clip_grad_norm_(max_norm) over ALL parameters -> SGD(momentum, weight_decay) -> ModelEMA.update (models/ema.py:28-37).

    The backbone's parameters live in the engine's flat buffer (one optimizer launch); the neck and head tensors (7 of them) get one
    launch each with the same global clip factor.  BatchNorm running statistics enter the EMA like every float entry of state_dict()."""

    def __init__(self, model: "FaceTrainingModel", lr: float, momentum: float = 0.9, weight_decay: float = 5e-4, label_smoothing: float = 0.0,
                 max_norm: float = 10.0, ema: bool = True, comm=None, layer_wise: bool = False, shard_head: bool = False, sync_bn: bool = False,
                 cos_planes: int = 3, precision: str = "bf16", init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        """precision: "bf16" = 16-bit MFMA operands in the BACKBONE's operand format (TimmWrapper(operand=...): bf16 or fp16) with fp32 accumulation, residual stream and
        master weights; "fp32" = the arithmetic of the reference's face / CBIR loop, which runs WITHOUT autocast (engine/procedure/train.py:217-227): fp32 activations, every
        contraction of the backbone and of the neck's Linear on the fp32 MFMA, split-plane (fp32-class) cosines in the head -- embeddings ~1e-5 and gradients ~1e-4 from
        the fp32 oracle at ~1/6 of the bf16 step's speed (CNN backbones; tests/test_face.py, tests/test_parity_fullsize_gpu.py).
        An fp16 backbone (operand="fp16") is the fast mode that meets north_star's tolerance against that fp32 loop (embeddings <= 1e-3, gradients <= 5e-3 at ConvNeXt-B +
        ArcFace(10^6), tests/test_parity_fullsize_gpu.py): backbone, neck and head multiply fp16 operands, and the step runs the GradScaler protocol the reference's face
        loop runs as well -- `Trainer.update(model, loss, self.scaler, ...)` (engine/procedure/train.py:199,203-215; the scaler is enabled on a GPU, vision_engine.py:232;
        only autocast is absent from that loop): the loss scale (device state {scale, growth tracker, skipped steps}; init_scale / growth_factor / backoff_factor /
        growth_interval = GradScaler's defaults) is multiplied into d(loss)/d(cos), every gradient carries it, vdk_sgd_step_amp un-scales, and a step whose gradients hold an
        inf / NaN is skipped while the EMA still moves.  With bf16 / fp32 arithmetic the scale stays 1 and only the inf / NaN skip remains -- exactly what the reference's
        scaler does for fp32 gradients.
        cos_planes: 3 = fp32-class cosines in the head (split-bf16 planes: the reference's CPU path), 1 = single bf16 operands (the reference's GPU path: the head runs
        under autocast, train.py:118).  shard_head (with comm): every rank keeps the columns [rank * C / world, (rank + 1) * C / world) of the margin head, trains them with
        `heads.sharded_margin_ce` (features all-gathered, per-row softmax statistics and the [B, D] feature gradient all-reduced) and never all-reduces the
        [D, C] head gradient (2 GB at C = 10^6).  `gather_head()` writes the shards back into `head.weight` for evaluation / checkpoints.
        layer_wise: the second entry of the yaml's `optimizer` list (cbir.yaml:113): two parameter groups, backbone + neck at lr and the
        margin head at 10 x lr (built/layer_optimizer.py:26-29); `param_groups[1]['lr']` is then the head's rate and a scheduler drives both.
        comm: visiondk_amd.comm.GradAllReduce for data parallelism (one process per GPU): parameters and BatchNorm buffers are broadcast from
        rank 0 at construction and the buffers again before every forward (torch DDP's broadcast_buffers=True, which the reference's
        DDP wrap at vision_engine.py:510 uses); the backbone's flat gradient is all-reduced in buckets while backward is still running, the neck /
        head gradients right after; BatchNorm statistics stay per-rank (the reference's default, SyncBN is its opt-in flag)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.cos_planes = 3 if precision == "fp32" else cos_planes
        self.model = model
        self.comm = comm
        self.bb = model.trainingwrapper["backbone"]
        if precision == "fp32" and shard_head:
            raise NotImplementedError("precision='fp32' with a class-sharded head is not built")
        if not hasattr(self.bb.model, "engine"):
            raise NotImplementedError("FaceTrainStep needs a backbone with a native engine (ViT, ConvNeXt, Swin with native=True); the autograd-node Swin trains under "
                                      "the reference's own Trainer (torch optimizer over model.parameters())")
        self.amp = getattr(self.bb.model.engine, "operand", "bf16") == "fp16"
        if self.amp and getattr(self.bb, "operand", "fp16") != "fp16":
            raise ValueError("an fp16 backbone engine under a TimmWrapper built with operand='bf16': build the wrapper with operand='fp16' (its neck follows the format)")
        if self.amp and precision == "fp32":
            raise ValueError("precision='fp32' runs fp32 activations: build the backbone with operand='bf16' (its 16-bit copies are not used in that mode)")
        if self.amp and shard_head and comm is not None and comm.active:      # (an inactive communicator -- one rank -- never shards: nothing to refuse)
            raise NotImplementedError("the class-sharded head is built for bf16 operands (its three passes take no loss scale yet): build the backbone with "
                                      "operand='bf16' -- `backbone: {timm-...: {operand: bf16}}` in the config -- when shard_head=True is used across ranks")
        if precision == "fp32" and not hasattr(self.bb.model.engine, "precision"):
            raise NotImplementedError("precision='fp32' is built for the ConvNeXt backbones of the face / CBIR task (the engine with an fp32-class training mode)")
        self.precision = precision
        self.bb.precision = precision
        if hasattr(self.bb.model.engine, "precision"):
            self.bb.model.engine.precision = precision
        # sync_bn (the reference's `sync_bn` flag converts every BatchNorm of the model, engine/vision_engine.py:224-225): the neck's BatchNorm2d / BatchNorm1d
        # all-reduce their batch statistics (forward) and gradient sums (backward) over comm's group
        self.bb.sync_group = comm.group if (sync_bn and comm is not None and comm.active) else False
        self.head = model.trainingwrapper["head"]
        self.eng = self.bb.model.engine
        self.be = self.eng.be
        self.lr, self.momentum, self.weight_decay, self.label_smoothing, self.max_norm = lr, momentum, weight_decay, label_smoothing, max_norm
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        if layer_wise:
            self.param_groups.append({"lr": lr * 10, "momentum": momentum, "weight_decay": weight_decay})
        self.updates = 0
        self.shard_head = bool(shard_head and comm is not None and comm.active)
        self.small = [p for p in self.bb.output_layer.parameters()] + ([] if self.shard_head else [self.head.weight])
        self.buffers = [b for b in self.bb.output_layer.buffers() if b.dtype.is_floating_point]
        mk = lambda t: torch.zeros_like(t)
        self.mom_flat = mk(self.eng.params)
        self.mom_small = [mk(p) for p in self.small]
        self.ema_flat = self.eng.params.clone() if ema else None
        self.ema_small = [p.detach().clone() for p in self.small] if ema else [None] * len(self.small)
        self.ema_buf = [b.detach().clone() for b in self.buffers] if ema else []
        self._zero = [mk(b) for b in self.buffers]
        self._zero_m = [mk(b) for b in self.buffers]
        self._nsq = torch.zeros(1, dtype=torch.float32, device=self.eng.device)
        self._nsq_part = torch.zeros(1, dtype=torch.float32, device=self.eng.device)
        # GradScaler state on the device (see the docstring): fp16 -> dynamic scale; otherwise the scale is pinned at 1 and the state only carries the inf / NaN skip
        self.loss_state = torch.tensor([init_scale if self.amp else 1.0, 0.0, 0.0], dtype=torch.float32, device=self.eng.device)
        self.growth_factor, self.backoff_factor, self.growth_interval = (growth_factor, backoff_factor, growth_interval) if self.amp else (1.0, 1.0, 1 << 30)
        need = C.c_size_t(0)
        self.be.check(self.be.lib.vdk_sumsq_workspace_bytes(C.byref(need)), "vdk_sumsq_workspace_bytes")
        self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.eng.device)
        self.loss_rows: Optional[torch.Tensor] = None
        if comm is not None and comm.active:
            import torch.distributed as dist
            comm.broadcast_params(self.eng.params, engine=self.eng)
            for t in self.small + self.buffers + ([self.head.weight] if self.shard_head else []):
                dist.broadcast(t.data, src=0, group=comm.group)
            if ema:
                self.ema_flat.copy_(self.eng.params)
                for e, p in zip(self.ema_small, self.small):
                    e.copy_(p.detach())
                for e, b in zip(self.ema_buf, self.buffers):
                    e.copy_(b)
        if self.shard_head:
            Cn = self.head.weight.shape[1]
            if Cn % comm.world_size:
                raise ValueError("shard_head needs num_class % world_size == 0")
            self.c_per = Cn // comm.world_size
            self.c0 = comm.rank * self.c_per
            self.hs = self.head.weight.detach()[:, self.c0:self.c0 + self.c_per].contiguous()      # this rank's columns, trained in place of head.weight
            self.hs_mom = torch.zeros_like(self.hs)
            self.hs_ema = self.hs.clone() if ema else None
            self._nsq_head = torch.zeros(1, dtype=torch.float32, device=self.eng.device)

    def _sumsq(self, g: torch.Tensor) -> None:
        be = self.be
        be.check(be.lib.vdk_sumsq_f32(be.ptr(g), g.numel(), be.ptr(self._nsq_part), be.ptr(self._ws), self._ws.numel(), be.stream()), "vdk_sumsq_f32")
        self._nsq += self._nsq_part          # 1-element accumulate of per-buffer partial norms

    def step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        import math
        bb, eng, be = self.bb, self.eng, self.be
        bb.model._sync_flat()
        self.updates += 1
        d = 0.9999 * (1 - math.exp(-self.updates / 2000)) if self.ema_flat is not None else 0.0
        # lr, momentum and weight decay are re-read every step: the reference's Trainer writes param_groups[i]['momentum'] after the warm-up
        # (vision_engine.py:169-171,544-546) and a scheduler drives every group's 'lr'; with layer_wise the last group is the margin head's
        g0, gh = self.param_groups[0], self.param_groups[-1]
        lr, mom, wd = g0["lr"], g0["momentum"], g0["weight_decay"]
        lr_head, mom_head, wd_head = gh["lr"], gh["momentum"], gh["weight_decay"]
        B = x.shape[0]
        world = self.comm.world_size if self.comm is not None else 1
        active = self.comm is not None and self.comm.active
        if active:
            import torch.distributed as dist
            for b in self.buffers:
                dist.broadcast(b, src=0, group=self.comm.group)
        # forward: backbone engine -> neck (autograd node over the HIP kernels) -> fused head + CE
        out = eng.forward(x)
        is_swin = isinstance(eng, swin.SwinEngine)
        if is_swin:                # the NHWC map goes into the CNN neck as it is (TimmWrapper: the reference reads any 4-D output as [B, channels, h, w])
            r = int(round(eng.map_rows ** 0.5))
            feat = out.view(B, r, r, eng.features).detach().requires_grad_(True)
        elif bb.is_cnn:
            feat = out.view(B, eng.out_hw, eng.out_hw, eng.out_ch).permute(0, 3, 1, 2).detach().requires_grad_(True)
        else:
            feat = out.view(B, eng.tokens, eng.spec.dim).detach().requires_grad_(True)
        ol = bb.output_layer
        fn = _NeckCNNFn if bb.is_cnn else _NeckFn
        emb = fn.apply(feat, ol[0].weight, ol[0].bias, ol[2].weight, ol[2].bias, ol[3].weight, ol[3].bias, bb)
        if self.shard_head:
            self.loss_rows, demb, dW = heads.sharded_margin_ce(self.head, emb.detach(), y, self.hs, self.c0, self.head.weight.shape[1], group=self.comm.group,
                                                              label_smoothing=self.label_smoothing)
        else:
            self.loss_rows, demb, dW = self.head.margin_ce(emb.detach(), y, self.label_smoothing, cos_planes=self.cos_planes, precise=self.precision == "fp32",
                                                           operand="fp16" if self.amp else "bf16", loss_scale=self.loss_state if self.amp else None)
        for p in self.small:
            p.grad = None
        emb.backward(demb)
        if not self.shard_head:
            self.head.weight.grad = dW
        dfeat = feat.grad
        if is_swin:
            dfeat = dfeat.contiguous().view(-1, eng.features)
        else:
            dfeat = dfeat.permute(0, 2, 3, 1).contiguous().view(-1, eng.out_ch) if bb.is_cnn else dfeat.contiguous().view(-1, eng.spec.dim)
        if active:
            import torch.distributed as dist
            self.comm.begin_step(eng.grads)
            eng.backward(dfeat, on_ready=self.comm.on_grad_ready)
            small_work = []
            for p in self.small:
                p.grad = p.grad.contiguous()
                small_work.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.comm.group, async_op=True))
            self.comm.finish_step()
            for w_ in small_work:
                w_.wait()
        else:
            eng.backward(dfeat)
        # clip_grad_norm_ over every parameter (of the rank-averaged gradient), then SGD + EMA
        self._nsq.zero_()
        self._sumsq(eng.grads)
        for p in self.small:
            self._sumsq(p.grad.contiguous())
        if self.shard_head:                                            # the head's share of the global gradient norm: sum over the shards
            import torch.distributed as dist
            be.check(be.lib.vdk_sumsq_f32(be.ptr(dW), dW.numel(), be.ptr(self._nsq_head), be.ptr(self._ws), self._ws.numel(), be.stream()), "vdk_sumsq_f32")
            dist.all_reduce(self._nsq_head, op=dist.ReduceOp.SUM, group=self.comm.group)
            self._nsq += self._nsq_head
        first = int(self.updates == 1)
        # scaler.unscale_ + clip + scaler.step + ema.update in one pass per buffer (train.py:205-215): every gradient carries loss_state[0]; a non-finite norm skips the update
        ls = be.ptr(self.loss_state)
        p16 = _abi.F16_ if self.amp else _abi.BF16
        be.check(be.lib.vdk_sgd_step_amp(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.mom_flat), be.ptr(self.ema_flat), be.ptr(eng.wb16), p16, eng.n_floats, lr,
                                         mom, wd, 1.0 / world, ls, be.ptr(self._nsq), self.max_norm, d, first, be.stream()), "vdk_sgd_step_amp")
        for p, m, e in zip(self.small, self.mom_small, self.ema_small):
            g = p.grad.contiguous()
            is_head = p is self.head.weight
            be.check(be.lib.vdk_sgd_step_amp(be.ptr(p.data), be.ptr(g), be.ptr(m), be.ptr(e), None, p16, p.numel(), lr_head if is_head else lr, mom_head if is_head else mom,
                                             wd_head if is_head else wd, 1.0 / world, ls, be.ptr(self._nsq), self.max_norm, d, first, be.stream()), "vdk_sgd_step_amp")
        if self.shard_head:
            be.check(be.lib.vdk_sgd_step_amp(be.ptr(self.hs), be.ptr(dW), be.ptr(self.hs_mom), be.ptr(self.hs_ema), None, p16, self.hs.numel(), lr_head, mom_head,
                                             wd_head, 1.0 / world, ls, be.ptr(self._nsq), self.max_norm, d, first, be.stream()), "vdk_sgd_step_amp")
        be.check(be.lib.vdk_loss_scale_update(ls, be.ptr(self._nsq), self.growth_factor, self.backoff_factor, self.growth_interval, be.stream()), "vdk_loss_scale_update")
        for b, z, zm, e in zip(self.buffers, self._zero, self._zero_m, self.ema_buf):   # EMA of the BatchNorm running statistics (lr = 0: value untouched)
            be.check(be.lib.vdk_sgd_step(be.ptr(b), be.ptr(z), be.ptr(zm), be.ptr(e), None, b.numel(), 0.0, 0.0, 0.0, 1.0, None, self.max_norm, d, first,
                                         be.stream()), "vdk_sgd_step")
        eng.refresh_weights(skip_wb16=True)
        return self.loss_rows

    def loss_scale(self) -> float:
        """GradScaler.get_scale() (one device read)"""
        return float(self.loss_state[0].item())

    def skipped_steps(self) -> int:
        return int(self.loss_state[2].item())

    def scaler_state_dict(self) -> dict:
        """`scaler.state_dict()` as the reference checkpoints it (engine/vision_engine.py:296,397): torch.cuda.amp.GradScaler's keys; {} without a scaler (bf16 / fp32
        backbone), like a disabled GradScaler"""
        if not self.amp:
            return {}
        st = self.loss_state.tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(st[1])}

    def load_scaler_state_dict(self, sd: dict) -> None:
        # (a bf16 / fp32 backbone has no scaler: a state carried over from an fp16 run or from a reference checkpoint -- scale 65536, vision_engine.py:296,397 -- is ignored,
        #  as vit.FusedTrainStep and resnet.ResNetTrainStep do; written into loss_state it would make vdk_sgd_step_amp divide unscaled gradients by 65536)
        if not sd or not self.amp:
            return
        self.growth_factor, self.backoff_factor, self.growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        self.loss_state[0] = float(sd["scale"]); self.loss_state[1] = float(sd.get("_growth_tracker", 0))

    def gather_head(self, ema: bool = False) -> torch.Tensor:
        """shard_head: all-gather the column shards (or their EMA) into `head.weight` ([D, C], every rank) for evaluation and checkpoints"""
        if not self.shard_head:
            return self.head.weight
        import torch.distributed as dist
        src = self.hs_ema if ema else self.hs
        parts = [torch.empty_like(src) for _ in range(self.comm.world_size)]
        dist.all_gather(parts, src, group=self.comm.group)
        with torch.no_grad():
            self.head.weight.copy_(torch.cat(parts, 1))
        return self.head.weight
