"""Host side of hot path A for BatchNorm CNNs: timm-compatible BasicBlock ResNets on the native HIP engine (csrc/resnet_engine.hip).

`create_model('resnet18', pretrained=False, num_classes=C)` is what the reference's VisionWrapper asks timm for (models/classifier/classify_model.py:49-54;
`timm-resnet18` is the reference's CPU-runnable config, BASELINE.json configs[0]).  The module tree mirrors timm's (conv1, bn1, layer{1..4}.{j}.{conv1,bn1,conv2,bn2,
downsample.0/1}, fc) with parameter / buffer holder modules, so state_dict() has timm's keys in timm's order (including running_mean / running_var /
num_batches_tracked) and reference checkpoints load unchanged.  No torch arithmetic."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.nn as nn

from . import _abi, _lib


@dataclass(frozen=True)
class ResNetSpec:
    img_size: int = 224
    in_chans: int = 3
    widths: Tuple[int, int, int, int] = (64, 128, 256, 512)
    depths: Tuple[int, int, int, int] = (2, 2, 2, 2)
    num_classes: int = 1000
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    mid: Tuple[int, int, int, int] = (0, 0, 0, 0)      # > 0: Bottleneck network, inner width of the 1x1 -> 3x3 -> 1x1 blocks; `widths` = block output channels
    stem_width: int = 0                                # 0: widths[0] (BasicBlock) / 64 (Bottleneck)


_BOTT = dict(widths=(256, 512, 1024, 2048))
TIMM_RESNETS = {
    "resnet18": dict(depths=(2, 2, 2, 2)), "resnet34": dict(depths=(3, 4, 6, 3)),
    # Bottleneck family (timm / torchvision v1.5: stride on the 3x3); wide_resnet*_2: base_width 128 doubles the inner width (pet.yaml:15 lists wide_resnet101_2)
    "resnet50": dict(depths=(3, 4, 6, 3), mid=(64, 128, 256, 512), **_BOTT), "resnet101": dict(depths=(3, 4, 23, 3), mid=(64, 128, 256, 512), **_BOTT),
    "resnet152": dict(depths=(3, 8, 36, 3), mid=(64, 128, 256, 512), **_BOTT),
    "wide_resnet50_2": dict(depths=(3, 4, 6, 3), mid=(128, 256, 512, 1024), **_BOTT), "wide_resnet101_2": dict(depths=(3, 4, 23, 3), mid=(128, 256, 512, 1024), **_BOTT),
}


class ResNetEngine:
    def __init__(self, spec: ResNetSpec, device=None, backend: Optional[_lib.Backend] = None, operand: str = "bf16"):
        # operand: the 16-bit format of every convolution / fc operand, saved activation and gradient operand -- "bf16", or "fp16" = what the reference's
        # `torch.autocast(device_type=...)` (engine/procedure/train.py:118, no dtype => float16 on a GPU) computes in, with GradScaler's loss scale around the backward
        assert operand in ("bf16", "fp16"), operand
        self.spec = spec
        self.operand = operand
        self.op_dtype = torch.float16 if operand == "fp16" else torch.bfloat16
        self.be = backend or _lib.load()
        self.device = torch.device(device if device is not None else ("cuda" if self.be.device_only else "cpu"))
        cfg = self._cfg(1)
        nf, nt, nbf, nb, wx = _abi.I64(0), _abi.I32(0), _abi.I64(0), _abi.I32(0), C.c_size_t(0)
        self.be.check(self.be.lib.vdk_resnet_param_count(C.byref(cfg), C.byref(nf), C.byref(nt), C.byref(nbf), C.byref(nb), C.byref(wx)), "vdk_resnet_param_count")
        self.n_floats = nf.value
        name = C.create_string_buffer(96)
        off, numel, ndim = _abi.I64(0), _abi.I64(0), _abi.I32(0)
        shape = (_abi.I64 * 4)()

        def entries(which, n):
            out = []
            for i in range(n):
                self.be.check(self.be.lib.vdk_resnet_param_info(C.byref(cfg), which, i, name, 96, C.byref(off), C.byref(numel), shape, C.byref(ndim)),
                              "vdk_resnet_param_info")
                out.append((name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
            return out
        self.entries = entries(0, nt.value)
        self.buffer_entries = entries(1, nb.value)
        dev = self.device
        self.params = torch.zeros(nf.value, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(nf.value, dtype=torch.float32, device=dev)
        self.buffers = torch.zeros(nbf.value, dtype=torch.float32, device=dev)
        self.wb16 = torch.zeros(nf.value, dtype=self.op_dtype, device=dev)
        self.wx = torch.zeros(wx.value, dtype=torch.uint8, device=dev)
        self.cp = (spec.num_classes + 7) // 8 * 8
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = -1
        self._logits: Optional[torch.Tensor] = None
        self._weights_version = None

    def _cfg(self, batch: int, img: Optional[int] = None, bn_momentum: Optional[float] = None) -> _abi.ResNetConfig:
        s = self.spec
        return _abi.ResNetConfig(batch, img or s.img_size, s.in_chans, (_abi.I32 * 4)(*s.widths), (_abi.I32 * 4)(*s.depths), s.num_classes, s.bn_eps,
                                 s.bn_momentum if bn_momentum is None else bn_momentum, (_abi.I32 * 4)(*s.mid), s.stem_width,
                                 _abi.F16_ if self.operand == "fp16" else _abi.BF16)

    def _workspace(self, batch: int, img: int) -> torch.Tensor:
        """keyed on (batch, image size): the reference's progressive resizing (engine/vision_engine.py:181-222) changes the input resolution between
        epochs; a fully convolutional network with a global pool takes any multiple of 32"""
        if self._ws is None or self._ws_batch != (batch, img):
            need = C.c_size_t(0)
            cfg = self._cfg(batch, img)
            self.be.check(self.be.lib.vdk_resnet_workspace_bytes(C.byref(cfg), C.byref(need)), "vdk_resnet_workspace_bytes")
            if self._ws is None or self._ws.numel() < need.value:      # grow-only (OHEM: a different batch size every iteration)
                self._ws = None
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._ws_batch = (batch, img)
            self._logits = torch.empty((batch, self.cp), dtype=torch.float32, device=self.device)
        return self._ws

    def refresh_weights(self, skip_wb16: bool = False) -> None:
        cfg = self._cfg(1)
        be = self.be
        be.check(be.lib.vdk_resnet_refresh_weights(C.byref(cfg), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wx), int(skip_wb16), be.stream()),
                 "vdk_resnet_refresh_weights")
        self._weights_version = self.params._version

    def _sync_cb(self, group):
        """vdk_stat_sync_fn for SyncBatchNorm: all-reduce (SUM) the statistics vector the engine hands over; it lives inside the workspace tensor"""
        if group is False:
            return _abi.STAT_SYNC_FN(0)
        import torch.distributed as dist

        def cb(user, ptr, n):
            off = ptr - self._ws.data_ptr()
            dist.all_reduce(self._ws[off:off + 4 * n].view(torch.float32), op=dist.ReduceOp.SUM, group=group)
        return _abi.STAT_SYNC_FN(cb)

    def forward(self, x: torch.Tensor, training: bool, sync_group=False, bn_momentum: Optional[float] = None) -> torch.Tensor:
        """sync_group: False = per-rank BatchNorm statistics; None or a process group = SyncBatchNorm over that group.
        bn_momentum: override of the running-statistics momentum for this pass (SAM's second pass runs with 0, optimizer.py:92-98)"""
        s = self.spec
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != s.in_chans or x.shape[2] != x.shape[3] or x.shape[2] % 32:
            raise ValueError(f"expected float32 [B, {s.in_chans}, S, S] with S % 32 == 0, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B, img = x.shape[0], x.shape[2]
        ws = self._workspace(B, img)
        if self._weights_version != self.params._version:
            self.refresh_weights()
        cfg = self._cfg(B, img, bn_momentum)
        be = self.be
        cb = self._sync_cb(sync_group)
        be.check(be.lib.vdk_resnet_forward(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.buffers), be.ptr(self.wb16), be.ptr(self.wx), int(training),
                                           be.ptr(ws), ws.numel(), be.ptr(self._logits), cb, None, be.stream()), "vdk_resnet_forward")
        return self._logits

    def backward(self, dlogits_bf16: torch.Tensor, on_ready: Optional[Callable[[int, int], None]] = None, sync_group=False) -> torch.Tensor:
        B = dlogits_bf16.shape[0]
        assert dlogits_bf16.dtype == self.op_dtype and dlogits_bf16.shape[1] == self.cp and B == self._ws_batch[0]      # (in the engine's operand format)
        cfg = self._cfg(B, self._ws_batch[1])
        be = self.be
        cb = _abi.GRAD_READY_FN(lambda user, off, n: on_ready(off, n)) if on_ready is not None else _abi.GRAD_READY_FN(0)
        be.check(be.lib.vdk_resnet_backward(C.byref(cfg), be.ptr(dlogits_bf16), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wx), be.ptr(self._ws),
                                            self._ws.numel(), be.ptr(self.grads), cb, None, self._sync_cb(sync_group), None, be.stream()), "vdk_resnet_backward")
        return self.grads


class _ResNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module.engine
        module._sync_flat()
        logits = eng.forward(x, module.training)
        ctx.module = module
        return logits[:, :eng.spec.num_classes].clone()

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.module.engine
        be = eng.be
        B, Cn = dlogits.shape
        stage = torch.zeros((B, eng.cp), dtype=torch.float32, device=dlogits.device)
        stage[:, :Cn].copy_(dlogits)
        dl = torch.empty((B, eng.cp), dtype=eng.op_dtype, device=dlogits.device)
        cast = be.lib.vdk_cast_f32_f16 if eng.operand == "fp16" else be.lib.vdk_cast_f32_bf16      # (fp16: the incoming gradient carries GradScaler's loss scale, train.py:205)
        be.check(cast(be.ptr(stage), be.ptr(dl), stage.numel(), be.stream()), "vdk_cast_f32_16")
        g = eng.backward(dl)
        return (None, None) + tuple(g[off:off + numel].view(shape) for (_, off, numel, shape) in eng.entries)


class _Holder(nn.Module):
    """empty container mirroring one level of timm's module tree; owns Parameters / buffers only"""


class ResNet(nn.Module):
    """Drop-in for `timm.create_model('resnet18' | 'resnet34', pretrained=False, num_classes=C)`."""

    def __init__(self, spec: ResNetSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None, operand: str = "bf16"):
        super().__init__()
        self.spec = spec
        self.engine = ResNetEngine(spec, device=device, backend=backend, operand=operand)
        self.num_classes = spec.num_classes
        eng = self.engine
        self._plist, self._blist = [], []
        bufs = {n: (off, numel, shape) for n, off, numel, shape in eng.buffer_entries}

        def holder(path):
            m = self
            for part in path:
                if part not in m._modules:
                    m.add_module(part, _Holder())
                m = m._modules[part]
            return m
        for name, off, numel, shape in eng.entries:
            parts = name.split(".")
            m = holder(parts[:-1])
            p = nn.Parameter(eng.params[off:off + numel].view(shape))
            m.register_parameter(parts[-1], p)
            self._plist.append((name, p))
            if parts[-1] == "bias" and ".".join(parts[:-1]) + ".running_mean" in bufs:      # a BatchNorm: its buffers follow weight / bias, like nn.BatchNorm2d
                for bn in ("running_mean", "running_var"):
                    boff, bnum, bshape = bufs[".".join(parts[:-1]) + "." + bn]
                    t = eng.buffers[boff:boff + bnum].view(bshape)
                    m.register_buffer(bn, t)
                    self._blist.append((".".join(parts[:-1]) + "." + bn, boff, bnum, bshape, m, bn))
                m.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long, device=eng.device))
        self.reset_parameters(seed)

    def reset_parameters(self, seed: Optional[int] = None) -> None:
        """the reference's re-init (classify_model.py:70-81): N(0, .02) Conv / Linear weights, zero Linear bias, BatchNorm (1, 0); running stats (0, 1)"""
        gen = torch.Generator(device="cpu")
        # no explicit seed: drawn from torch's global generator, so torch.manual_seed(s) reproduces the initialisation as it does for the reference's model
        gen.manual_seed(seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        with torch.no_grad():
            for name, p in self._plist:
                if name.endswith(".bias"):
                    v = torch.zeros(p.shape)
                elif p.dim() == 1:
                    v = torch.ones(p.shape)
                else:
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen)
                p.copy_(v.to(p.device))
            for name, off, numel, shape, m, bn in self._blist:
                getattr(m, bn).fill_(1.0 if bn == "running_var" else 0.0)

    def _sync_flat(self) -> None:
        eng = self.engine
        base = eng.params.data_ptr()
        for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
            if p.data_ptr() != base + off * 4:
                with torch.no_grad():
                    eng.params[off:off + numel].view(shape).copy_(p.detach().to(eng.device))
                    p.data = eng.params[off:off + numel].view(shape)
        bbase = eng.buffers.data_ptr()
        for name, off, numel, shape, m, bn in self._blist:
            t = getattr(m, bn)
            if t.data_ptr() != bbase + off * 4:       # load_state_dict copies in place, so this only triggers after .to() / deepcopy
                with torch.no_grad():
                    eng.buffers[off:off + numel].view(shape).copy_(t.to(eng.device))
                    m._buffers[bn] = eng.buffers[off:off + numel].view(shape)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self.engine.device))
        if probe.device != self.engine.device or probe.dtype != torch.float32:
            if probe.dtype != torch.float32:
                raise RuntimeError("visiondk_amd ResNet keeps fp32 master weights; bf16 copies are internal")
            if self.engine.be.device_only and probe.device.type != "cuda":
                raise RuntimeError("visiondk_amd ResNet lives on the GPU (no CPU fallback)")
            eng = self.engine
            eng.device = probe.device
            for attr in ("params", "grads", "buffers", "wb16", "wx"):
                setattr(eng, attr, getattr(eng, attr).to(probe.device))
            eng._ws, eng._ws_batch, eng._weights_version = None, -1, None
            for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
                p.data = eng.params[off:off + numel].view(shape)
            for name, off, numel, shape, m, bn in self._blist:
                m._buffers[bn] = eng.buffers[off:off + numel].view(shape)
                m._buffers["num_batches_tracked"] = m._buffers["num_batches_tracked"].to(probe.device)
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            for m in self.modules():
                if "num_batches_tracked" in m._buffers:
                    m._buffers["num_batches_tracked"] += 1
        if torch.is_grad_enabled():
            return _ResNetFunction.apply(x, self, *[p for _, p in self._plist])
        self._sync_flat()
        return self.engine.forward(x, self.training)[:, :self.spec.num_classes].clone()


def create_model(name: str, pretrained: bool = False, num_classes: int = 1000, device=None, backend=None, img_size: Optional[int] = None, operand: str = "bf16",
                 **kwargs) -> ResNet:
    if name not in TIMM_RESNETS:
        raise NotImplementedError(f"timm model '{name}' is not covered by the HIP engine yet (have: {sorted(TIMM_RESNETS)})")
    if pretrained:
        raise RuntimeError("pretrained weights need network access; load a checkpoint with load_state_dict instead")
    return ResNet(ResNetSpec(img_size=img_size or 224, num_classes=num_classes, **TIMM_RESNETS[name]), device=device, backend=backend, operand=operand)


class ResNetTrainStep:
    """Trainer.compute_loss + Trainer.update (engine/procedure/train.py:177-215) for a CNN classifier -- the ResNet engine, or the ConvNeXt engine in
    classifier mode (`ClassifierTrainStep` is the same class) -- as a fixed kernel sequence: forward ->
    BCE-with-logits (multi-label CSV datasets, checks.py:163-167; mean over B*C like nn.BCEWithLogitsLoss) or CE(label_smoothing) -> backward ->
    clip_grad_norm_(max_norm) -> SGD(momentum, weight_decay) -> EMA -> bf16 weight refresh.  `param_groups[0]['lr']` stays readable / writable."""

    def __init__(self, model: ResNet, lr: float, momentum: float = 0.937, weight_decay: float = 5e-4, loss: str = "bce", label_smoothing: float = 0.0,
                 max_norm: float = 10.0, ema: bool = True, comm=None, sync_bn: bool = False, graph: bool = False, sam: bool = False, sam_rho: float = 0.05,
                 sam_adaptive: bool = True, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        """An engine with fp16 operands (convnext.create_model(..., operand="fp16"): the reference's autocast dtype on a GPU, train.py:118) runs the step under the
        GradScaler protocol of Trainer.update (train.py:203-215), as vit.FusedTrainStep does: the loss scale lives on the device ({scale, growth tracker, skipped steps};
        init_scale / growth_factor / backoff_factor / growth_interval = torch's defaults), the loss kernel multiplies it into dlogits, vdk_sgd_step_amp un-scales, checks for
        inf / NaN and skips the update while the EMA still moves, vdk_loss_scale_update grows / backs off.
        sam: Trainer.update_sam (train.py:150-175) over engine/optimizer.py's SAM(base SGD): forward-backward at w with LOCAL gradients, climb to w + e(w),
        forward-backward there with the BatchNorm running statistics frozen (momentum 0) and the gradients all-reduced, back to w, base SGD step
        (no clipping on this path), EMA; the returned loss is the first pass's.
        graph: capture the whole step (about 250 launches of a few microseconds each at ResNet-18 / bs 32: launch-bound) once per batch shape in a hipGraph
        (torch.cuda.CUDAGraph) and replay it; the per-step scalars (lr, momentum, weight decay, EMA decay, first-step flag) travel through a 5-float
        device vector (vdk_sgd_step_graph).  Single-process only; results are bit-identical to the eager sequence.
        sync_bn: SyncBatchNorm over comm's group (the reference's `sync_bn` flag -> nn.SyncBatchNorm.convert_sync_batchnorm, vision_engine.py:224-225).
        comm: visiondk_amd.comm.GradAllReduce (one process per GPU): weights and BatchNorm buffers broadcast from rank 0 at construction, buffers again
        before every forward (torch DDP's broadcast_buffers), the flat gradient all-reduced in buckets from inside vdk_resnet_backward; BatchNorm statistics
        stay per rank (SyncBN is the reference's opt-in flag and is not built)."""
        assert loss in ("bce", "ce")
        self.amp = getattr(model.engine, "operand", "bf16") == "fp16"
        if self.amp and (sam or graph):
            raise NotImplementedError("fp16 operands (GradScaler protocol) are built for the eager SGD step; SAM / graph replay run on bf16 operands")
        self.dt16 = torch.float16 if self.amp else torch.bfloat16
        self.loss_state = torch.tensor([init_scale, 0.0, 0.0], dtype=torch.float32, device=model.engine.device) if self.amp else None
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.comm = comm
        self.sync_group = (comm.group if (sync_bn and comm is not None and comm.active) else False)
        self.model, self.eng, self.be = model, model.engine, model.engine.be
        self.loss, self.label_smoothing, self.max_norm = loss, label_smoothing, max_norm
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        self.momentum_buf = torch.zeros_like(self.eng.params)
        self.ema = self.eng.params.clone() if ema else None
        # ModelEMA averages EVERY floating entry of state_dict() (models/ema.py:28-37): the BatchNorm running statistics too, and the reference validates and
        # checkpoints from that copy -- an EMA of the weights alone would pair lagged weights with live statistics
        has_buf = ema and getattr(self.eng, "buffers", None) is not None and self.eng.buffers.numel() > 0     # the ConvNeXt classifier has no BatchNorm
        self.ema_buffers = self.eng.buffers.clone() if has_buf else None
        self._zero_b = torch.zeros_like(self.eng.buffers) if has_buf else None       # gradient / momentum stand-ins of the lr = 0 pass that carries the EMA
        self._zero_bm = torch.zeros_like(self.eng.buffers) if has_buf else None
        self.updates = 0
        self._normsq = torch.zeros(1, dtype=torch.float32, device=self.eng.device)
        need = C.c_size_t(0)
        self.be.check(self.be.lib.vdk_sumsq_workspace_bytes(C.byref(need)), "vdk_sumsq_workspace_bytes")
        self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.eng.device)
        self.loss_rows: Optional[torch.Tensor] = None
        if graph and comm is not None and comm.active:
            raise NotImplementedError("graph capture of the data-parallel step (host callbacks issue the collectives) is not built")
        self.graph = graph
        self.focal_gamma, self.focal_alpha = 0.0, 0.25          # set_focal(): the reference switches BCE -> focal at `strategy.focal[0]` epochs (vision_engine.py:160,367-368)
        self.sam, self.sam_rho, self.sam_adaptive = sam, sam_rho, sam_adaptive
        self._old_params = torch.empty_like(self.eng.params) if sam else None
        self._graphs = {}                       # (B, y shape/dtype) -> (CUDAGraph, static x, static y)
        self._hyper = torch.zeros(10, dtype=torch.float32, device=self.eng.device) if graph else None   # [0:5] the step's scalars, [5:10] the same with lr = momentum = wd = 0 (EMA of the buffers)
        self._hyper_ring = [(torch.zeros(10, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)] if graph else []
        if comm is not None and comm.active:
            import torch.distributed as dist
            comm.broadcast_params(self.eng.params, engine=self.eng)
            dist.broadcast(self.eng.buffers, src=0, group=comm.group)
            if ema:
                self.ema.copy_(self.eng.params)
                if self.ema_buffers is not None:
                    self.ema_buffers.copy_(self.eng.buffers)

    def set_focal(self, gamma: float = 2.0, alpha: float = 0.25) -> None:
        """Trainer's focal switch (engine/vision_engine.py:367-368): from now on the BCE loss is the reference's focal loss (models/losses/loss.py:27-54,75);
        gamma = 0 switches back.  A captured graph holds the old values by value: graphs are dropped."""
        if self.loss != "bce":
            raise ValueError("the reference's focal loss replaces the BCE loss (multi-label datasets)")
        self.focal_gamma, self.focal_alpha = float(gamma), float(alpha)
        self._graphs = {}

    def ohem_select(self, x: torch.Tensor, y: torch.Tensor, min_kept: int, thresh: float, ignore_index: int = 255):
        """OHEM-Softmax pre-pass (engine/procedure/train.py:113-117, structure/sampler.py:11-31): one extra no-grad forward in TRAINING mode, as the reference
        runs it (BatchNorm running statistics move), keep the samples whose target probability is below max(thresh, the min_kept-th smallest).
        Single-label (class-index) targets only, like the sampler's `gather`."""
        from . import ops
        self.model._sync_flat()
        self.model.train()
        for m in self.model.modules():
            if "num_batches_tracked" in m._buffers:
                m._buffers["num_batches_tracked"] += 1
        logits = self.eng.forward(x, True, sync_group=self.sync_group)
        mask = ops.ohem_mask(logits[:, :self.eng.spec.num_classes].contiguous(), y, min_kept, thresh, ignore_index, backend=self.be)
        keep = mask.nonzero().squeeze(1)
        return x[keep].contiguous(), y[keep].contiguous()

    def _dl_rows(self, B: int) -> int:
        return self.eng.dlogits_rows(B) if hasattr(self.eng, "dlogits_rows") else B

    def step(self, x: torch.Tensor, y: torch.Tensor, y_b: Optional[torch.Tensor] = None, lam: float = 1.0) -> torch.Tensor:
        """y_b, lam: mixup_criterion (train.py:34-35): lam * loss(pred, y) + (1 - lam) * loss(pred, y_b).  CE takes the pair as is; BCE-with-logits is
        linear in its targets, so the pair is folded into one soft target."""
        if y_b is not None and self.loss == "bce":
            if self.focal_gamma > 0.0:
                # focal(pred, t) is NOT linear in t (alpha_t and (1 - p_t)^gamma depend on it): the pair cannot be folded into one soft target.
                # The reference's mixup_criterion evaluates the loss twice; so does the kernel sequence (second pass accumulates), eager mode only.
                if self.graph:
                    raise NotImplementedError("graph replay of focal loss with a mixup pair")
                return self._step_eager(x, y, None, y_b=y_b, lam=lam)
            y, y_b = lam * y + (1.0 - lam) * y_b, None
        if self.graph:
            if y_b is not None:
                raise NotImplementedError("graph replay with a CE mixup pair (lam is a by-value kernel argument)")
            return self._step_graph(x, y)
        return self._step_eager(x, y, None, y_b=y_b, lam=lam)

    def _step_graph(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        import math
        model = self.model
        model._sync_flat()
        model.train()
        self.updates += 1
        g = self.param_groups[0]
        d = 0.9999 * (1 - math.exp(-self.updates / 2000)) if self.ema is not None else 0.0
        for m in model.modules():
            if "num_batches_tracked" in m._buffers:
                m._buffers["num_batches_tracked"] += 1
        host, ev = self._hyper_ring[self.updates % len(self._hyper_ring)]     # pinned staging slots: a slot is rewritten only after its copy has run
        ev.synchronize()
        first = 1.0 if self.updates == 1 else 0.0
        host.copy_(torch.tensor([g["lr"], g["momentum"], g["weight_decay"], d, first, 0.0, 0.0, 0.0, d, first]))
        self._hyper.copy_(host, non_blocking=True)
        ev.record()
        key = (tuple(x.shape), tuple(y.shape), y.dtype)
        if key not in self._graphs:
            sx, sy = x.clone(), y.clone()
            eng = self.eng
            if isinstance(eng, ResNetEngine):                                # allocations and the weight refresh happen before the capture, not inside it
                eng._workspace(x.shape[0], x.shape[-1])
            else:
                eng._workspace(x.shape[0])
            if eng._weights_version != eng.params._version:
                eng.refresh_weights()
            if self.loss_rows is None or self.loss_rows.shape[0] != x.shape[0]:
                self.loss_rows = torch.empty(x.shape[0], dtype=torch.float32, device=self.eng.device)
                self._dl = torch.zeros((self._dl_rows(x.shape[0]), self.eng.cp), dtype=torch.bfloat16, device=self.eng.device)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                self._step_eager(sx, sy, self._hyper, count=False)
            self._graphs[key] = (gr, sx, sy, self.loss_rows)
        gr, sx, sy, rows = self._graphs[key]
        sx.copy_(x, non_blocking=True)
        sy.copy_(y, non_blocking=True)
        gr.replay()
        self.loss_rows = rows
        return rows

    def _step_eager(self, x: torch.Tensor, y: torch.Tensor, hyper, count: bool = True, y_b: Optional[torch.Tensor] = None, lam: float = 1.0) -> torch.Tensor:
        import math
        eng, be, model = self.eng, self.be, self.model
        world = self.comm.world_size if self.comm is not None else 1
        active = self.comm is not None and self.comm.active
        if active:
            import torch.distributed as dist
            dist.broadcast(eng.buffers, src=0, group=self.comm.group)
        if count:
            model._sync_flat()
            model.train()
            self.updates += 1
            for m in model.modules():
                if "num_batches_tracked" in m._buffers:
                    m._buffers["num_batches_tracked"] += 1
        g = self.param_groups[0]
        B, ncls = x.shape[0], eng.spec.num_classes
        if self.loss_rows is None or self.loss_rows.shape[0] != B:
            self.loss_rows = torch.empty(B, dtype=torch.float32, device=eng.device)
            self._dl = torch.zeros((self._dl_rows(B), eng.cp), dtype=self.dt16, device=eng.device)
        ls = be.ptr(self.loss_state) if self.amp else None                    # the device loss scale the loss kernels multiply into dlogits (GradScaler.scale(loss), train.py:205)
        dlt = _abi.F16_ if self.amp else _abi.BF16

        def fwd_loss_bwd(sync: bool, bn_momentum=None):
            kw = {} if bn_momentum is None else {"bn_momentum": bn_momentum}
            logits = eng.forward(x, True, sync_group=self.sync_group, **kw)
            if self.loss == "bce" and y_b is not None:
                # focal loss with a mixup pair: lam * focal(pred, y) + (1 - lam) * focal(pred, y_b) (mixup_criterion, train.py:34-35) -- two evaluations, gradients added in fp32
                rows2 = torch.empty_like(self.loss_rows)
                g1 = torch.zeros((self._dl.shape[0], eng.cp), dtype=torch.float32, device=eng.device)
                g2 = torch.zeros_like(g1)
                for tgt, wgt, rows, gg in ((y, lam, self.loss_rows, g1), (y_b, 1.0 - lam, rows2, g2)):
                    be.check(be.lib.vdk_bce_logits(be.ptr(logits), eng.cp, be.ptr(tgt), tgt.stride(0), B, ncls, wgt / (B * ncls), self.focal_gamma, self.focal_alpha, be.ptr(rows),
                                                   None, 0, be.ptr(gg), eng.cp, be.stream()), "vdk_bce_logits")
                self.loss_rows.mul_(lam).add_(rows2, alpha=1.0 - lam)
                g1.add_(g2)
                if self.amp:
                    be.check(be.lib.vdk_scale_dev_f32(be.ptr(g1), g1.numel(), ls, 0, be.stream()), "vdk_scale_dev_f32")
                    be.check(be.lib.vdk_cast_f32_f16(be.ptr(g1), be.ptr(self._dl), g1.numel(), be.stream()), "vdk_cast_f32_f16")
                else:
                    be.check(be.lib.vdk_cast_f32_bf16(be.ptr(g1), be.ptr(self._dl), g1.numel(), be.stream()), "vdk_cast_f32_bf16")
            elif self.loss == "bce":
                be.check(be.lib.vdk_bce_logits_amp(be.ptr(logits), eng.cp, be.ptr(y), y.stride(0), B, ncls, 1.0 / (B * ncls), ls, self.focal_gamma, self.focal_alpha,
                                                   be.ptr(self.loss_rows), be.ptr(self._dl), eng.cp, dlt, None, 0, be.stream()), "vdk_bce_logits")
            else:
                be.check(be.lib.vdk_softmax_ce_amp(be.ptr(logits), eng.cp, B, ncls, be.ptr(y), be.ptr(y_b), lam, self.label_smoothing, 1.0 / B, ls, be.ptr(self.loss_rows),
                                                   be.ptr(self._dl), eng.cp, dlt, None, 0, be.stream()), "vdk_softmax_ce")
            if active and sync:
                self.comm.begin_step(eng.grads)
                eng.backward(self._dl, on_ready=self.comm.on_grad_ready, sync_group=self.sync_group)
                self.comm.finish_step()
            else:
                eng.backward(self._dl, sync_group=self.sync_group)

        if self.sam:
            fwd_loss_bwd(sync=False)                                          # model.no_sync() in the reference: local gradients
            loss_first = self.loss_rows.clone()
            be.check(be.lib.vdk_sam_first_step(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self._old_params), eng.n_floats, self.sam_rho, int(self.sam_adaptive),
                                               be.ptr(self._normsq), be.ptr(self._ws), self._ws.numel(), be.stream()), "vdk_sam_first_step")
            eng.refresh_weights()
            for m in model.modules():                                         # torch's BatchNorm counts every training forward, whatever the momentum
                if "num_batches_tracked" in m._buffers:
                    m._buffers["num_batches_tracked"] += 1
            fwd_loss_bwd(sync=True, bn_momentum=0.0 if hasattr(eng, "spec") and hasattr(eng.spec, "bn_momentum") else None)
            eng.params.copy_(self._old_params)
            self.loss_rows.copy_(loss_first)                                  # update_sam returns the FIRST loss (train.py:175)
            nsq = None                                                        # no clipping on the SAM path
        else:
            fwd_loss_bwd(sync=True)
            be.check(be.lib.vdk_sumsq_f32(be.ptr(eng.grads), eng.n_floats, be.ptr(self._normsq), be.ptr(self._ws), self._ws.numel(), be.stream()), "vdk_sumsq_f32")
            nsq = be.ptr(self._normsq)
        if hyper is not None:
            be.check(be.lib.vdk_sgd_step_graph(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), eng.n_floats,
                                               be.ptr(hyper), 1.0 / world, nsq, self.max_norm, be.stream()), "vdk_sgd_step_graph")
            if self.ema_buffers is not None:
                be.check(be.lib.vdk_sgd_step_graph(be.ptr(eng.buffers), be.ptr(self._zero_b), be.ptr(self._zero_bm), be.ptr(self.ema_buffers), None, eng.buffers.numel(),
                                                   be.ptr(hyper[5:]), 1.0, None, self.max_norm, be.stream()), "vdk_sgd_step_graph")
        else:
            d = 0.9999 * (1 - math.exp(-self.updates / 2000)) if self.ema is not None else 0.0
            if self.amp:      # scaler.unscale_ + clip + scaler.step (skipped on inf / NaN) + ema.update, then scaler.update (train.py:205-215)
                be.check(be.lib.vdk_sgd_step_amp(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), _abi.F16_, eng.n_floats,
                                                 g["lr"], g["momentum"], g["weight_decay"], 1.0 / world, ls, nsq, self.max_norm, d, int(self.updates == 1), be.stream()),
                         "vdk_sgd_step_amp")
                be.check(be.lib.vdk_loss_scale_update(ls, nsq, self.growth_factor, self.backoff_factor, self.growth_interval, be.stream()), "vdk_loss_scale_update")
            else:
                be.check(be.lib.vdk_sgd_step(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), eng.n_floats, g["lr"],
                                             g["momentum"], g["weight_decay"], 1.0 / world, nsq, self.max_norm, d, int(self.updates == 1), be.stream()),
                         "vdk_sgd_step")
            if self.ema_buffers is not None:      # lr = 0: the statistics themselves are untouched, their EMA moves with the same decay
                be.check(be.lib.vdk_sgd_step(be.ptr(eng.buffers), be.ptr(self._zero_b), be.ptr(self._zero_bm), be.ptr(self.ema_buffers), None, eng.buffers.numel(), 0.0, 0.0, 0.0,
                                             1.0, None, self.max_norm, d, int(self.updates == 1), be.stream()), "vdk_sgd_step")
        eng.refresh_weights(skip_wb16=True)
        return self.loss_rows


    def loss_scale(self) -> float:
        return float(self.loss_state[0].item()) if self.amp else 1.0

    def skipped_steps(self) -> int:
        return int(self.loss_state[2].item()) if self.amp else 0

    def scaler_state_dict(self) -> dict:
        """`scaler.state_dict()` as the reference checkpoints it (engine/vision_engine.py:296,397); {} without a scaler (bf16 operands), like a disabled GradScaler"""
        if not self.amp:
            return {}
        st = self.loss_state.tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(st[1])}

    def load_scaler_state_dict(self, sd: dict) -> None:
        if not sd or not self.amp:
            return
        self.growth_factor, self.backoff_factor, self.growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        self.loss_state[0] = float(sd["scale"]); self.loss_state[1] = float(sd.get("_growth_tracker", 0))


ClassifierTrainStep = ResNetTrainStep      # engine-agnostic: needs forward(x, training) -> logits f32 [B, cp], backward(dlogits bf16), flat params / grads / wb16
