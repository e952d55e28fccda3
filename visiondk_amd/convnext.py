"""Host side of hot path A for CNN backbones: timm-compatible ConvNeXt (feature mode) on the native HIP engine
(csrc/convnext_engine.hip + csrc/conv.hip).

`create_model('convnext_base', pretrained=False, num_classes=0, global_pool='')` is what the reference's TimmWrapper asks timm for
(models/faceX/backbone/timm_wrapper.py:16-21; `convnext_base` in configs/faceX/cbir.yaml:4-8); forward returns the head-normed
map [B, C, H/32, W/32] like timm does.  The module tree mirrors timm's (stem.0/1, stages.i.downsample.0/1,
stages.i.blocks.j.{gamma, conv_dw, norm, mlp.fc1, mlp.fc2}, head.norm) with parameter-only holder modules, so state_dict() /
load_state_dict() carry timm's key names and reference checkpoints load unchanged.  No torch arithmetic: the NHWC -> NCHW view of
the output is a stride permutation.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.nn as nn

from . import _abi, _lib


@dataclass(frozen=True)
class ConvNeXtSpec:
    img_size: int = 224
    in_chans: int = 3
    depths: Tuple[int, int, int, int] = (3, 3, 27, 3)
    dims: Tuple[int, int, int, int] = (128, 256, 512, 1024)
    ln_eps: float = 1e-6
    num_classes: int = 0          # 0: feature mode (TimmWrapper); > 0: timm's classifier head (global average pool -> head.norm -> head.fc)


# timm 0.9.16 model ids the engine covers (dims % 8 == 0)
TIMM_CONVNEXTS = {
    "convnext_tiny": dict(depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)),
    "convnext_small": dict(depths=(3, 3, 27, 3), dims=(96, 192, 384, 768)),
    "convnext_base": dict(depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024)),
    "convnext_large": dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536)),
}


class ConvNeXtEngine:
    """Owns the flat HBM buffers (fp32 master params, bf16 / derived operand copies, grads, workspace); calls vdk_convnext_*."""

    def __init__(self, spec: ConvNeXtSpec, device=None, backend: Optional[_lib.Backend] = None, operand: str = "bf16"):
        """operand: "bf16" | "fp16" -- the format of the GEMM operands, saved 16-bit activations and 16-bit gradient tensors (VdkConvNextConfig.operand).  fp16 is the
        reference's autocast dtype on a GPU (engine/procedure/train.py:118) and the format that keeps the face / CBIR embeddings within 1e-3 of its fp32 loop (train.py:217-227);
        the train steps then run GradScaler's protocol (loss scale into the output gradient, un-scale + inf check in the optimizer kernel)."""
        if operand not in ("bf16", "fp16"):
            raise ValueError("operand must be 'bf16' or 'fp16'")
        self.spec = spec
        self.operand = operand
        self.dt16 = torch.float16 if operand == "fp16" else torch.bfloat16
        self.be = backend or _lib.load()
        self.device = torch.device(device if device is not None else ("cuda" if self.be.device_only else "cpu"))
        cfg = self._cfg(1)
        nf, nt, wx = _abi.I64(0), _abi.I32(0), C.c_size_t(0)
        self.be.check(self.be.lib.vdk_convnext_param_count(C.byref(cfg), C.byref(nf), C.byref(nt), C.byref(wx)), "vdk_convnext_param_count")
        self.n_floats, self.n_tensors = nf.value, nt.value
        self.entries = []
        name = C.create_string_buffer(96)
        off, numel, ndim = _abi.I64(0), _abi.I64(0), _abi.I32(0)
        shape = (_abi.I64 * 4)()
        for i in range(self.n_tensors):
            self.be.check(self.be.lib.vdk_convnext_param_info(C.byref(cfg), i, name, 96, C.byref(off), C.byref(numel), shape, C.byref(ndim)),
                          "vdk_convnext_param_info")
            self.entries.append((name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
        dev = self.device
        self.params = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.wb16 = torch.zeros(self.n_floats, dtype=self.dt16, device=dev)
        self.wx = torch.zeros(wx.value, dtype=torch.uint8, device=dev)
        self.out_hw = spec.img_size // 32
        self.out_ch = spec.dims[3]
        self.cp = (spec.num_classes + 7) // 8 * 8          # logits row stride in classifier mode
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = -1
        self._ws_img = spec.img_size
        self._out: Optional[torch.Tensor] = None
        self._weights_version = None
        self.buffers = torch.zeros(0, dtype=torch.float32, device=dev)      # no BatchNorm: nothing to broadcast (the train step shared with the ResNet engine asks)
        # "bf16": bf16 MFMA operands, fp32 accumulation / residual stream / master weights (the reference's autocast arithmetic, classifier loop).
        # "fp32": forward() / backward() run vdk_convnext_forward_train_f32 / vdk_convnext_backward_train_f32 -- fp32 activations, every contraction on the fp32 MFMA: the
        # arithmetic of the reference's face / CBIR loop, which has no autocast (engine/procedure/train.py:217-227).  Feature mode only.
        self.precision = "bf16"
        self._ws_t32: Optional[torch.Tensor] = None

    def _cfg(self, batch: int, img: Optional[int] = None) -> _abi.ConvNextConfig:
        s = self.spec
        return _abi.ConvNextConfig(batch, img or s.img_size, s.in_chans, (_abi.I32 * 4)(*s.depths), (_abi.I32 * 4)(*s.dims), s.ln_eps, s.num_classes,
                                   _abi.F16_ if self.operand == "fp16" else _abi.BF16)

    def _workspace(self, batch: int, img: Optional[int] = None) -> torch.Tensor:
        """keyed on (batch, image size): the classifier (global average pool head) takes any multiple of 32 -- the reference's progressive resizing
        (engine/vision_engine.py:181-222) changes the input resolution between epochs; feature mode stays at spec.img_size (the neck's Linear fixes it)"""
        img = img or self.spec.img_size
        if self._ws is None or self._ws_batch != batch or self._ws_img != img:
            need = C.c_size_t(0)
            cfg = self._cfg(batch, img)
            self._ws_img = img
            self.out_hw = img // 32
            self.be.check(self.be.lib.vdk_convnext_workspace_bytes(C.byref(cfg), C.byref(need)), "vdk_convnext_workspace_bytes")
            if self._ws is None or self._ws.numel() < need.value:      # grow-only (OHEM: a different batch size every iteration)
                self._ws = None
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
            if self.spec.num_classes > 0:
                self._out = torch.empty((batch, self.cp), dtype=torch.float32, device=self.device)
            else:
                self._out = torch.empty((batch * self.out_hw * self.out_hw, self.out_ch), dtype=torch.float32, device=self.device)
        return self._ws

    def refresh_weights(self, skip_wb16: bool = False) -> None:
        cfg = self._cfg(1)
        be = self.be
        be.check(be.lib.vdk_convnext_refresh_weights(C.byref(cfg), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wx), int(skip_wb16), be.stream()),
                 "vdk_convnext_refresh_weights")
        self._weights_version = self.params._version

    def _ensure_fresh(self) -> None:
        if self._weights_version != self.params._version:
            self.refresh_weights()

    def dlogits_rows(self, batch: int) -> int:
        """rows of the 16-bit dlogits buffer backward() expects in classifier mode (the fc weight gradient contracts over 64-row K tiles)"""
        return (batch + 63) // 64 * 64

    def forward(self, x: torch.Tensor, training: bool = True, sync_group=False) -> torch.Tensor:
        """x f32 [B, Cin, H, W] (NCHW) -> f32 [B*h*w, C] NHWC rows of the head-normed map, or (classifier mode) logits f32 [B, cp]; activations stay in
        the workspace."""
        s = self.spec
        ok = x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == s.in_chans and x.shape[2] == x.shape[3]
        if ok and s.num_classes > 0:
            ok = x.shape[2] % 32 == 0 and x.shape[2] >= 32
        elif ok:
            ok = x.shape[2] == s.img_size
        if not ok:
            want = "S, S] with S % 32 == 0" if s.num_classes > 0 else f"{s.img_size}, {s.img_size}]"
            raise ValueError(f"expected float32 [B, {s.in_chans}, {want}, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B, img = x.shape[0], x.shape[2]
        if self.precision == "fp32":
            return self._forward_train_f32(x)
        ws = self._workspace(B, img)
        self._ensure_fresh()
        cfg = self._cfg(B, img)
        be = self.be
        be.check(be.lib.vdk_convnext_forward(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wx), be.ptr(ws), ws.numel(),
                                             be.ptr(self._out), be.stream()), "vdk_convnext_forward")
        return self._out

    def _forward_train_f32(self, x: torch.Tensor) -> torch.Tensor:
        if self.spec.num_classes > 0:
            raise NotImplementedError("precision='fp32' training is built for feature mode (the face / CBIR task)")
        B = x.shape[0]
        be = self.be
        cfg = self._cfg(B)
        if self._ws_t32 is None or self._ws_batch != B:
            need = C.c_size_t(0)
            be.check(be.lib.vdk_convnext_train_f32_workspace_bytes(C.byref(cfg), C.byref(need)), "vdk_convnext_train_f32_workspace_bytes")
            if self._ws_t32 is None or self._ws_t32.numel() < need.value:
                self._ws_t32 = None
                self._ws_t32 = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._ws_batch, self._ws_img = B, self.spec.img_size
            self._out = torch.empty((B * self.out_hw * self.out_hw, self.out_ch), dtype=torch.float32, device=self.device)
        self._ensure_fresh()                      # the tap-major depthwise weights live in wx
        be.check(be.lib.vdk_convnext_forward_train_f32(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.wx), be.ptr(self._ws_t32), self._ws_t32.numel(),
                                                       be.ptr(self._out), be.stream()), "vdk_convnext_forward_train_f32")
        return self._out

    def forward_precise(self, x: torch.Tensor) -> torch.Tensor:
        """Evaluation forward with fp32 activations and fp32-MFMA contractions (vdk_convnext_forward_f32) -> f32 [B*h*w, C] NHWC rows"""
        s = self.spec
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (s.in_chans, s.img_size, s.img_size):
            raise ValueError(f"expected float32 [B, {s.in_chans}, {s.img_size}, {s.img_size}], got {tuple(x.shape)} {x.dtype}")
        if s.num_classes > 0:
            raise NotImplementedError("the fp32-MFMA evaluation forward is built for feature mode (embeddings); classifier logits use forward()")
        x = x.contiguous()
        B = x.shape[0]
        self._ensure_fresh()                      # the tap-major depthwise weights live in wx
        cfg = self._cfg(B)
        be = self.be
        need = C.c_size_t(0)
        be.check(be.lib.vdk_convnext_workspace_f32_bytes(C.byref(cfg), C.byref(need)), "vdk_convnext_workspace_f32_bytes")
        if getattr(self, "_ws32", None) is None or self._ws32.numel() < need.value:
            self._ws32 = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        out = torch.empty((B * self.out_hw * self.out_hw, self.out_ch), dtype=torch.float32, device=self.device)
        be.check(be.lib.vdk_convnext_forward_f32(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.wx), be.ptr(self._ws32), self._ws32.numel(),
                                                 be.ptr(out), be.stream()), "vdk_convnext_forward_f32")
        return out

    def backward(self, dout: torch.Tensor, on_ready: Optional[Callable[[int, int], None]] = None, sync_group=False) -> torch.Tensor:
        """dout f32 [B*h*w, C] (feature mode) or dlogits in the operand format [up(B, 64), cp] with zero padding (classifier mode) -> self.grads (flat fp32,
        overwritten).  Needs the workspace of the matching forward.  fp16 operands: dout carries the caller's loss scale, and so do the gradients."""
        if self.spec.num_classes > 0:
            assert dout.dtype == self.dt16 and dout.is_contiguous() and tuple(dout.shape) == ((self._ws_batch + 63) // 64 * 64, self.cp)
        else:
            assert dout.dtype == torch.float32 and dout.is_contiguous() and dout.shape == self._out.shape
        cfg = self._cfg(self._ws_batch, self._ws_img)
        be = self.be
        cb = _abi.GRAD_READY_FN(lambda user, off, n: on_ready(off, n)) if on_ready is not None else _abi.GRAD_READY_FN(0)
        if self.precision == "fp32":
            be.check(be.lib.vdk_convnext_backward_train_f32(C.byref(cfg), be.ptr(dout), be.ptr(self.params), be.ptr(self.wx), be.ptr(self._ws_t32), self._ws_t32.numel(),
                                                            be.ptr(self.grads), cb, None, be.stream()), "vdk_convnext_backward_train_f32")
            return self.grads
        be.check(be.lib.vdk_convnext_backward(C.byref(cfg), be.ptr(dout), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wx), be.ptr(self._ws),
                                              self._ws.numel(), be.ptr(self.grads), cb, None, be.stream()), "vdk_convnext_backward")
        return self.grads


class _ConvNeXtFunction(torch.autograd.Function):
    """model(x) as ONE autograd node: forward = vdk_convnext_forward, backward = vdk_convnext_backward."""

    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module.engine
        module._sync_flat()
        out = eng.forward(x)
        ctx.module = module
        B = x.shape[0]
        if eng.spec.num_classes > 0:
            return out[:, :eng.spec.num_classes].clone()
        # [B*h*w, C] NHWC rows -> the NCHW tensor timm returns (a strided view of a private copy; no arithmetic)
        return out.view(B, eng.out_hw, eng.out_hw, eng.out_ch).clone().permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.module.engine
        if eng.spec.num_classes > 0:
            B, n = dout.shape
            d = torch.zeros(((B + 63) // 64 * 64, eng.cp), dtype=eng.dt16, device=dout.device)
            d[:B, :n] = dout.to(eng.dt16)
        else:
            d = dout.permute(0, 2, 3, 1).contiguous().view(-1, eng.out_ch)
        g = eng.backward(d)
        return (None, None) + tuple(g[off:off + numel].view(shape) for (_, off, numel, shape) in eng.entries)


class _Holder(nn.Module):
    """empty container mirroring one level of timm's module tree; owns Parameters only"""


class ConvNeXt(nn.Module):
    """Drop-in for `timm.create_model('convnext_*', pretrained=False, num_classes=0, global_pool='')` (feature mode, TimmWrapper) and for
    `timm.create_model('convnext_*', num_classes=N)` (classifier: head = global average pool -> head.norm -> head.fc, VisionWrapper)."""

    def __init__(self, spec: ConvNeXtSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None, operand: str = "bf16"):
        super().__init__()
        self.spec = spec
        self.engine = ConvNeXtEngine(spec, device=device, backend=backend, operand=operand)
        self.num_classes = spec.num_classes
        self.num_features = spec.dims[3]
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
        """timm's own init (trunc_normal .02 Conv/Linear weights, zero biases, LayerNorm (1,0), layer scale 1e-6): the reference's
        FaceTrainingWrapper.reset_parameters is defined but never called."""
        gen = torch.Generator(device="cpu")
        # no explicit seed: drawn from torch's global generator, so torch.manual_seed(s) reproduces the initialisation as it does for the reference's model
        gen.manual_seed(seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        with torch.no_grad():
            for name, p in self._plist:
                if name.endswith(".gamma"):
                    v = torch.full(p.shape, 1e-6)
                elif ".norm." in name or name.startswith("stem.1.") or ".downsample.0." in name:
                    v = torch.ones(p.shape) if name.endswith("weight") else torch.zeros(p.shape)
                elif name.endswith(".bias"):
                    v = torch.zeros(p.shape)
                elif name == "head.fc.weight":
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen).clamp_(-2.0, 2.0)      # timm: trunc_normal_(.02) then * head_init_scale (1.0)
                else:
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen).clamp_(-2.0, 2.0)
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
                raise RuntimeError("visiondk_amd ConvNeXt keeps fp32 master weights; bf16 copies are internal")
            if self.engine.be.device_only and probe.device.type != "cuda":
                raise RuntimeError("visiondk_amd ConvNeXt lives on the GPU (no CPU fallback)")
            eng = self.engine
            eng.device = probe.device
            for attr in ("params", "grads", "wb16", "wx"):
                setattr(eng, attr, getattr(eng, attr).to(probe.device))
            eng._ws, eng._ws_batch, eng._weights_version = None, -1, None
            for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
                p.data = eng.params[off:off + numel].view(shape)
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _ConvNeXtFunction.apply(x, self, *[p for _, p in self._plist])

    @torch.no_grad()
    def forward_precise(self, x: torch.Tensor) -> torch.Tensor:
        """`model(x)` for evaluation with fp32-MFMA contractions (no autograd) -> [B, C, H/32, W/32]"""
        eng = self.engine
        self._sync_flat()
        out = eng.forward_precise(x)
        return out.view(x.shape[0], eng.out_hw, eng.out_hw, eng.out_ch).permute(0, 3, 1, 2)


def create_model(name: str, pretrained: bool = False, num_classes: int = 0, global_pool: str = "", device=None, backend=None, img_size: int = 224,
                 operand: str = "bf16", **kwargs) -> ConvNeXt:
    if name not in TIMM_CONVNEXTS:
        raise NotImplementedError(f"timm model '{name}' is not covered by the HIP engine yet (have: {sorted(TIMM_CONVNEXTS)})")
    if num_classes == 0 and global_pool != "":
        raise NotImplementedError("feature mode is built with global_pool='' (what TimmWrapper asks for)")
    if num_classes > 0 and global_pool not in ("", "avg"):
        raise NotImplementedError("the classifier head is built with timm's default global average pool")
    if pretrained:
        raise RuntimeError("pretrained weights need network access; load a checkpoint with load_state_dict instead")
    if kwargs:
        raise TypeError(f"convnext.create_model: unsupported keyword arguments {sorted(kwargs)}")
    return ConvNeXt(ConvNeXtSpec(img_size=img_size, num_classes=num_classes, **TIMM_CONVNEXTS[name]), device=device, backend=backend, operand=operand)
