"""Host side of hot path A: timm-compatible VisionTransformer on the native HIP engine (csrc/vit_engine.hip).

Three layers, thinnest first:
  * `VitEngine`      — owns the flat HBM buffers (fp32 master params, bf16 operand copies, grads, workspace) and
                       calls vdk_vit_forward / vdk_vit_backward.  No torch arithmetic.
  * `VisionTransformer(nn.Module)` — what `timm.create_model('vit_*', num_classes=C)` returns to the reference
                       (models/classifier/classify_model.py:49-54): parameters with timm's state_dict names (views into
                       the flat buffer), `model(x) -> logits` differentiable through ONE autograd node, so the
                       reference's Trainer (engine/procedure/train.py:177-215) drives it unchanged.
  * `FusedTrainStep` — the whole of Trainer.compute_loss + Trainer.update as a fixed kernel sequence: forward, CE,
                       backward, [bucketed gradient all-reduce], global-norm clip + SGD + EMA + bf16 refresh.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.nn as nn

from . import _abi, _lib, ops


@dataclass(frozen=True)
class VitSpec:
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 1000
    dim: int = 768
    depth: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    ln_eps: float = 1e-6
    class_token: bool = True      # timm class_token=False (the SigLIP ViTs): tokens are the patches only; feature mode (num_classes=0) in the engine
    pre_norm: bool = False        # timm pre_norm=True (the CLIP ViTs): LayerNorm `norm_pre` in front of the blocks, patch embedding without a bias


# timm 0.9.16 model ids the engine covers (head_dim 64; patch 14 works through zero-padded operand copies of the patch-embedding weight)
TIMM_VITS = {
    "vit_tiny_patch16_224": dict(dim=192, depth=12, heads=3, mlp_dim=768),
    "vit_small_patch16_224": dict(dim=384, depth=12, heads=6, mlp_dim=1536),
    "vit_base_patch16_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072),
    "vit_base_patch32_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072, patch_size=32),
    # 28 x 28 + 1 = 785 tokens: the streaming attention kernels (csrc/attention_long.hip); `vit_base_patch8_224.dino` is an alternative listed in the shipped YAMLs (cbir.yaml:8)
    "vit_small_patch8_224": dict(dim=384, depth=12, heads=6, mlp_dim=1536, patch_size=8),
    "vit_base_patch8_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072, patch_size=8),
    "vit_large_patch16_224": dict(dim=1024, depth=24, heads=16, mlp_dim=4096),
    "vit_base_patch16_384": dict(dim=768, depth=12, heads=12, mlp_dim=3072, img_size=384),
    "vit_large_patch14_224": dict(dim=1024, depth=24, heads=16, mlp_dim=4096, patch_size=14),
    # the CLIP ViTs (pre_norm=True, LayerNorm eps 1e-5; GELU -- the quick-GELU ids are separate models); `vit_base_patch16_clip_224.laion2b_ft_in1k` is listed in the shipped YAMLs
    "vit_base_patch32_clip_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072, patch_size=32, pre_norm=True, ln_eps=1e-5),
    "vit_base_patch16_clip_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072, pre_norm=True, ln_eps=1e-5),
    "vit_large_patch14_clip_224": dict(dim=1024, depth=24, heads=16, mlp_dim=4096, patch_size=14, pre_norm=True, ln_eps=1e-5),
    "vit_large_patch14_clip_336": dict(dim=1024, depth=24, heads=16, mlp_dim=4096, patch_size=14, img_size=336, pre_norm=True, ln_eps=1e-5),
    # class_token=False + global_pool='map' (SigLIP): served by VisionTransformerMap
    "vit_base_patch16_siglip_224": dict(dim=768, depth=12, heads=12, mlp_dim=3072, class_token=False),
    "vit_large_patch16_siglip_256": dict(dim=1024, depth=24, heads=16, mlp_dim=4096, img_size=256, class_token=False),
    "vit_large_patch14_siglip_336": dict(dim=1024, depth=24, heads=16, mlp_dim=4096, patch_size=14, img_size=336, class_token=False),   # BASELINE.json configs[4]'s geometry
}


def spec_from_timm_name(name: str, num_classes: int, img_size: Optional[int] = None) -> VitSpec:
    if name not in TIMM_VITS:
        raise NotImplementedError(f"timm model '{name}' is not covered by the HIP engine yet (have: {sorted(TIMM_VITS)})")
    kw = dict(TIMM_VITS[name])
    if img_size is not None:
        kw["img_size"] = img_size
    return VitSpec(num_classes=num_classes, **kw)


OPERANDS = {"bf16": torch.bfloat16, "fp16": torch.float16}


class VitEngine:
    def __init__(self, spec: VitSpec, device=None, backend: Optional[_lib.Backend] = None, operand: str = "bf16"):
        self.spec = spec
        self.be = backend or _lib.load()
        self.device = torch.device(device if device is not None else ("cuda" if self.be.device_only else "cpu"))
        # 16-bit format of the GEMM operands / saved activations / gradient tensors: "bf16" (BASELINE.json configs[1]) or "fp16" -- what the reference's
        # `torch.autocast(device_type=...)` (engine/procedure/train.py:118, no dtype => float16 on a GPU) computes in; see set_operand()
        assert operand in OPERANDS, operand
        self.operand = operand
        self.fp8 = 0                     # 0 = 16-bit operands; see enable_fp8
        self.fp8_w: Optional[torch.Tensor] = None
        self.fp8_state: Optional[torch.Tensor] = None
        cfg = self._cfg(1)
        nf, nt, ntr = _abi.I64(0), _abi.I32(0), _abi.I64(0)
        self.be.check(self.be.lib.vdk_vit_param_count(C.byref(cfg), C.byref(nf), C.byref(nt), C.byref(ntr)), "vdk_vit_param_count")
        self.n_floats, self.n_tensors, self.n_transposed = nf.value, nt.value, ntr.value
        self.entries = []
        name = C.create_string_buffer(96)
        off, numel, ndim = _abi.I64(0), _abi.I64(0), _abi.I32(0)
        shape = (_abi.I64 * 4)()
        for i in range(self.n_tensors):
            self.be.check(self.be.lib.vdk_vit_param_info(C.byref(cfg), i, name, 96, C.byref(off), C.byref(numel), shape, C.byref(ndim)),
                          "vdk_vit_param_info")
            self.entries.append((name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
        dev = self.device
        self.params = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.wb16 = torch.zeros(self.n_floats, dtype=self.op_dtype, device=dev)
        self.wt16 = torch.zeros(self.n_transposed, dtype=self.op_dtype, device=dev)
        self.cp = (spec.num_classes + 7) // 8 * 8          # 0 in feature mode (num_classes == 0)
        self.tokens = (spec.img_size // spec.patch_size) ** 2 + (1 if spec.class_token else 0)
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = -1
        self._logits: Optional[torch.Tensor] = None
        self._weights_version = None
        # optional side stream for the weight-gradient GEMMs of vdk_vit_backward.  Measured on MI355X (ViT-B/16 bs256): no gain
        # (49.4 ms/step either way; one 128-KB-LDS workgroup per CU means the two kernels only time-share CUs), so it is off
        # unless VDK_SIDE_STREAM=1.
        import os
        self.side_stream = (torch.cuda.Stream(device=self.device)
                            if (os.environ.get("VDK_SIDE_STREAM") == "1" and self.be.device_only and self.device.type == "cuda") else None)

    def __deepcopy__(self, memo):   # ModelEMA deep-copies the model; streams and library handles are not copyable
        import copy
        new = object.__new__(VitEngine)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "side_stream":
                new.side_stream = torch.cuda.Stream(device=self.device) if v is not None else None
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    # ---- plumbing ------------------------------------------------------------------------------
    @property
    def op_dtype(self) -> torch.dtype:
        return OPERANDS[self.operand]

    def set_operand(self, operand: str) -> None:
        """switch the engine between bf16 and fp16 operands (weights stay the fp32 master copy; the 16-bit copies are rebuilt on the next forward)"""
        assert operand in OPERANDS, operand
        if operand == self.operand:
            return
        if operand == "fp16" and self.fp8:
            raise RuntimeError("the fp8 mode goes with bf16 operands")
        self.operand = operand
        self.wb16 = torch.zeros(self.n_floats, dtype=self.op_dtype, device=self.device)
        self.wt16 = torch.zeros(self.n_transposed, dtype=self.op_dtype, device=self.device)
        self._weights_version = None

    def _cfg(self, batch: int) -> _abi.VitConfig:
        s = self.spec
        return _abi.VitConfig(batch, s.img_size, s.patch_size, s.in_chans, s.dim, s.depth, s.heads, s.mlp_dim, s.num_classes, s.ln_eps, 0 if s.class_token else 1,
                              self.fp8, self.be.ptr(self.fp8_w) if self.fp8 else None, self.be.ptr(self.fp8_state) if self.fp8 else None,
                              _abi.F16_ if self.operand == "fp16" else _abi.BF16, 1 if s.pre_norm else 0)

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        for n, off, numel, shape in self.entries:
            if n == name:
                return flat[off:off + numel].view(shape)
        raise KeyError(name)

    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch != batch:
            need = C.c_size_t(0)
            cfg = self._cfg(batch)
            self.be.check(self.be.lib.vdk_vit_workspace_bytes(C.byref(cfg), C.byref(need)), "vdk_vit_workspace_bytes")
            if self._ws is None or self._ws.numel() < need.value:      # grow-only: OHEM hands the step a different (smaller) batch every iteration
                self._ws = None
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
            shape = (batch, self.cp) if self.cp else (batch * self.tokens, self.spec.dim)   # logits | final-normed tokens
            self._logits = torch.empty(shape, dtype=torch.float32, device=self.device)
        return self._ws

    def refresh_weights(self, skip_wb16: bool = False) -> None:
        """Rebuild the bf16 operand copies from the fp32 master weights (after init / load_state_dict / a foreign
        optimizer step; `FusedTrainStep` refreshes wb16 inside the optimizer kernel and passes skip_wb16=True)."""
        cfg = self._cfg(1)
        be = self.be
        be.check(be.lib.vdk_vit_refresh_weights(C.byref(cfg), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wt16),
                                                int(skip_wb16), be.stream()), "vdk_vit_refresh_weights")
        self._weights_version = self.params._version

    def enable_fp8(self, mode: int = 1) -> None:
        """BASELINE.json configs[4] "fp8 MFMA": the forward and input-gradient GEMMs of every block Linear on OCP fp8 operands (csrc/gemm_fp8.hip) with per-tensor
        scaling; mode 1 = delayed scaling (this step's scale from the last step's amax), 2 = current scaling (an amax pass per tensor: calibration), 0 = off.
        Weight gradients, attention, LayerNorm, the embeddings and the head stay as they are.  Call fp8_update() after every backward."""
        assert mode in (0, 1, 2)
        if mode and self.operand != "bf16":
            raise RuntimeError("the fp8 mode goes with bf16 operands")
        if mode and self.fp8_w is None:
            L = self.spec.depth
            self.fp8_w = torch.zeros(self.n_floats + self.n_transposed, dtype=torch.uint8, device=self.device)
            st = torch.zeros((3, 12 * L), dtype=torch.float32, device=self.device)
            st[1:] = 1.0
            self.fp8_state = st
        if bool(mode) != bool(self.fp8):
            self._ws, self._ws_batch, self._weights_version = None, -1, None        # the workspace gains / loses the quantised-operand scratch; fp8 weight copies to build
        self.fp8 = mode

    def fp8_update(self) -> None:
        if self.fp8:
            cfg = self._cfg(1)
            self.be.check(self.be.lib.vdk_vit_fp8_update(C.byref(cfg), self.be.stream()), "vdk_vit_fp8_update")

    def _ensure_fresh(self) -> None:
        if self._weights_version != self.params._version:
            self.refresh_weights()

    # ---- the two calls -----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x f32 [B, C, H, W] -> logits f32 [B, Cp] (padded columns beyond num_classes); activations stay in the workspace."""
        s = self.spec
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (s.in_chans, s.img_size, s.img_size):
            raise ValueError(f"expected float32 [B, {s.in_chans}, {s.img_size}, {s.img_size}], got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B = x.shape[0]
        ws = self._workspace(B)
        self._ensure_fresh()
        cfg = self._cfg(B)
        be = self.be
        be.check(be.lib.vdk_vit_forward(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self.wb16), be.ptr(ws), ws.numel(),
                                        be.ptr(self._logits), be.stream()), "vdk_vit_forward")
        return self._logits

    def forward_precise(self, x: torch.Tensor) -> torch.Tensor:
        """Evaluation forward with every contraction on the fp32 MFMA (vdk_vit_forward_f32): logits f32 [B, Cp] / tokens f32 [B*N, D] within
        ~1e-6 of the reference's PyTorch-CPU fp32 path.  Reads the fp32 master weights; keeps nothing for a backward."""
        s = self.spec
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (s.in_chans, s.img_size, s.img_size):
            raise ValueError(f"expected float32 [B, {s.in_chans}, {s.img_size}, {s.img_size}], got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B = x.shape[0]
        cfg = self._cfg(B)
        be = self.be
        need = C.c_size_t(0)
        be.check(be.lib.vdk_vit_workspace_f32_bytes(C.byref(cfg), C.byref(need)), "vdk_vit_workspace_f32_bytes")
        if getattr(self, "_ws32", None) is None or self._ws32.numel() < need.value:
            self._ws32 = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        shape = (B, self.cp) if self.cp else (B * self.tokens, s.dim)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        be.check(be.lib.vdk_vit_forward_f32(C.byref(cfg), be.ptr(x), be.ptr(self.params), be.ptr(self._ws32), self._ws32.numel(), be.ptr(out), be.stream()),
                 "vdk_vit_forward_f32")
        return out

    def backward(self, dlogits_bf16: torch.Tensor, on_ready: Optional[Callable[[int, int], None]] = None) -> torch.Tensor:
        """dlogits [B, Cp] in the operand format (feature mode: d tokens f32 [B*N, D]) -> self.grads (flat fp32, overwritten).  Needs the
        workspace of the matching forward."""
        if self.cp:
            B = dlogits_bf16.shape[0]
            assert dlogits_bf16.dtype == self.op_dtype and dlogits_bf16.shape[1] == self.cp and B == self._ws_batch
        else:
            B = dlogits_bf16.shape[0] // self.tokens
            assert dlogits_bf16.dtype == torch.float32 and dlogits_bf16.shape == (B * self.tokens, self.spec.dim) and B == self._ws_batch
        cfg = self._cfg(B)
        be = self.be
        cb = _abi.GRAD_READY_FN(lambda user, off, n: on_ready(off, n)) if on_ready is not None else _abi.GRAD_READY_FN(0)
        be.check(be.lib.vdk_vit_backward(C.byref(cfg), be.ptr(dlogits_bf16), be.ptr(self.params), be.ptr(self.wb16), be.ptr(self.wt16),
                                         be.ptr(self._ws), self._ws.numel(), be.ptr(self.grads), cb, None, be.stream(),
                                         self.side_stream.cuda_stream if self.side_stream is not None else None),
                 "vdk_vit_backward")
        return self.grads


# =====================================================================================================
class _VitFunction(torch.autograd.Function):
    """model(x) as ONE autograd node: forward = vdk_vit_forward, backward = vdk_vit_backward."""

    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module.engine
        module._sync_flat()
        logits = eng.forward(x)
        ctx.module = module
        ctx.batch = x.shape[0]
        if eng.cp == 0:   # feature mode: [B, N, D] final-normed tokens (timm num_classes=0, global_pool='')
            return logits.view(x.shape[0], eng.tokens, eng.spec.dim).clone()
        return logits[:, :eng.spec.num_classes].clone()

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        eng = module.engine
        be = eng.be
        if eng.cp == 0:
            g = eng.backward(dlogits.contiguous().view(-1, eng.spec.dim))
            grads = tuple(g[off:off + numel].view(shape) for (_, off, numel, shape) in eng.entries)
            return (None, None) + grads
        B, Cn = dlogits.shape
        # bf16 + zero-padded columns for the head GEMMs: one cast kernel on a padded staging buffer
        stage = torch.zeros((B, eng.cp), dtype=torch.float32, device=dlogits.device)
        stage[:, :Cn].copy_(dlogits)
        # (fp16 operands: under the reference's GradScaler the incoming gradient already carries the loss scale, train.py:205)
        dl = torch.empty((B, eng.cp), dtype=eng.op_dtype, device=dlogits.device)
        cast = be.lib.vdk_cast_f32_f16 if eng.operand == "fp16" else be.lib.vdk_cast_f32_bf16
        be.check(cast(be.ptr(stage), be.ptr(dl), stage.numel(), be.stream()), "vdk_cast_f32_bf16")
        g = eng.backward(dl)
        grads = tuple(g[off:off + numel].view(shape) for (_, off, numel, shape) in eng.entries)
        return (None, None) + grads


class _Holder(nn.Module):
    """empty container mirroring one level of timm's module tree (patch_embed, blocks.3.attn, ...); owns Parameters only"""


class VisionTransformer(nn.Module):
    """Drop-in for the object `timm.create_model('vit_*', pretrained=False, num_classes=C)` hands the reference.

    The module tree mirrors timm's (patch_embed.proj, blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm, head) with
    parameter-only holder modules, so named_parameters()/state_dict()/load_state_dict() carry timm's key names at any nesting
    depth; every Parameter is a view into the engine's flat fp32 buffer."""

    def __init__(self, spec: VitSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None, operand: str = "bf16"):
        super().__init__()
        self.spec = spec
        self.engine = VitEngine(spec, device=device, backend=backend, operand=operand)
        self.num_classes = spec.num_classes
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

    @property
    def _names(self):   # (timm name, Parameter) pairs in flat-buffer order
        return self._plist

    def reset_parameters(self, seed: Optional[int] = None) -> None:
        """timm defaults + the reference's override (classify_model.py:70-81): N(0,.02) Conv/Linear weights, zero Linear
        biases (conv bias left at its default init), LayerNorm (1,0), pos_embed trunc-normal .02, cls_token normal 1e-6."""
        gen = torch.Generator(device="cpu")
        # no explicit seed: draw it from torch's global generator, so that `torch.manual_seed(s)` reproduces the initialisation like it does for the
        # reference's model (and data-parallel ranks seeded alike start alike even before the broadcast)
        gen.manual_seed(seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        s = self.spec
        with torch.no_grad():
            for name, p in self._plist:
                if name == "pos_embed":
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen).clamp_(-2.0, 2.0)
                elif name == "cls_token":
                    v = torch.empty(p.shape).normal_(0, 1e-6, generator=gen)
                elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name in ("norm.weight", "norm_pre.weight"):
                    v = torch.ones(p.shape)
                elif name == "patch_embed.proj.bias":
                    bound = 1.0 / math.sqrt(s.in_chans * s.patch_size * s.patch_size)
                    v = (torch.rand(p.shape, generator=gen) * 2 - 1) * bound
                elif name.endswith(".bias"):
                    v = torch.zeros(p.shape)
                else:
                    v = torch.empty(p.shape).normal_(0, 0.02, generator=gen)
                p.copy_(v.to(p.device))

    # keep the flat buffer authoritative even if someone re-pointed a parameter (.to(), deepcopy, p.data = ...)
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
                raise RuntimeError("visiondk_amd ViT keeps fp32 master weights; bf16 copies are internal")
            if self.engine.be.device_only and probe.device.type != "cuda":
                raise RuntimeError("visiondk_amd ViT lives on the GPU (no CPU fallback)")
            eng = self.engine
            eng.device = probe.device
            for attr in ("params", "grads", "wb16", "wt16"):      # (.to keeps each buffer's dtype: fp32 masters, 16-bit operand copies)
                setattr(eng, attr, getattr(eng, attr).to(probe.device))
            eng._ws, eng._ws_batch, eng._weights_version = None, -1, None
            for (name, off, numel, shape), (_, p) in zip(eng.entries, self._plist):
                p.data = eng.params[off:off + numel].view(shape)
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _VitFunction.apply(x, self, *[p for _, p in self._plist])

    @torch.no_grad()
    def forward_precise(self, x: torch.Tensor) -> torch.Tensor:
        """`model(x)` for evaluation with fp32-MFMA contractions (no autograd): logits [B, C] or, in feature mode, tokens [B, N, D]"""
        eng = self.engine
        self._sync_flat()
        out = eng.forward_precise(x)
        if eng.cp == 0:
            return out.view(x.shape[0], eng.tokens, eng.spec.dim)
        return out[:, :eng.spec.num_classes]


# ---- global_pool='map': timm AttentionPoolLatent on the HIP kernels ----------------------------------------------------------------------
def _gelu(u):
    return 0.5 * u * (1.0 + torch.erf(u * 0.7071067811865476))


def _gelu_grad(u):
    return 0.5 * (1.0 + torch.erf(u * 0.7071067811865476)) + u * torch.exp(-0.5 * u * u) * 0.3989422804014327


def _pad4(t: torch.Tensor) -> torch.Tensor:
    """zero-pad the contraction (last) dim to a multiple of 4 (16-byte rows for the fp32 GEMM)"""
    k = t.shape[-1]
    return t.contiguous() if k % 4 == 0 else torch.nn.functional.pad(t, (0, 4 - k % 4)).contiguous()


def _wgrad_f32(dy: torch.Tensor, x: torch.Tensor, be) -> torch.Tensor:
    """dW [out, in] = dy^T x for [rows, out] / [rows, in] fp32 operands with few rows (the pooled [B, D] tail): fp32 MFMA GEMM over the transposed operands"""
    return ops.gemm_f32(_pad4(dy.t()), _pad4(x.t()), backend=be)


class _AttnPoolFn(torch.autograd.Function):
    """AttentionPoolLatent.forward as one autograd node: the kv Linear over all tokens on the bf16 MFMA GEMM (the only part with real work: 2 T D^2 MACs), the
    one-query attention in csrc/attn_pool.hip, the [B, D]-sized tail (proj, LayerNorm, MLP) on the fp32 MFMA GEMM; the backward mirrors it (TN weight gradient
    straight from dkv / tokens)."""

    @staticmethod
    def forward(ctx, tokens, latent, q_w, q_b, kv_w, kv_b, proj_w, proj_b, n_w, n_b, fc1_w, fc1_b, fc2_w, fc2_b, mod):
        be = mod.be
        B, N, D = tokens.shape
        H = mod.num_heads
        dev = tokens.device
        tok = tokens.detach().contiguous().view(B * N, D)
        odt = mod.op_dtype                                                  # the trunk's operand format: bf16, or fp16 (the reference's autocast dtype)
        tb = ops.cast_16(tok, odt, backend=be)
        kv = ops.gemm_nt(tb, ops.cast_16(kv_w.detach().contiguous(), odt, backend=be), bias=kv_b.detach(), backend=be)          # 16-bit [T, 2D]
        q = ops.gemm_f32(latent.detach().reshape(1, D).contiguous(), q_w.detach(), bias=q_b.detach(), backend=be).view(D)
        pooled = torch.empty((B, D), dtype=torch.float32, device=dev)
        probs = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        be.check(be.lib.vdk_attn_pool_fwd_dt(be.ptr(q), be.ptr(kv), 2 * D, B, N, H, mod.scale, be.ptr(pooled), D, be.ptr(probs), ops._dt(odt), be.stream()), "vdk_attn_pool_fwd")
        y1 = ops.gemm_f32(pooled, proj_w.detach(), bias=proj_b.detach(), backend=be)
        h, mean, rstd = ops.layernorm_fwd(y1, n_w.detach(), n_b.detach(), eps=mod.eps, out_dtype=torch.float32, backend=be)
        u = ops.gemm_f32(h, fc1_w.detach(), bias=fc1_b.detach(), backend=be)
        g = _gelu(u)
        out = ops.gemm_f32(g, fc2_w.detach(), bias=fc2_b.detach(), residual=y1, backend=be)
        ctx.save_for_backward(tb, kv, q, probs, pooled, y1, h, mean, rstd, u, g, latent, q_w, kv_w, proj_w, n_w, fc1_w, fc2_w)
        ctx.mod, ctx.dims = mod, (B, N, D, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        tb, kv, q, probs, pooled, y1, h, mean, rstd, u, g, latent, q_w, kv_w, proj_w, n_w, fc1_w, fc2_w = ctx.saved_tensors
        mod = ctx.mod
        be = mod.be
        B, N, D, H = ctx.dims
        dev = dout.device
        dout = dout.contiguous().float()
        f32 = lambda a, b, **k: ops.gemm_f32(a, b.detach(), backend=be, **k)
        dg = f32(dout, fc2_w, b_kmajor=True)                               # [B, Dm] = dout . W2
        dfc2_w, dfc2_b = _wgrad_f32(dout, g, be), dout.sum(0)
        du = dg * _gelu_grad(u)
        dfc1_w, dfc1_b = _wgrad_f32(du, h, be), du.sum(0)
        dh = f32(du, fc1_w, b_kmajor=True)
        dy1, _, dn_w, dn_b = ops.layernorm_bwd(dh, y1, mean, rstd, n_w.detach(), dres=dout, want_bf16=False, backend=be)   # + the skip connection
        dproj_w, dproj_b = _wgrad_f32(dy1, pooled, be), dy1.sum(0)
        dpooled = f32(dy1, proj_w, b_kmajor=True)
        odt = kv.dtype
        dkv = torch.empty((B * N, 2 * D), dtype=odt, device=dev)
        dq_part = torch.empty((B, D), dtype=torch.float32, device=dev)
        be.check(be.lib.vdk_attn_pool_bwd_dt(be.ptr(q), be.ptr(kv), 2 * D, be.ptr(probs), be.ptr(dpooled), D, B, N, H, mod.scale, be.ptr(dkv), 2 * D, be.ptr(dq_part),
                                             ops._dt(odt), be.stream()), "vdk_attn_pool_bwd")
        dq = ops.reduce_rows(dq_part, backend=be)                          # [D]
        lat = latent.detach().reshape(1, D)
        dq_w, dq_b = dq.view(D, 1) * lat, dq
        dlatent = f32(dq.view(1, D).contiguous(), q_w, b_kmajor=True).view(latent.shape)
        T = B * N
        if T % 64 == 0:
            dkv_w = ops.gemm_nt(dkv, tb, out_dtype=torch.float32, trans=True, backend=be)                    # [2D, D] = dkv^T tokens, operands as they lie
        else:
            Tp = (T + 63) // 64 * 64
            dkv_w = ops.gemm_nt(ops.transpose_pad(dkv, rpad=Tp, backend=be), ops.transpose_pad(tb, rpad=Tp, backend=be), out_dtype=torch.float32, backend=be)
        if odt == torch.bfloat16:
            dkv_b = ops.colsum_bf16(dkv, backend=be)
            wkv_t = ops.transpose_cast(kv_w.detach().contiguous(), backend=be)
        else:      # fp16: the [2D] column sums and the [D, 2D] transposed weight copy through torch (2 of the step's ~10^3 launches; fp32 accumulation)
            dkv_b = torch.sum(dkv, dim=0, dtype=torch.float32)
            wkv_t = kv_w.detach().t().contiguous().to(odt)
        dtok = ops.gemm_nt(dkv, wkv_t, out_dtype=torch.float32, backend=be)   # [T, D] = dkv . Wkv
        return (dtok.view(B, N, D), dlatent, dq_w, dq_b, dkv_w, dkv_b, dproj_w, dproj_b, dn_w, dn_b, dfc1_w, dfc1_b, dfc2_w, dfc2_b, None)


class AttentionPoolLatent(nn.Module):
    """timm.layers.AttentionPoolLatent as the SigLIP ViTs configure it (latent_len 1, qkv_bias, no qk_norm, LayerNorm eps 1e-6, mlp_ratio 4, exact GELU, pool 'token');
    parameter names equal timm's (latent, q, kv, proj, norm, mlp.fc1, mlp.fc2).  head_dim 64."""

    def __init__(self, dim: int, num_heads: int, mlp_dim: Optional[int] = None, eps: float = 1e-6, backend: Optional[_lib.Backend] = None, device=None,
                 generator: Optional[torch.Generator] = None, op_dtype=torch.bfloat16):
        super().__init__()
        assert dim == num_heads * 64, "head_dim must be 64"
        self.op_dtype = op_dtype             # 16-bit format of the kv Linear's operands and of kv / dkv (the trunk's)
        self.be = backend or _lib.load()
        dev = device if device is not None else ("cuda" if self.be.device_only else "cpu")
        self.num_heads, self.scale, self.eps = num_heads, 0.125, eps
        mlp_dim = mlp_dim or 4 * dim
        self.latent = nn.Parameter(torch.empty(1, 1, dim, device=dev))
        self.q, self.kv, self.proj = _Holder(), _Holder(), _Holder()
        self.norm, self.mlp = _Holder(), _Holder()
        self.mlp.fc1, self.mlp.fc2 = _Holder(), _Holder()
        for holder, shape in ((self.q, (dim, dim)), (self.kv, (2 * dim, dim)), (self.proj, (dim, dim)), (self.mlp.fc1, (mlp_dim, dim)), (self.mlp.fc2, (dim, mlp_dim))):
            holder.weight = nn.Parameter(torch.empty(shape, device=dev)); holder.bias = nn.Parameter(torch.zeros(shape[0], device=dev))
        self.norm.weight = nn.Parameter(torch.ones(dim, device=dev)); self.norm.bias = nn.Parameter(torch.zeros(dim, device=dev))
        self.reset_parameters(generator)

    def reset_parameters(self, generator: Optional[torch.Generator] = None):
        with torch.no_grad():
            D = self.latent.shape[-1]
            lat = torch.empty(1, 1, D)
            torch.nn.init.trunc_normal_(lat, std=D ** -0.5, a=-2 * D ** -0.5, b=2 * D ** -0.5, generator=generator)     # timm: trunc_normal_tf_(latent, std=D^-0.5)
            self.latent.copy_(lat.to(self.latent.device))
            for h in (self.q, self.kv, self.proj, self.mlp.fc1, self.mlp.fc2):     # the reference's reset_parameters override: N(0, .02) Linear weights, zero biases
                h.weight.copy_(torch.empty(h.weight.shape).normal_(0, 0.02, generator=generator).to(h.weight.device)); h.bias.zero_()

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        return _AttnPoolFn.apply(tokens, self.latent, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, self.proj.weight, self.proj.bias, self.norm.weight,
                                 self.norm.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, self)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the fp32 MFMA GEMM (the classifier head on pooled [B, D] rows)"""

    @staticmethod
    def forward(ctx, x, w, b, be):
        ctx.save_for_backward(x, w); ctx.be = be
        return ops.gemm_f32(x.contiguous(), w.detach(), bias=b.detach(), backend=be)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        Cn = dy.shape[1]
        if Cn % 4:      # class counts that are not multiples of 4: zero columns / rows keep the contraction 16-byte aligned
            dx = ops.gemm_f32(_pad4(dy), torch.nn.functional.pad(w.detach(), (0, 0, 0, 4 - Cn % 4)).contiguous(), b_kmajor=True, backend=ctx.be)
        else:
            dx = ops.gemm_f32(dy, w.detach(), b_kmajor=True, backend=ctx.be)
        return dx, _wgrad_f32(dy, x, ctx.be), dy.sum(0), None


class VisionTransformerMap(nn.Module):
    """timm VisionTransformer(class_token=False, global_pool='map', num_classes=C): the vit_*_siglip_* family (BASELINE.json configs[4]).  The trunk is the native
    engine in feature mode without a class token; attn_pool and head run as autograd nodes over the same kernels.  state_dict names equal timm's."""

    def __init__(self, spec: VitSpec, device=None, backend: Optional[_lib.Backend] = None, seed: Optional[int] = None, operand: str = "bf16"):
        super().__init__()
        import dataclasses
        self.spec = spec
        self.num_classes = spec.num_classes
        trunk = VisionTransformer(dataclasses.replace(spec, num_classes=0, class_token=False), device=device, backend=backend, seed=seed, operand=operand)
        self.engine = trunk.engine
        self._trunk = [trunk]                 # not a registered submodule: its parameters are re-registered below under timm's flat names
        for name, child in trunk._modules.items():
            self.add_module(name, child)
        for name, p in trunk._parameters.items():
            self.register_parameter(name, p)
        be = trunk.engine.be
        dev = trunk.engine.device
        gen = None
        if seed is not None:      # the pooling head and the classifier draw from the same seed as the trunk: a seeded model is reproducible (and equal on every rank)
            gen = torch.Generator(); gen.manual_seed(int(seed) + 1)
        self.attn_pool = AttentionPoolLatent(spec.dim, spec.heads, spec.mlp_dim, spec.ln_eps, backend=be, device=dev, generator=gen, op_dtype=trunk.engine.op_dtype)
        self.head = _Holder()
        if spec.num_classes > 0:
            self.head.weight = nn.Parameter(torch.empty(spec.num_classes, spec.dim).normal_(0, 0.02, generator=gen).to(dev)); self.head.bias = nn.Parameter(torch.zeros(spec.num_classes, device=dev))

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        return self._trunk[0](x)              # [B, N, D], final-normed

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self.attn_pool.op_dtype = self.engine.op_dtype          # (follows engine.set_operand)
        pooled = self.attn_pool(self.forward_features(x))
        if self.num_classes == 0:
            return pooled
        return _LinearFn.apply(pooled, self.head.weight, self.head.bias, self.engine.be)


def create_model(name: str, pretrained: bool = False, num_classes: int = 1000, device=None, backend=None, img_size=None,
                 global_pool: str = "token", operand: str = "bf16", **kwargs) -> VisionTransformer:
    """`timm.create_model(name, pretrained=..., num_classes=...)` for the ids in TIMM_VITS.  `num_classes=0, global_pool=''`
    (what TimmWrapper asks for, timm_wrapper.py:16-21) gives the feature model: forward -> final-normed tokens [B, N, D]."""
    if pretrained:
        raise RuntimeError("pretrained weights need network access; load a checkpoint with load_state_dict instead")
    spec = spec_from_timm_name(name, num_classes, img_size)
    if not spec.class_token:      # the SigLIP family: attention-pool head (timm's default global_pool for these ids is 'map'); global_pool='' -> token features
        if global_pool == "":
            return VisionTransformer(dataclasses.replace(spec, num_classes=0), device=device, backend=backend, operand=operand)
        if global_pool not in ("map", "token"):      # ("token" is this function's default argument, i.e. "not given": timm then uses the id's own default, 'map')
            raise NotImplementedError(f"global_pool={global_pool!r} is not built for the SigLIP ids (only 'map' and '')")
        return VisionTransformerMap(spec, device=device, backend=backend, operand=operand)
    if num_classes == 0 and global_pool != "":
        raise NotImplementedError("num_classes=0 is supported with global_pool='' (token features) only")
    return VisionTransformer(spec, device=device, backend=backend, operand=operand)


# =====================================================================================================
class FusedTrainStep:
    """Trainer.compute_loss (plain or mixup) + Trainer.update as one fixed kernel sequence over flat buffers.

    loss = CE(label_smoothing) [mixup: lam*CE(ya) + (1-lam)*CE(yb)]  ->  backward  ->  [all-reduce(mean) of the flat
    gradient in buckets, overlapped with the rest of backward]  ->  clip_grad_norm_(max_norm)  ->  SGD(momentum,
    weight_decay)  ->  ModelEMA.update  ->  bf16 weight refresh.   (train.py:196,203-215; optimizer.py:119-121; ema.py:28-37)

    fp16 operands (`model.engine.operand == "fp16"`, the reference's autocast dtype): the step also does what the reference's GradScaler does around it
    (vision_engine.py:232, train.py:205-211) -- the loss gradient is multiplied by a loss scale that lives on the device (`loss_state` = [scale, growth tracker, skipped
    steps], torch's defaults: 65536, x2 every 2000 clean steps, x0.5 on overflow), the optimizer pass un-scales, and a step whose gradient holds an inf / NaN is skipped.
    No host synchronisation anywhere: `loss_scale()` / `skipped_steps()` read the state back when asked.
    """

    def __init__(self, model: VisionTransformer, lr: float, momentum: float = 0.937, weight_decay: float = 5e-4,
                 label_smoothing: float = 0.0, max_norm: float = 10.0, ema: bool = True, comm=None, sam: bool = False,
                 sam_rho: float = 0.05, sam_adaptive: bool = True, init_scale: float = 65536.0):
        self.model = model
        self.eng = model.engine
        self.be = self.eng.be
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.label_smoothing, self.max_norm = label_smoothing, max_norm
        dev = self.eng.device
        self.momentum_buf = torch.zeros_like(self.eng.params)
        self.ema = self.eng.params.clone() if ema else None
        self.updates = 0
        self.comm = comm
        # Trainer.update_sam (train.py:150-175) over engine/optimizer.py's SAM(base SGD): 2 fwd/bwd, first un-synchronised, no clipping
        self.sam, self.sam_rho, self.sam_adaptive = sam, sam_rho, sam_adaptive
        self._old_params = torch.empty_like(self.eng.params) if sam else None
        self._normsq = torch.zeros(1, dtype=torch.float32, device=dev)
        need = C.c_size_t(0)
        self.be.check(self.be.lib.vdk_sumsq_workspace_bytes(C.byref(need)), "vdk_sumsq_workspace_bytes")
        self._sumsq_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._loss_rows: Optional[torch.Tensor] = None
        self._dl: Optional[torch.Tensor] = None
        # GradScaler state on the device (fp16 operands only): scale, growth tracker, skipped-step count
        self.amp = self.eng.operand == "fp16"
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        self.loss_state = torch.tensor([init_scale, 0.0, 0.0], dtype=torch.float32, device=dev) if self.amp else None
        if self.amp and sam:
            raise NotImplementedError("update_sam calls loss.backward() without the scaler (train.py:157-170): the SAM step runs on bf16 operands")
        # Trainer reads param_groups[0]['lr'] and WRITES 'momentum' after the warm-up (vision_engine.py:169-171,350-352): every step reads all three from here
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        if comm is not None and comm.active:          # DDP-constructor semantics: every rank starts from rank 0's weights
            comm.broadcast_params(self.eng.params, src=0, engine=self.eng)
            if self.ema is not None:
                self.ema.copy_(self.eng.params)

    def _fwd_loss_bwd(self, x, y, y_b, lam, sync: bool) -> None:
        eng, be = self.eng, self.be
        B = x.shape[0]
        logits = eng.forward(x)
        if self._loss_rows is None or self._loss_rows.shape[0] != B:
            self._loss_rows = torch.empty(B, dtype=torch.float32, device=eng.device)
            self._dl = torch.empty((B, eng.cp), dtype=eng.op_dtype, device=eng.device)
        be.check(be.lib.vdk_softmax_ce_amp(be.ptr(logits), eng.cp, B, eng.spec.num_classes, be.ptr(y), be.ptr(y_b), lam,
                                           self.label_smoothing, 1.0 / B, be.ptr(self.loss_state), be.ptr(self._loss_rows), be.ptr(self._dl), eng.cp,
                                           _abi.F16_ if self.amp else _abi.BF16, None, 0, be.stream()), "vdk_softmax_ce")
        if self.comm is not None and sync:
            self.comm.begin_step(eng.grads)
            eng.backward(self._dl, on_ready=self.comm.on_grad_ready)
            self.comm.finish_step()
        else:
            eng.backward(self._dl)
        eng.fp8_update()

    def ohem_select(self, x: torch.Tensor, y: torch.Tensor, min_kept: int, thresh: float, ignore_index: int = 255):
        """OHEM-Softmax pre-pass of the reference's loop (engine/procedure/train.py:113-117, structure/sampler.py:11-31): one extra no-grad forward in
        training mode, keep the samples whose target probability is below max(thresh, the min_kept-th smallest); returns the kept (images, labels).
        The batch that reaches step() then changes size from iteration to iteration (workspaces only grow)."""
        from . import ops
        self.model._sync_flat()
        logits = self.eng.forward(x)
        mask = ops.ohem_mask(logits[:, :self.eng.spec.num_classes].contiguous(), y, min_kept, thresh, ignore_index, backend=self.be)
        keep = mask.nonzero().squeeze(1)          # the one host sync boolean indexing costs in the reference too
        return x[keep].contiguous(), y[keep].contiguous()

    def step(self, x: torch.Tensor, y: torch.Tensor, y_b: Optional[torch.Tensor] = None, lam: float = 1.0) -> torch.Tensor:
        eng, be = self.eng, self.be
        self.model._sync_flat()
        world = self.comm.world_size if self.comm is not None else 1
        g0 = self.param_groups[0]
        lr, momentum, weight_decay = g0["lr"], g0["momentum"], g0["weight_decay"]
        self.updates += 1
        if eng.fp8:
            eng.fp8 = 2 if self.updates == 1 else 1          # first step: current scaling (nothing to delay from yet)
        d = 0.9999 * (1 - math.exp(-self.updates / 2000)) if self.ema is not None else 0.0
        if self.sam:
            # first forward-backward on w with LOCAL gradients (model.no_sync(), train.py:157-159), then climb to w + e(w)
            self._fwd_loss_bwd(x, y, y_b, lam, sync=False)
            loss_first = self._loss_rows.clone()
            be.check(be.lib.vdk_sam_first_step(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self._old_params), eng.n_floats, self.sam_rho,
                                               int(self.sam_adaptive), be.ptr(self._normsq), be.ptr(self._sumsq_ws), self._sumsq_ws.numel(),
                                               be.stream()), "vdk_sam_first_step")
            eng.refresh_weights()
            # second forward-backward at w + e(w) (gradients all-reduced), back to w, base optimizer step (no clipping on this path)
            self._fwd_loss_bwd(x, y, y_b, lam, sync=True)
            eng.params.copy_(self._old_params)
            be.check(be.lib.vdk_sgd_step(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16),
                                         eng.n_floats, lr, momentum, weight_decay, 1.0 / world, None, self.max_norm, d,
                                         int(self.updates == 1), be.stream()), "vdk_sgd_step")
            eng.refresh_weights(skip_wb16=True)
            self._loss_rows.copy_(loss_first)     # update_sam returns the FIRST loss (train.py:175)
            return self._loss_rows
        self._fwd_loss_bwd(x, y, y_b, lam, sync=True)
        be.check(be.lib.vdk_sumsq_f32(be.ptr(eng.grads), eng.n_floats, be.ptr(self._normsq), be.ptr(self._sumsq_ws),
                                      self._sumsq_ws.numel(), be.stream()), "vdk_sumsq_f32")
        if self.amp:
            be.check(be.lib.vdk_sgd_step_amp(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), _abi.F16_,
                                             eng.n_floats, lr, momentum, weight_decay, 1.0 / world, be.ptr(self.loss_state), be.ptr(self._normsq), self.max_norm, d,
                                             int(self.updates == 1), be.stream()), "vdk_sgd_step_amp")
            be.check(be.lib.vdk_loss_scale_update(be.ptr(self.loss_state), be.ptr(self._normsq), self.growth_factor, self.backoff_factor, self.growth_interval,
                                                  be.stream()), "vdk_loss_scale_update")
        else:
            be.check(be.lib.vdk_sgd_step(be.ptr(eng.params), be.ptr(eng.grads), be.ptr(self.momentum_buf), be.ptr(self.ema),
                                         be.ptr(eng.wb16), eng.n_floats, lr, momentum, weight_decay, 1.0 / world,
                                         be.ptr(self._normsq), self.max_norm, d, int(self.updates == 1), be.stream()), "vdk_sgd_step")
        eng.refresh_weights(skip_wb16=True)
        return self._loss_rows

    def loss_scale(self) -> float:
        """GradScaler.get_scale(): the current loss scale (a device->host read; 1.0 on bf16 operands)"""
        return float(self.loss_state[0].item()) if self.amp else 1.0

    def skipped_steps(self) -> int:
        """steps whose scaled gradient overflowed and were skipped (GradScaler's found_inf path)"""
        return int(self.loss_state[2].item()) if self.amp else 0

    def scaler_state_dict(self) -> dict:
        """`scaler.state_dict()` with torch.cuda.amp.GradScaler's keys -- what the reference checkpoints and restores (engine/vision_engine.py:296,397), so a resumed fp16
        run continues at the scale it had reached instead of re-discovering it through skipped steps; {} on bf16 operands (a disabled GradScaler's state_dict)."""
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

    def loss_value(self) -> float:
        """mean loss of the last step (this is the only device->host sync; the reference does it every step, train.py:122)."""
        return float(self._loss_rows.mean().item())


class MapTrainStep:
    """FusedTrainStep for VisionTransformerMap (class_token=False + attention-pool head): the same Trainer.compute_loss / update / update_sam sequence
    (train.py:150-215) over ONE flat fp32 buffer that holds the engine's parameters followed by the attn_pool / head parameters, so the gradient norm, SAM's
    e(w), the clipped SGD step and the EMA are single kernels over everything.  The trunk runs through the engine's forward / backward directly (no autograd
    copies of 300 M gradients); only the pooled [B, D] tail goes through autograd nodes.  comm (GradAllReduce): data parallel like FusedTrainStep -- the pooling head's
    gradients (produced first) leave as one bucket, the trunk's in flat buckets from inside vdk_vit_backward, overlapped with the rest of the backward; the SAM
    step's first pass stays local (model.no_sync(), train.py:157-159)."""

    def __init__(self, model: VisionTransformerMap, lr: float, momentum: float = 0.937, weight_decay: float = 5e-4, label_smoothing: float = 0.0,
                 max_norm: float = 10.0, ema: bool = True, sam: bool = False, sam_rho: float = 0.05, sam_adaptive: bool = True, comm=None, init_scale: float = 65536.0):
        self.model, self.eng = model, model.engine
        self.comm = comm
        eng = self.eng
        self.be = eng.be
        dev = eng.device
        trunk = model._trunk[0]
        trunk._sync_flat()
        self.extras = [p for p in model.attn_pool.parameters()] + [p for p in model.head.parameters()]
        n = eng.n_floats
        offs, cur = [], n
        for p in self.extras:
            offs.append(cur); cur += (p.numel() + 63) // 64 * 64
        self.n_total = cur
        big = torch.zeros(cur, dtype=torch.float32, device=dev)
        big[:n].copy_(eng.params)
        eng.params = big[:n]
        for (name, off, numel, shape), (_, p) in zip(eng.entries, trunk._plist):
            p.data = eng.params[off:off + numel].view(shape)
        self.gbig = torch.zeros(cur, dtype=torch.float32, device=dev)
        eng.grads = self.gbig[:n]
        self._extra_offs = offs
        for p, off in zip(self.extras, offs):
            big[off:off + p.numel()].copy_(p.detach().reshape(-1))
            p.data = big[off:off + p.numel()].view(p.shape)
        self._bind_extra_grads()
        eng._weights_version = None
        self.big = big
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        self.label_smoothing, self.max_norm = label_smoothing, max_norm
        self.momentum_buf = torch.zeros_like(big)
        self.ema = big.clone() if ema else None
        self.sam, self.sam_rho, self.sam_adaptive = sam, sam_rho, sam_adaptive
        self._old = torch.empty_like(big) if sam else None
        self._normsq = torch.zeros(1, dtype=torch.float32, device=dev)
        need = C.c_size_t(0)
        self.be.check(self.be.lib.vdk_sumsq_workspace_bytes(C.byref(need)), "vdk_sumsq_workspace_bytes")
        self._sumsq_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self.updates = 0
        self._loss_rows = None
        # fp16 operands (the reference's autocast dtype): the GradScaler protocol of FusedTrainStep -- loss scale on the device, un-scaling inside the optimizer pass, a step
        # whose gradient holds an inf / NaN is skipped and the scale backed off.  The SAM step: the reference's update_sam (train.py:150-175) calls loss.backward() WITHOUT
        # the scaler, i.e. its fp16 gradients are unscaled and small ones flush to zero; here both passes of a SAM step run at the current loss scale (e(w) = rho g / |g| is
        # invariant under it), the base step un-scales, and an overflow in either pass -- the first one poisons e(w) and with it the second -- reaches the second pass's
        # gradient norm: the update is skipped, the weights return to w (they are restored before the base step anyway) and the scale halves.  Same arithmetic as the
        # reference wherever nothing overflows or underflows; where fp16 would have flushed a gradient the scaled pass keeps it.
        self.amp = eng.operand == "fp16"
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        self.loss_state = torch.tensor([init_scale, 0.0, 0.0], dtype=torch.float32, device=dev) if self.amp else None
        if comm is not None and comm.active:          # DDP-constructor semantics: every rank starts from rank 0's weights (trunk and head alike)
            comm.broadcast_params(self.big, src=0, engine=eng)
            if self.ema is not None:
                self.ema.copy_(self.big)

    def _bind_extra_grads(self) -> None:
        """p.grad of the pooling head / classifier = views of the flat gradient buffer, so autograd accumulates in place.  Re-bound before every backward: a
        model.zero_grad() (set_to_none=True is torch's default) would otherwise leave autograd allocating fresh tensors and the flat buffer at zero -- the head
        would silently train on weight decay alone and drop out of the clip norm, SAM and the all-reduce."""
        for p, off in zip(self.extras, self._extra_offs):
            g = self.gbig[off:off + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self, set_to_none: bool = False) -> None:
        """optimizer-protocol no-op: the step zeroes its own flat gradient buffer"""

    def _fwd_loss_bwd(self, x, y, y_b, lam, sync: bool = True):
        eng, m = self.eng, self.model
        B = x.shape[0]
        self._bind_extra_grads()
        self.gbig[eng.n_floats:].zero_()
        tokens = eng.forward(x).view(B, eng.tokens, eng.spec.dim).detach().requires_grad_(True)
        pooled = m.attn_pool(tokens)
        logits = _LinearFn.apply(pooled, m.head.weight, m.head.bias, self.be)
        self._loss_rows, _, dlf = ops.softmax_ce(logits.detach().contiguous(), y, y_b, lam, self.label_smoothing, 1.0 / B, backend=self.be)
        if self.amp:
            ops.scale_dev(dlf, self.loss_state, backend=self.be)        # scaler.scale(loss).backward(): the loss scale enters with the loss gradient
        logits.backward(dlf)
        dtok = tokens.grad.contiguous().view(-1, eng.spec.dim)
        if self.comm is not None and sync and self.comm.active:
            self.comm.begin_step(self.gbig)
            self.comm.on_grad_ready(eng.n_floats, self.n_total - eng.n_floats)      # the head's gradients are complete: first bucket
            self.comm._flush()
            eng.backward(dtok, on_ready=self.comm.on_grad_ready)
            self.comm.finish_step()
        else:
            eng.backward(dtok)
        eng.fp8_update()

    def step(self, x: torch.Tensor, y: torch.Tensor, y_b: Optional[torch.Tensor] = None, lam: float = 1.0) -> torch.Tensor:
        eng, be, n = self.eng, self.be, self.eng.n_floats
        g0 = self.param_groups[0]
        lr, momentum, weight_decay = g0["lr"], g0["momentum"], g0["weight_decay"]
        self.updates += 1
        if eng.fp8:
            eng.fp8 = 2 if self.updates == 1 else 1          # first step: current scaling (nothing to delay from yet)
        d = 0.9999 * (1 - math.exp(-self.updates / 2000)) if self.ema is not None else 0.0
        first = int(self.updates == 1)
        nx = self.n_total - n
        gs = 1.0 / (self.comm.world_size if self.comm is not None else 1)

        def sgd(normsq):
            ex = lambda t: be.ptr(t[n:]) if t is not None else None
            if self.amp:
                # (the norm of the scaled gradient is what detects an overflow; normsq None = "no clipping" (SAM's base step): the norm is still taken, the clip disabled)
                be.check(be.lib.vdk_sumsq_f32(be.ptr(self.gbig), self.n_total, be.ptr(self._normsq), be.ptr(self._sumsq_ws), self._sumsq_ws.numel(), be.stream()), "vdk_sumsq_f32")
                mx = self.max_norm if normsq is not None else 3.0e38
                be.check(be.lib.vdk_sgd_step_amp(be.ptr(self.big), be.ptr(self.gbig), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), _abi.F16_, n, lr, momentum,
                                                 weight_decay, gs, be.ptr(self.loss_state), be.ptr(self._normsq), mx, d, first, be.stream()), "vdk_sgd_step_amp")
                be.check(be.lib.vdk_sgd_step_amp(ex(self.big), ex(self.gbig), ex(self.momentum_buf), ex(self.ema), None, _abi.F16_, nx, lr, momentum, weight_decay, gs,
                                                 be.ptr(self.loss_state), be.ptr(self._normsq), mx, d, first, be.stream()), "vdk_sgd_step_amp")
                be.check(be.lib.vdk_loss_scale_update(be.ptr(self.loss_state), be.ptr(self._normsq), self.growth_factor, self.backoff_factor, self.growth_interval,
                                                      be.stream()), "vdk_loss_scale_update")
                eng.refresh_weights(skip_wb16=True)
                return
            be.check(be.lib.vdk_sgd_step(be.ptr(self.big), be.ptr(self.gbig), be.ptr(self.momentum_buf), be.ptr(self.ema), be.ptr(eng.wb16), n, lr, momentum,
                                         weight_decay, gs, normsq, self.max_norm, d, first, be.stream()), "vdk_sgd_step")
            be.check(be.lib.vdk_sgd_step(ex(self.big), ex(self.gbig), ex(self.momentum_buf), ex(self.ema), None, nx, lr, momentum, weight_decay, gs, normsq,
                                         self.max_norm, d, first, be.stream()), "vdk_sgd_step")
            eng.refresh_weights(skip_wb16=True)

        if self.sam:
            self._fwd_loss_bwd(x, y, y_b, lam, sync=False)
            loss_first = self._loss_rows.clone()
            be.check(be.lib.vdk_sam_first_step(be.ptr(self.big), be.ptr(self.gbig), be.ptr(self._old), self.n_total, self.sam_rho, int(self.sam_adaptive),
                                               be.ptr(self._normsq), be.ptr(self._sumsq_ws), self._sumsq_ws.numel(), be.stream()), "vdk_sam_first_step")
            eng.refresh_weights()
            self._fwd_loss_bwd(x, y, y_b, lam)
            self.big.copy_(self._old)
            sgd(None)
            self._loss_rows = loss_first
            return self._loss_rows
        self._fwd_loss_bwd(x, y, y_b, lam)
        if not self.amp:
            be.check(be.lib.vdk_sumsq_f32(be.ptr(self.gbig), self.n_total, be.ptr(self._normsq), be.ptr(self._sumsq_ws), self._sumsq_ws.numel(), be.stream()), "vdk_sumsq_f32")
        sgd(be.ptr(self._normsq))
        return self._loss_rows

    def loss_scale(self) -> float:
        return float(self.loss_state[0].item()) if self.amp else 1.0

    def skipped_steps(self) -> int:
        return int(self.loss_state[2].item()) if self.amp else 0

    def loss_value(self) -> float:
        return float(self._loss_rows.mean().item())

