"""TEST INFRASTRUCTURE: the oracle's 16-bit-operand arithmetic modes -- "bf16 operands, fp32 accumulate" and "fp16 operands, fp32 accumulate".

The reference trains its classifier under autocast (/root/reference engine/procedure/train.py:118, `with torch.autocast(device_type=self.device.type,
enabled=(self.device != cpu))`): no dtype argument, so on a GPU the autocast dtype is torch's default for the device type, **float16**, and the loop wraps the backward
in a GradScaler (train.py:205-211, engine/vision_engine.py:232).  Every matmul-shaped op (Linear, conv-as-GEMM, the two attention products) then reads fp16 operands and
accumulates in fp32; everything else (LayerNorm, softmax, GELU, residual adds, the loss) is fp32 arithmetic on fp32 values.  (Rounds 1-3 of this repo described that path
as bf16: wrong for the reference, right for BASELINE.json's configs[1], which names bf16.  Both formats are restated here and both are engine modes.)

The fp32 oracle (autocast off, what the reference does on CPU) cannot tell an engine that rounds its GEMM operands to 16 bits from one that is simply wrong by that
much; these modes can: they restate the SAME network with a rounding to the chosen format at exactly the tensors autocast holds in 16 bits, so that what is left between
the oracle and the HIP engine is fp32 summation order plus the rare element whose rounding flips.

mode "fp16_operands": rounding to IEEE half (10 mantissa bits: 8x finer than bf16).  Gradients are carried through the narrow exponent by a loss scale exactly as
GradScaler does: callers multiply the loss by S before backward() and divide the gradients by S afterwards (tests/, oracle/parity.py); overflow to inf is the caller's to
detect, as in the reference.

Rounding points (forward -> the same tensors' gradients are rounded on the way back, like autocast's 16-bit outputs):
  * Linear / conv-as-GEMM: both operands;  dgrad and wgrad: dY rounded, db = column sums of the rounded dY
  * `q(x)` boundaries: LayerNorm outputs, the fused qkv projection's output, the attention output
  * GELU: output rounded; the backward multiplies by gelu'(r(u)) and rounds the product (the pre-activation is KEPT in 16 bits)
  * attention: softmax probabilities (normalised, fp32) are rounded once as the left operand of P.V; backward re-derives
    P from the row log-sum-exp, rounds P for dV and dS = P*(dP - D) for dQ / dK; D = rowsum(dO * O) on the rounded tensors
Nothing here is imported by the product (`visiondk_amd/`); tests/ and bench.py's parity leg only.
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.nn.functional as F

_MODE = "fp32"      # "fp32" (reference on CPU: autocast off) | "bf16_operands" | "fp16_operands" (the reference under autocast on a GPU, restated)
_RDT = {"bf16_operands": torch.bfloat16, "fp16_operands": torch.float16}


def mode() -> str:
    return _MODE


@contextlib.contextmanager
def precision(m: str):
    global _MODE
    assert m in ("fp32", "bf16_operands", "fp16_operands"), m
    old, _MODE = _MODE, m
    try:
        yield
    finally:
        _MODE = old


def rb(t: torch.Tensor) -> torch.Tensor:
    """round to the mode's 16-bit format (nearest even; bfloat16 outside a 16-bit mode, as before) and back to the tensor's own dtype (float32; float64 when the oracle
    itself is run in double precision to measure how much of a deviation is the fp32 summation order of the oracle, tests/test_parity_bf16.py)"""
    return t.to(_RDT.get(_MODE, torch.bfloat16)).to(t.dtype)


class _Boundary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


def q(x: torch.Tensor) -> torch.Tensor:
    """a tensor autocast holds in bf16: value rounded going forward, gradient rounded coming back"""
    return _Boundary.apply(x) if _MODE != "fp32" else x


class _LinearBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, bias_grad_unrounded):
        xr, wr = rb(x), rb(w)
        ctx.save_for_backward(xr, wr)
        ctx.has_bias = b is not None
        ctx.bias_grad_unrounded = bias_grad_unrounded
        y = xr.reshape(-1, xr.shape[-1]) @ wr.t()
        if b is not None:
            y = y + b
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = rb(dy)
        dy2 = dyr.reshape(-1, dyr.shape[-1])
        dx = (dy2 @ wr).reshape(xr.shape)
        dw = dy2.t() @ xr.reshape(-1, xr.shape[-1])
        db = None
        if ctx.has_bias:
            src = dy.reshape(-1, dy.shape[-1]) if ctx.bias_grad_unrounded else dy2
            db = src.sum(0)
        return dx, dw, db, None


def linear(x, w, b=None, bias_grad_unrounded: bool = False):
    if _MODE == "fp32":
        return F.linear(x, w, b)
    return _LinearBf16.apply(x, w, b, bias_grad_unrounded)


def _gelu_grad(u):
    cdf = 0.5 * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))
    return cdf + u * pdf


class _GeluBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(rb(u))
        return rb(F.gelu(u))

    @staticmethod
    def backward(ctx, dg):
        (ur,) = ctx.saved_tensors
        return rb(dg * _gelu_grad(ur))


def gelu(u):
    return F.gelu(u) if _MODE == "fp32" else _GeluBf16.apply(u)


class _AttentionBf16(torch.autograd.Function):
    """softmax(scale * q k^T) v for [B, H, N, hd] tensors that are already bf16-valued (q() boundaries upstream)."""

    @staticmethod
    def forward(ctx, qh, kh, vh, scale):
        s = (qh @ kh.transpose(-2, -1)) * scale
        lse = torch.logsumexp(s, dim=-1, keepdim=True)
        p = torch.exp(s - lse)
        o = rb(rb(p) @ vh)
        ctx.save_for_backward(qh, kh, vh, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        qh, kh, vh, o, lse = ctx.saved_tensors
        scale = ctx.scale
        do = rb(do)
        p = torch.exp((qh @ kh.transpose(-2, -1)) * scale - lse)
        dv = rb(p).transpose(-2, -1) @ do
        dp = do @ vh.transpose(-2, -1)
        d = (do * o).sum(-1, keepdim=True)
        ds = rb(p * (dp - d))
        dq = (ds @ kh) * scale
        dk = (ds.transpose(-2, -1) @ qh) * scale
        return dq, dk, dv, None


def attention(qh, kh, vh, scale: float):
    if _MODE == "fp32":
        attn = (qh * scale) @ kh.transpose(-2, -1)
        return attn.softmax(dim=-1) @ vh
    return _AttentionBf16.apply(qh, kh, vh, scale)
