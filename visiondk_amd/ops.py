"""Thin per-op Python wrappers over the C ABI (device tensors in, device tensors out).

These exist for the parity tests and for host code that composes ops outside the native ViT engine; the
training step itself is orchestrated in C++ (csrc/vit_engine.hip).  No arithmetic happens in Python/torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi, _lib
from ._abi import ACT_DGELU, ACT_GELU, ACT_NONE  # noqa: F401


def _be(backend):
    return backend or _lib.load()


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, out_dtype=torch.bfloat16, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, aux: Optional[torch.Tensor] = None,
            alpha: float = 1.0, splitk: int = 1, out: Optional[torch.Tensor] = None, backend=None) -> torch.Tensor:
    """out[M,N] = epilogue(alpha * a[M,K] @ b[N,K].T); a, b bf16 (row stride may exceed K)."""
    be = _be(backend)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1 and a.shape[1] == b.shape[1]
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    d = _abi.GemmDesc()
    d.A, d.lda = a.data_ptr(), a.stride(0)
    d.B, d.ldb = b.data_ptr(), b.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.c_dtype = _abi.F32_ if out.dtype == torch.float32 else _abi.BF16
    d.bias = be.ptr(bias) if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act = act
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = aux.stride(0) if aux is not None else 0
    d.alpha = alpha
    d.splitk = splitk
    for t in (a, b, out, residual, aux):
        if t is not None and be.device_only and not t.is_cuda:
            raise RuntimeError("visiondk_amd: HIP backend got a CPU tensor (there is no CPU fallback)")
    ws = None
    nbytes = 0
    if splitk > 1:
        need = C.c_size_t(0)
        be.check(be.lib.vdk_gemm_splitk_workspace_bytes(M, N, splitk, C.byref(need)), "vdk_gemm_splitk_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=a.device)
        nbytes = need.value
    be.check(be.lib.vdk_gemm_bf16_nt(C.byref(d), be.ptr(ws), nbytes, be.stream()), "vdk_gemm_bf16_nt")
    return out


def transpose_pad(x: torch.Tensor, rpad: Optional[int] = None, backend=None) -> torch.Tensor:
    """x bf16 [R, C] -> [C, Rpad] with zero-filled padding columns (Rpad even, default: R rounded up to 64)."""
    be = _be(backend)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    if rpad is None:
        rpad = (R + 63) // 64 * 64
    out = torch.empty((Cc, rpad), dtype=torch.bfloat16, device=x.device)
    be.check(be.lib.vdk_transpose_bf16(be.ptr(x) if x.is_contiguous() else x.data_ptr(), x.stride(0), R, Cc,
                                       be.ptr(out), rpad, rpad, be.stream()), "vdk_transpose_bf16")
    return out
