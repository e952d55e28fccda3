"""Thin per-op Python wrappers over the C ABI (device tensors in, device tensors out).

These exist for the parity tests and for host code that composes ops outside the native ViT engine; the
training step itself is orchestrated in C++ (csrc/vit_engine.hip).  No arithmetic happens in Python/torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi, _lib
from ._abi import ACT_DGELU, ACT_GELU, ACT_GELU_SAVE_GRAD, ACT_MUL_AUX, ACT_NONE  # noqa: F401


def _be(backend):
    return backend or _lib.load()


_H16 = (torch.bfloat16, torch.float16)      # the two 16-bit operand formats of the GEMM path (VdkGemmDesc.ab_dtype / VdkVitConfig.operand)


def _dt(dtype) -> int:
    """torch dtype -> the ABI's dtype code"""
    return {torch.bfloat16: _abi.BF16, torch.float16: _abi.F16_, torch.float32: _abi.F32_}[dtype]


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, out_dtype=None, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, aux: Optional[torch.Tensor] = None,
            alpha: float = 1.0, splitk: int = 1, out: Optional[torch.Tensor] = None, row_group: int = 0,
            trans: bool = False, a_row_group: int = 0, a_rows: Optional[int] = None, a_colsum: Optional[torch.Tensor] = None,
            streamk_ws: Optional[torch.Tensor] = None, c_colsum: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None, rows_per_scale: int = 0, backend=None) -> torch.Tensor:
    """out[M,N] = epilogue(alpha * a[M,K] @ b[N,K].T); a, b bf16 or (both) fp16 (row stride may exceed K); out_dtype: fp32 or the operands' format (the default).
    streamk_ws: persistent workspace from streamk_workspace() -> stream-K allowed"""
    be = _be(backend)
    assert a.dtype in _H16 and b.dtype == a.dtype and a.dim() == 2 and b.dim() == 2
    out_dtype = a.dtype if out_dtype is None else out_dtype
    assert out_dtype in (torch.float32, a.dtype) and (aux is None or aux.dtype in _H16)
    assert a.stride(1) == 1 and b.stride(1) == 1 and (trans or a.shape[1] == b.shape[1])
    if trans:   # a: [K(+), M], b: [K(+), N]
        K, M = (a_rows if a_rows is not None else a.shape[0]), a.shape[1]
        N = b.shape[1]
    else:
        M, K = a.shape
        N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    d = _abi.GemmDesc()
    d.A, d.lda = a.data_ptr(), a.stride(0)
    d.B, d.ldb = b.data_ptr(), b.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.c_dtype = _dt(out.dtype)
    d.ab_dtype = _dt(a.dtype)
    d.bias = be.ptr(bias) if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act = act
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = aux.stride(0) if aux is not None else 0
    d.alpha = alpha
    d.splitk = splitk
    d.row_group = row_group
    d.trans = int(trans)
    d.a_row_group = a_row_group
    d.a_colsum = a_colsum.data_ptr() if a_colsum is not None else None      # f32 [vdk_gemm_a_colsum_rows(M, N, K), K] by-product (sum rows -> colsum(a))
    d.row_scale = be.ptr(row_scale) if row_scale is not None else None       # f32 [ceil(M / rows_per_scale)]: residual + row_scale[m // rows_per_scale] * (acc + bias) (stochastic depth)
    d.rows_per_scale = rows_per_scale
    d.col_scale = be.ptr(col_scale) if col_scale is not None else None       # f32 [N]: acc * col_scale before bias / residual (VdkGemmDesc.col_scale)
    d.c_colsum = c_colsum.data_ptr() if c_colsum is not None else None      # f32 [vdk_gemm_c_colsum_rows(M, N, K), N] by-product (sum rows -> colsum(stored bf16 out))
    for t in (a, b, out, residual, aux):
        if t is not None and be.device_only and not t.is_cuda:
            raise RuntimeError("visiondk_amd: HIP backend got a CPU tensor (there is no CPU fallback)")
    ws = None
    nbytes = 0
    if streamk_ws is not None:
        assert splitk <= 1
        d.splitk = -1
        ws, nbytes = streamk_ws, streamk_ws.numel()
    if splitk > 1:
        need = C.c_size_t(0)
        be.check(be.lib.vdk_gemm_splitk_workspace_bytes(M, N, splitk, C.byref(need)), "vdk_gemm_splitk_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=a.device)
        nbytes = need.value
    be.check(be.lib.vdk_gemm_bf16_nt(C.byref(d), be.ptr(ws), nbytes, be.stream()), "vdk_gemm_bf16_nt")
    return out


def streamk_workspace(device, backend=None) -> torch.Tensor:
    """persistent stream-K workspace (zeroed tile counters + accumulator slabs) for gemm_nt(streamk_ws=...); one per stream"""
    be = _be(backend)
    need = C.c_size_t(0)
    be.check(be.lib.vdk_gemm_streamk_workspace_bytes(C.byref(need)), "vdk_gemm_streamk_workspace_bytes")
    return torch.zeros(need.value, dtype=torch.uint8, device=device)


FP8_E4M3, FP8_E5M2 = 0, 1


def quant_fp8(x: torch.Tensor, scale: Optional[torch.Tensor] = None, fmt: int = FP8_E4M3, amax: Optional[torch.Tensor] = None, backend=None) -> torch.Tensor:
    """x (bf16 | f32, numel % 16 == 0) -> uint8 tensor of OCP fp8 bytes = fp8(clamp(x * scale[0])); amax[0] (f32 device scalar) accumulates max |x| (delayed scaling)"""
    be = _be(backend)
    assert x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32) and x.numel() % 16 == 0
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    be.check(be.lib.vdk_quant_fp8(be.ptr(x), _abi.BF16 if x.dtype == torch.bfloat16 else _abi.F32_, x.numel(), be.ptr(scale) if scale is not None else None, be.ptr(out), fmt,
                                  be.ptr(amax) if amax is not None else None, be.stream()), "vdk_quant_fp8")
    return out


def fp8_scale_update(amax: torch.Tensor, scale: torch.Tensor, scale_inv: torch.Tensor, fmt: int = FP8_E4M3, margin: float = 1.0, backend=None) -> None:
    """delayed scaling bookkeeping for n tensors at once: scale = fmt_max / (margin * amax), scale_inv = 1 / scale (unchanged where amax == 0), amax := 0"""
    be = _be(backend)
    be.check(be.lib.vdk_fp8_scale_update(be.ptr(amax), be.ptr(scale), be.ptr(scale_inv), amax.numel(), fmt, margin, be.stream()), "vdk_fp8_scale_update")


def gemm_fp8_nt(a8: torch.Tensor, b8: torch.Tensor, a_scale_inv: Optional[torch.Tensor] = None, b_scale_inv: Optional[torch.Tensor] = None, *, a_fmt: int = FP8_E4M3,
                out_dtype=torch.bfloat16, bias=None, residual=None, act: int = ACT_NONE, aux=None, backend=None, q8: Optional[dict] = None,
                c_colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(a_scale_inv * b_scale_inv * a8[M,K] @ b8[N,K].T): uint8 tensors of fp8 bytes (a: e4m3 or e5m2, b: e4m3), fp32 accumulation on the scaled MFMA.
    q8 = {"fmt": FP8_E4M3 | FP8_E5M2, "scale": f32[1] | None, "amax": f32[1] | None}: the epilogue also writes the fp8 quantisation of `out` (GELU / DGELU forms) and
    the call returns (out, out8) -- bit-identical to quant_fp8(out, ...)."""
    be = _be(backend)
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.dim() == 2 and b8.dim() == 2 and a8.shape[1] == b8.shape[1]
    M, K = a8.shape
    N = b8.shape[0]
    out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    d = _abi.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a8.data_ptr(), a8.stride(0), b8.data_ptr(), b8.stride(0), out.data_ptr(), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.c_dtype = _abi.F32_ if out_dtype == torch.float32 else _abi.BF16
    d.bias = be.ptr(bias) if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act = act
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = aux.stride(0) if aux is not None else 0
    d.alpha, d.splitk = 1.0, 1
    if c_colsum is not None:       # f32 [2 * ceil(M / 256), N] partial column sums of the stored bf16 output (dGELU form): sum(0) = the bias gradient of the Linear before
        assert c_colsum.dtype == torch.float32 and c_colsum.is_contiguous() and c_colsum.shape == (2 * ((M + 255) // 256), N)
        d.c_colsum = be.ptr(c_colsum)
    for t in (a8, b8, out, residual, aux):
        if t is not None and be.device_only and not t.is_cuda:
            raise RuntimeError("visiondk_amd: HIP backend got a CPU tensor (there is no CPU fallback)")
    if q8 is not None:
        out8 = torch.empty((M, N), dtype=torch.uint8, device=a8.device)
        be.check(be.lib.vdk_gemm_fp8_nt_q8(C.byref(d), a_fmt, FP8_E4M3, be.ptr(a_scale_inv) if a_scale_inv is not None else None,
                                           be.ptr(b_scale_inv) if b_scale_inv is not None else None, out8.data_ptr(), out8.stride(0), int(q8.get("fmt", FP8_E4M3)),
                                           be.ptr(q8.get("scale")), be.ptr(q8.get("amax")), be.stream()), "vdk_gemm_fp8_nt_q8")
        return out, out8
    be.check(be.lib.vdk_gemm_fp8_nt(C.byref(d), a_fmt, FP8_E4M3, be.ptr(a_scale_inv) if a_scale_inv is not None else None,
                                    be.ptr(b_scale_inv) if b_scale_inv is not None else None, be.stream()), "vdk_gemm_fp8_nt")
    return out


def transpose_pad(x: torch.Tensor, rpad: Optional[int] = None, backend=None, rows: Optional[int] = None,
                  row_group: int = 0, colsum_partial: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 [R, C] -> [C, Rpad] with zero-filled padding columns (Rpad even, default: R rounded up to 64).  An fp16 tensor moves as its 16-bit patterns (no arithmetic
    without colsum_partial) and comes back as fp16."""
    be = _be(backend)
    if x.dtype == torch.float16:
        assert colsum_partial is None
        return transpose_pad(x.view(torch.bfloat16), rpad, backend, rows, row_group).view(torch.float16)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    if rows is not None:
        R = rows
    if rpad is None:
        rpad = (R + 63) // 64 * 64
    out = torch.empty((Cc, rpad), dtype=torch.bfloat16, device=x.device)
    be.check(be.lib.vdk_transpose_bf16(be.ptr(x) if x.is_contiguous() else x.data_ptr(), x.stride(0), R, Cc,
                                       be.ptr(out), rpad, rpad, row_group, be.ptr(colsum_partial), be.stream()), "vdk_transpose_bf16")
    return out


def _ws(be, fn, *args, device):
    need = C.c_size_t(0)
    be.check(fn(*args, C.byref(need)), fn.__name__)
    return torch.empty(max(need.value, 16), dtype=torch.uint8, device=device), need.value


def attention_fwd(qkv: torch.Tensor, heads: int, scale: Optional[float] = None, backend=None):
    """qkv bf16 [B, N, 3*heads*64] -> (o bf16 [B, N, heads*64], lse f32 [B, heads, N])."""
    be = _be(backend)
    B, N, three_d = qkv.shape
    D = three_d // 3
    hd = D // heads
    scale = hd ** -0.5 if scale is None else scale
    assert qkv.dtype in _H16
    o = torch.empty((B, N, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    be.check(be.lib.vdk_attention_fwd_dt(be.ptr(qkv), three_d, be.ptr(o), D, be.ptr(lse), B, N, heads, hd, scale, _dt(qkv.dtype), be.stream()),
             "vdk_attention_fwd")
    return o, lse


def attention_bwd(qkv, o, dout, lse, heads: int, scale: Optional[float] = None, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, N, three_d = qkv.shape
    D = three_d // 3
    hd = D // heads
    scale = hd ** -0.5 if scale is None else scale
    dqkv = torch.empty_like(qkv)
    dvec = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    dout = dout.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    assert qkv.dtype in _H16 and o.dtype == qkv.dtype and dout.dtype == qkv.dtype
    be.check(be.lib.vdk_attention_bwd_dt(be.ptr(qkv), three_d, be.ptr(o), be.ptr(dout), D, be.ptr(lse), be.ptr(dqkv),
                                         three_d, be.ptr(dvec), B, N, heads, hd, scale, _dt(qkv.dtype), be.stream()), "vdk_attention_bwd")
    return dqkv


def layernorm_fwd(x: torch.Tensor, gamma, beta, eps: float = 1e-6, out_dtype=torch.bfloat16, rows: Optional[int] = None,
                  ldx: Optional[int] = None, backend=None):
    """x f32 [T, C] (or a strided row view via rows/ldx) -> (y, mean, rstd)."""
    be = _be(backend)
    C_ = gamma.numel()
    T = rows if rows is not None else x.numel() // C_
    ldx = ldx if ldx is not None else C_
    y = torch.empty((T, C_), dtype=out_dtype, device=x.device)
    mean = torch.empty(T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_layernorm_fwd(be.ptr(x), ldx, T, C_, be.ptr(gamma), be.ptr(beta), eps, be.ptr(y), C_,
                                      _dt(out_dtype), be.ptr(mean), be.ptr(rstd),
                                      be.stream()), "vdk_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_q8(x: torch.Tensor, gamma, beta, scale: Optional[torch.Tensor], amax: Optional[torch.Tensor], fmt: int = FP8_E4M3, eps: float = 1e-6, backend=None):
    """x f32 [T, C], 128 < C <= 1024 -> (y bf16, y8 uint8 = quant_fp8(y, scale, fmt) bit for bit, mean, rstd); amax[0] accumulates max |y| (the fp8 mode of the ViT engine)"""
    be = _be(backend)
    C_ = gamma.numel()
    T = x.numel() // C_
    y = torch.empty((T, C_), dtype=torch.bfloat16, device=x.device)
    y8 = torch.empty((T, C_), dtype=torch.uint8, device=x.device)
    mean = torch.empty(T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_layernorm_fwd_q8(be.ptr(x), C_, T, C_, be.ptr(gamma), be.ptr(beta), eps, be.ptr(y), C_, be.ptr(mean), be.ptr(rstd), be.ptr(y8), C_, fmt,
                                         be.ptr(scale), be.ptr(amax), be.stream()), "vdk_layernorm_fwd_q8")
    return y, y8, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, want_bf16=True, backend=None):
    """-> (dx f32 [T,C], dx in dy's 16-bit format (bf16 for an fp32 dy) or None, dgamma, dbeta)"""
    be = _be(backend)
    C_ = gamma.numel()
    T = mean.numel()
    dx = torch.empty((T, C_), dtype=torch.float32, device=x.device)
    dxb = torch.empty((T, C_), dtype=dy.dtype if dy.dtype in _H16 else torch.bfloat16, device=x.device) if want_bf16 else None
    dg = torch.empty(C_, dtype=torch.float32, device=x.device)
    db = torch.empty(C_, dtype=torch.float32, device=x.device)
    ws, n = _ws(be, be.lib.vdk_layernorm_bwd_workspace_bytes, T, C_, device=x.device)
    be.check(be.lib.vdk_layernorm_bwd(be.ptr(dy), C_, _dt(dy.dtype), be.ptr(x), C_,
                                      be.ptr(mean), be.ptr(rstd), be.ptr(gamma), be.ptr(dres), C_, T, C_, be.ptr(dx), C_,
                                      be.ptr(dxb), C_, be.ptr(dg), be.ptr(db), be.ptr(ws), n, be.stream()), "vdk_layernorm_bwd")
    return dx, dxb, dg, db


def reduce_rows(x: torch.Tensor, scale: float = 1.0, backend=None) -> torch.Tensor:
    be = _be(backend)
    S, n = x.shape[0], x[0].numel()
    out = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_reduce_rows_f32(be.ptr(x), n, S, n, be.ptr(out), scale, be.stream()), "vdk_reduce_rows_f32")
    return out


def colsum_bf16(x: torch.Tensor, backend=None) -> torch.Tensor:
    be = _be(backend)
    T, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws, n = _ws(be, be.lib.vdk_colsum_bf16_workspace_bytes, T, N, device=x.device)
    be.check(be.lib.vdk_colsum_bf16(be.ptr(x), N, T, N, be.ptr(out), be.ptr(ws), n, be.stream()), "vdk_colsum_bf16")
    return out


def softmax_ce(logits, ya, yb=None, lam: float = 1.0, label_smoothing: float = 0.0, grad_scale: float = 1.0,
               pad_to: Optional[int] = None, backend=None, dl_dtype=torch.bfloat16, loss_scale: Optional[torch.Tensor] = None):
    """-> (loss_rows f32 [B], dlogits (dl_dtype: bf16 | fp16) [B, pad_to], dlogits f32 [B, C]); loss_scale: f32 device scalar multiplied into both gradients (GradScaler)"""
    be = _be(backend)
    B, C_ = logits.shape
    pad_to = pad_to or (C_ + 7) // 8 * 8
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlb = torch.empty((B, pad_to), dtype=dl_dtype, device=logits.device)
    dlf = torch.empty((B, C_), dtype=torch.float32, device=logits.device)
    be.check(be.lib.vdk_softmax_ce_amp(be.ptr(logits), C_, B, C_, be.ptr(ya), be.ptr(yb), lam, label_smoothing, grad_scale, be.ptr(loss_scale),
                                       be.ptr(loss), be.ptr(dlb), pad_to, _dt(dl_dtype), be.ptr(dlf), C_, be.stream()), "vdk_softmax_ce")
    return loss, dlb, dlf


def bce_logits(logits, targets, grad_scale: float = 1.0, focal_gamma: float = 0.0, focal_alpha: float = 0.25, backend=None):
    be = _be(backend)
    B, C_ = logits.shape
    pad_to = (C_ + 7) // 8 * 8
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlb = torch.empty((B, pad_to), dtype=torch.bfloat16, device=logits.device)
    dlf = torch.empty((B, C_), dtype=torch.float32, device=logits.device)
    be.check(be.lib.vdk_bce_logits(be.ptr(logits), C_, be.ptr(targets), C_, B, C_, grad_scale, focal_gamma, focal_alpha, be.ptr(loss), be.ptr(dlb), pad_to,
                                   be.ptr(dlf), C_, be.stream()), "vdk_bce_logits")
    return loss, dlb, dlf


def patchify(x: torch.Tensor, patch: int, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, Cin, H, W = x.shape
    K = Cin * patch * patch
    Kp = (K + 7) // 8 * 8
    out = torch.empty((B * (H // patch) * (W // patch), Kp), dtype=torch.bfloat16, device=x.device)
    be.check(be.lib.vdk_patchify_bf16(be.ptr(x), B, Cin, H, W, patch, be.ptr(out), Kp, be.stream()), "vdk_patchify_bf16")
    return out


def cast_bf16(x: torch.Tensor, backend=None) -> torch.Tensor:
    be = _be(backend)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    be.check(be.lib.vdk_cast_f32_bf16(be.ptr(x), be.ptr(out), x.numel(), be.stream()), "vdk_cast_f32_bf16")
    return out


def cast_16(x: torch.Tensor, dtype=torch.bfloat16, backend=None) -> torch.Tensor:
    """x f32 -> the 16-bit operand format `dtype` (torch.bfloat16 | torch.float16)"""
    if dtype == torch.bfloat16:
        return cast_bf16(x, backend=backend)
    assert dtype == torch.float16
    be = _be(backend)
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    be.check(be.lib.vdk_cast_f32_f16(be.ptr(x), be.ptr(out), x.numel(), be.stream()), "vdk_cast_f32_f16")
    return out


def scale_dev(x: torch.Tensor, scale: torch.Tensor, reciprocal: bool = False, backend=None) -> torch.Tensor:
    """x (f32, contiguous) *= scale[0] in place, the factor read on the device (vdk_scale_dev_f32: GradScaler's loss scale entering a gradient tensor)"""
    be = _be(backend)
    assert x.dtype == torch.float32 and x.is_contiguous() and scale.dtype == torch.float32
    be.check(be.lib.vdk_scale_dev_f32(be.ptr(x), x.numel(), be.ptr(scale), int(reciprocal), be.stream()), "vdk_scale_dev_f32")
    return x


def transpose_cast(w: torch.Tensor, rpad: Optional[int] = None, backend=None) -> torch.Tensor:
    """w f32 [R, C] -> bf16 [C, Rpad]"""
    be = _be(backend)
    R, Cc = w.shape
    rpad = rpad or (R + 7) // 8 * 8
    out = torch.empty((Cc, rpad), dtype=torch.bfloat16, device=w.device)
    be.check(be.lib.vdk_transpose_cast_f32_bf16(be.ptr(w), Cc, R, Cc, be.ptr(out), rpad, rpad, be.stream()),
             "vdk_transpose_cast_f32_bf16")
    return out


def sumsq(g: torch.Tensor, backend=None) -> torch.Tensor:
    be = _be(backend)
    out = torch.empty(1, dtype=torch.float32, device=g.device)
    ws, n = _ws(be, be.lib.vdk_sumsq_workspace_bytes, device=g.device)
    be.check(be.lib.vdk_sumsq_f32(be.ptr(g), g.numel(), be.ptr(out), be.ptr(ws), n, be.stream()), "vdk_sumsq_f32")
    return out


def sgd_step(p, g, m, *, lr, momentum, weight_decay, ema=None, p_bf16=None, grad_scale=1.0, normsq=None, max_norm=10.0,
             ema_decay=0.0, first_step=False, backend=None) -> None:
    be = _be(backend)
    be.check(be.lib.vdk_sgd_step(be.ptr(p), be.ptr(g), be.ptr(m), be.ptr(ema), be.ptr(p_bf16), p.numel(), lr, momentum,
                                 weight_decay, grad_scale, be.ptr(normsq), max_norm, ema_decay, int(first_step), be.stream()),
             "vdk_sgd_step")


def mixup(x: torch.Tensor, perm: torch.Tensor, lam: float, backend=None) -> torch.Tensor:
    be = _be(backend)
    out = torch.empty_like(x)
    be.check(be.lib.vdk_mixup(be.ptr(x), be.ptr(perm), lam, x.shape[0], x[0].numel(), be.ptr(out), be.stream()), "vdk_mixup")
    return out


def sam_first_step(p, g, old_p, rho: float = 0.05, adaptive: bool = True, backend=None) -> torch.Tensor:
    be = _be(backend)
    nsq = torch.empty(1, dtype=torch.float32, device=p.device)
    ws, n = _ws(be, be.lib.vdk_sumsq_workspace_bytes, device=p.device)
    be.check(be.lib.vdk_sam_first_step(be.ptr(p), be.ptr(g), be.ptr(old_p), p.numel(), rho, int(adaptive), be.ptr(nsq), be.ptr(ws), n, be.stream()),
             "vdk_sam_first_step")
    return nsq


def ohem_mask(logits, labels, min_kept: int, thresh: float, ignore_index: int = 255, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, C_ = logits.shape
    prob = torch.empty(B, dtype=torch.float32, device=logits.device)
    mask = torch.empty(B, dtype=torch.uint8, device=logits.device)
    be.check(be.lib.vdk_ohem_mask(be.ptr(logits), C_, B, C_, be.ptr(labels), min_kept, thresh, ignore_index, be.ptr(prob), be.ptr(mask), be.stream()),
             "vdk_ohem_mask")
    return mask.bool()


def topk_rows(x, k: int, backend=None):
    be = _be(backend)
    B, C_ = x.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=x.device)
    val = torch.empty((B, k), dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_topk_rows(be.ptr(x), C_, B, C_, k, be.ptr(idx), be.ptr(val), be.stream()), "vdk_topk_rows")
    return val, idx


# ---- CNN backbone pieces (csrc/conv.hip) -------------------------------------------------------------------------------
def dwconv7_weight_prep(w: torch.Tensor, backend=None) -> torch.Tensor:
    """conv_dw.weight [C,1,7,7] f32 -> tap-major [49, C] f32"""
    be = _be(backend)
    Cc = w.shape[0]
    w = w.reshape(Cc, 49).contiguous()
    wt = torch.empty((49, Cc), dtype=torch.float32, device=w.device)
    be.check(be.lib.vdk_dwconv7_weight_prep(be.ptr(w), be.ptr(wt), Cc, be.stream()), "vdk_dwconv7_weight_prep")
    return wt


def dwconv7(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, *, flip: bool = False,
            want_bf16: bool = False, backend=None):
    """x f32 [B,H,W,C] (NHWC); wt [49,C]; returns f32 [B,H,W,C] (and a bf16 copy if want_bf16)."""
    be = _be(backend)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    outb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    be.check(be.lib.vdk_dwconv7_fwd(be.ptr(x), be.ptr(wt), be.ptr(bias) if bias is not None else None, be.ptr(res) if res is not None else None,
                                    be.ptr(out), be.ptr(outb) if outb is not None else None, B, H, W, Cc, int(flip), be.stream()), "vdk_dwconv7_fwd")
    return (out, outb) if want_bf16 else out


def dwconv7_wgrad(x: torch.Tensor, dy: torch.Tensor, backend=None):
    """-> (dw f32 [C,49], db f32 [C])"""
    be = _be(backend)
    B, H, W, Cc = x.shape
    need = C.c_size_t(0)
    be.check(be.lib.vdk_dwconv7_wgrad_workspace_bytes(B, H, W, Cc, C.byref(need)), "vdk_dwconv7_wgrad_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
    dw = torch.empty((Cc, 49), dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_dwconv7_wgrad(be.ptr(x), be.ptr(dy), be.ptr(dw), be.ptr(db), B, H, W, Cc, be.ptr(ws), ws.numel(), be.stream()), "vdk_dwconv7_wgrad")
    return dw, db


def space_to_depth2(x: torch.Tensor, inverse: bool = False, backend=None) -> torch.Tensor:
    """bf16 [B,H,W,C] -> [B,H/2,W/2,4C] (k = (2*(y&1)+(x&1))*C + c); inverse: the other way round"""
    be = _be(backend)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if inverse:
        B, h, w, c4 = x.shape
        out = torch.empty((B, 2 * h, 2 * w, c4 // 4), dtype=torch.bfloat16, device=x.device)
        B, H, W, Cc = out.shape
    else:
        B, H, W, Cc = x.shape
        out = torch.empty((B, H // 2, W // 2, 4 * Cc), dtype=torch.bfloat16, device=x.device)
    be.check(be.lib.vdk_space_to_depth2_bf16(be.ptr(x), be.ptr(out), B, H, W, Cc, int(inverse), be.stream()), "vdk_space_to_depth2_bf16")
    return out


def conv2x2_weight_prep(w: torch.Tensor, backend=None):
    """Conv2d weight [Co,Ci,2,2] f32 -> (bf16 [Co,4Ci], bf16 [4Ci,Co])"""
    be = _be(backend)
    Co, Ci = w.shape[:2]
    w = w.contiguous()
    wb = torch.empty((Co, 4 * Ci), dtype=torch.bfloat16, device=w.device)
    wtb = torch.empty((4 * Ci, Co), dtype=torch.bfloat16, device=w.device)
    be.check(be.lib.vdk_conv2x2_weight_prep(be.ptr(w), be.ptr(wb), be.ptr(wtb), Co, Ci, be.stream()), "vdk_conv2x2_weight_prep")
    return wb, wtb


def conv2x2_wgrad_unpermute(dwp: torch.Tensor, Ci: int, backend=None) -> torch.Tensor:
    be = _be(backend)
    Co = dwp.shape[0]
    dw = torch.empty((Co, Ci, 2, 2), dtype=torch.float32, device=dwp.device)
    dwp = dwp.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_conv2x2_wgrad_unpermute(be.ptr(dwp), be.ptr(dw), Co, Ci, be.stream()), "vdk_conv2x2_wgrad_unpermute")
    return dw


def layerscale_weight_prep(w2: torch.Tensor, b2: torch.Tensor, gamma: torch.Tensor, backend=None):
    be = _be(backend)
    Cc, M = w2.shape
    w2p = torch.empty((Cc, M), dtype=torch.bfloat16, device=w2.device)
    w2pt = torch.empty((M, Cc), dtype=torch.bfloat16, device=w2.device)
    b2p = torch.empty(Cc, dtype=torch.float32, device=w2.device)
    w2 = w2.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_layerscale_weight_prep(be.ptr(w2), be.ptr(b2), be.ptr(gamma), be.ptr(w2p), be.ptr(w2pt), be.ptr(b2p), Cc, M, be.stream()),
             "vdk_layerscale_weight_prep")
    return w2p, w2pt, b2p


def layerscale_grad(dw2p: torch.Tensor, db2p: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, gamma: torch.Tensor, backend=None):
    be = _be(backend)
    Cc, M = w2.shape
    dw2 = torch.empty_like(w2); db2 = torch.empty_like(b2); dg = torch.empty_like(gamma)
    dw2p = dw2p.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    w2 = w2.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_layerscale_grad(be.ptr(dw2p), be.ptr(db2p), be.ptr(w2), be.ptr(b2), be.ptr(gamma), be.ptr(dw2), be.ptr(db2),
                                        be.ptr(dg), Cc, M, be.stream()), "vdk_layerscale_grad")
    return dw2, db2, dg


def batchnorm_fwd(x: torch.Tensor, gamma, beta, running_mean, running_var, *, training: bool, eps: float = 1e-5, momentum: float = 0.1, backend=None):
    """nn.BatchNorm1d on rows [B, F] (also BatchNorm2d on NHWC rows [B*H*W, C]) -> (y, save_mean, save_invstd); running stats updated in place"""
    be = _be(backend)
    B, F = x.shape
    y = torch.empty_like(x)
    sm = torch.empty(F, dtype=torch.float32, device=x.device); si = torch.empty(F, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_batchnorm1d_fwd(be.ptr(x), x.stride(0), B, F, be.ptr(gamma), be.ptr(beta), eps, momentum, int(training), be.ptr(running_mean),
                                        be.ptr(running_var), be.ptr(y), F, be.ptr(sm), be.ptr(si), be.stream()), "vdk_batchnorm1d_fwd")
    return y, sm, si


def batchnorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma, save_mean, save_invstd, backend=None):
    be = _be(backend)
    B, F = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(F, dtype=torch.float32, device=x.device); db = torch.empty(F, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_batchnorm1d_bwd(be.ptr(dy), dy.stride(0), be.ptr(x), x.stride(0), B, F, be.ptr(gamma), be.ptr(save_mean), be.ptr(save_invstd), be.ptr(dx),
                                        F, be.ptr(dg), be.ptr(db), be.stream()), "vdk_batchnorm1d_bwd")
    return dx, dg, db


# ---- precise (fp32 MFMA) path ---------------------------------------------------------------------------------------------
def gemm_f32(a: torch.Tensor, b: torch.Tensor, *, bias=None, residual=None, act: int = ACT_NONE, alpha: float = 1.0, b_kmajor: bool = False, a_kmajor: bool = False,
             k_splits: int = 1, backend=None):
    """out f32 [M, N] = epilogue(alpha * a[M, K] @ b[N, K].T)  (b_kmajor: b is [K, N]; a_kmajor: a is [K, M], with b_kmajor the weight-gradient form a.T @ b);
    fp32 operands, fp32 MFMA.  k_splits > 1: the contraction in that many slabs (a long K with few output tiles), summed in slab order; no epilogue then."""
    be = _be(backend)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    K, M = a.shape if a_kmajor else a.shape[::-1]
    N = b.shape[1] if b_kmajor else b.shape[0]
    if k_splits > 1:
        assert bias is None and residual is None and act == ACT_NONE and alpha == 1.0
        kc = ((K + k_splits - 1) // k_splits + 3) // 4 * 4
        S = (K + kc - 1) // kc
        slabs = torch.empty((S, M, N), dtype=torch.float32, device=a.device)
        d = _abi.GemmF32Desc()
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), slabs.data_ptr(), N
        d.M, d.N, d.K, d.alpha, d.b_kmajor, d.a_kmajor = M, N, kc, 1.0, int(b_kmajor), int(a_kmajor)
        d.batch1, d.batch2, d.sc1, d.k_total = S, 1, M * N, K
        d.sa1 = kc * a.stride(0) if a_kmajor else kc
        d.sb1 = kc * b.stride(0) if b_kmajor else kc
        be.check(be.lib.vdk_gemm_f32_nt(C.byref(d), be.stream()), "vdk_gemm_f32_nt")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        be.check(be.lib.vdk_reduce_rows_f32(be.ptr(slabs), M * N, S, M * N, be.ptr(out), 1.0, be.stream()), "vdk_reduce_rows_f32")
        return out
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    d = _abi.GemmF32Desc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N
    d.M, d.N, d.K = M, N, K
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act, d.alpha, d.b_kmajor, d.a_kmajor = act, alpha, int(b_kmajor), int(a_kmajor)
    be.check(be.lib.vdk_gemm_f32_nt(C.byref(d), be.stream()), "vdk_gemm_f32_nt")
    return out


def softmax_rows_f32(x: torch.Tensor, cols: int, scale: float = 1.0, backend=None) -> torch.Tensor:
    be = _be(backend)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    be.check(be.lib.vdk_softmax_rows_f32(be.ptr(x), x.shape[1], x.shape[0], cols, scale, be.stream()), "vdk_softmax_rows_f32")
    return x


# ---- BatchNorm-based CNN pieces (csrc/resnet_ops.hip + the implicit-GEMM convolution of csrc/gemm.hip) ----------------------------------------
def conv_weight_prep(w: torch.Tensor, cip: Optional[int] = None, backend=None):
    """w f32 [Co,Ci,KH,KW] -> (wf bf16 [Co, KH*KW*Cip], wd bf16 [Cip, KH*KW*Co])"""
    be = _be(backend)
    Co, Ci, KH, KW = w.shape
    cip = cip or (Ci + 7) // 8 * 8
    w = w.contiguous()
    wf = torch.empty((Co, KH * KW * cip), dtype=torch.bfloat16, device=w.device)
    wd = torch.empty((cip, KH * KW * Co), dtype=torch.bfloat16, device=w.device)
    be.check(be.lib.vdk_conv_weight_prep(be.ptr(w), be.ptr(wf), be.ptr(wd), Co, Ci, cip, KH, KW, be.stream()), "vdk_conv_weight_prep")
    return wf, wd


def conv_gemm(a_nhwc: torch.Tensor, wmat: torch.Tensor, *, oh: int, ow: int, kh: int, kw: int, stride: int, pad: int, transposed: bool = False,
              out_dtype=torch.float32, residual: Optional[torch.Tensor] = None, backend=None) -> torch.Tensor:
    """Implicit-GEMM convolution: a bf16 [B,H,W,Cin] gathered on the fly, wmat bf16 [N, kh*kw*Cin] -> [B*oh*ow, N]"""
    be = _be(backend)
    B, H, W, Cin = a_nhwc.shape
    N = wmat.shape[0]
    out = torch.empty((B * oh * ow, N), dtype=out_dtype, device=a_nhwc.device)
    geom = _abi.ConvGeom(Cin, H, W, oh, ow, kh, kw, stride, pad, int(transposed))
    d = _abi.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a_nhwc.data_ptr(), 0, wmat.data_ptr(), wmat.stride(0), out.data_ptr(), N
    d.M, d.N, d.K = B * oh * ow, N, kh * kw * Cin
    d.c_dtype = _abi.BF16 if out_dtype == torch.bfloat16 else _abi.F32_
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act, d.alpha, d.splitk = ACT_NONE, 1.0, 1
    d.conv = C.cast(C.pointer(geom), C.c_void_p)
    be.check(be.lib.vdk_gemm_bf16_nt(C.byref(d), None, 0, be.stream()), "vdk_gemm_bf16_nt(conv)")
    return out


def conv_wgrad_implicit(dy: torch.Tensor, a_nhwc: torch.Tensor, *, oh: int, ow: int, kh: int, kw: int, stride: int, pad: int, splitk: int = 1, backend=None) -> torch.Tensor:
    """dW' f32 [Co, kh*kw*Cin] = dy^T . im2col(a): dy bf16 [B*oh*ow, Co], a bf16 [B,H,W,Cin]; the im2col operand is gathered inside the TN GEMM (VdkConvGeom.rows)."""
    be = _be(backend)
    B, H, W, Cin = a_nhwc.shape
    rows, Co = dy.shape
    assert rows == B * oh * ow and dy.is_contiguous() and a_nhwc.is_contiguous()
    N = kh * kw * Cin
    out = torch.empty((Co, N), dtype=torch.float32, device=dy.device)
    geom = _abi.ConvGeom(Cin, H, W, oh, ow, kh, kw, stride, pad, 0, rows)
    d = _abi.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = dy.data_ptr(), Co, a_nhwc.data_ptr(), N, out.data_ptr(), N
    d.M, d.N, d.K = Co, N, (rows + 127) // 128 * 128
    d.c_dtype, d.act, d.alpha, d.splitk, d.trans = _abi.F32_, ACT_NONE, 1.0, splitk, 1
    d.conv = C.cast(C.pointer(geom), C.c_void_p)
    ws = torch.empty((max(splitk, 1) * Co * N + 64,), dtype=torch.float32, device=dy.device)
    be.check(be.lib.vdk_gemm_bf16_nt(C.byref(d), be.ptr(ws), ws.numel() * 4, be.stream()), "vdk_gemm_bf16_nt(conv wgrad)")
    return out


def im2col(a_nhwc: torch.Tensor, oh: int, ow: int, kh: int, kw: int, stride: int, pad: int, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, H, W, Cc = a_nhwc.shape
    col = torch.empty((B * oh * ow, kh * kw * Cc), dtype=torch.bfloat16, device=a_nhwc.device)
    be.check(be.lib.vdk_im2col_bf16(be.ptr(a_nhwc), be.ptr(col), B, H, W, Cc, oh, ow, kh, kw, stride, pad, be.stream()), "vdk_im2col_bf16")
    return col


def conv_wgrad_unpermute(dwp: torch.Tensor, ci: int, kh: int, kw: int, backend=None) -> torch.Tensor:
    be = _be(backend)
    Co = dwp.shape[0]
    cip = dwp.shape[1] // (kh * kw)
    dw = torch.empty((Co, ci, kh, kw), dtype=torch.float32, device=dwp.device)
    dwp = dwp.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_conv_wgrad_unpermute(be.ptr(dwp), be.ptr(dw), Co, ci, cip, kh, kw, be.stream()), "vdk_conv_wgrad_unpermute")
    return dw


def nchw_to_nhwc_bf16(x: torch.Tensor, cp: int, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, cp), dtype=torch.bfloat16, device=x.device)
    x = x.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_nchw_to_nhwc_bf16(be.ptr(x), be.ptr(out), B, Cc, H, W, cp, be.stream()), "vdk_nchw_to_nhwc_bf16")
    return out


def _bn_ws(be, R, Cc, device):
    need = C.c_size_t(0)
    be.check(be.lib.vdk_bn_rows_workspace_bytes(R, Cc, C.byref(need)), "vdk_bn_rows_workspace_bytes")
    return torch.empty(need.value, dtype=torch.uint8, device=device)


def bn_act_fwd(x: torch.Tensor, gamma, beta, running_mean, running_var, *, training=True, eps=1e-5, momentum=0.1, res=None, relu=True, want_f32=False, backend=None):
    """x f32 [R, C] -> (out bf16, out f32 | None, save_mean, save_invstd)"""
    be = _be(backend)
    R, Cc = x.shape
    ws = _bn_ws(be, R, Cc, x.device)
    outb = torch.empty((R, Cc), dtype=torch.bfloat16, device=x.device)
    outf = torch.empty((R, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    sm = torch.empty(Cc, dtype=torch.float32, device=x.device); si = torch.empty(Cc, dtype=torch.float32, device=x.device)
    rf = res if (res is not None and res.dtype == torch.float32) else None
    rb = res if (res is not None and res.dtype == torch.bfloat16) else None
    be.check(be.lib.vdk_bn_act_fwd(be.ptr(x), R, Cc, be.ptr(gamma), be.ptr(beta), eps, momentum, int(training), be.ptr(running_mean), be.ptr(running_var),
                                   be.ptr(rf) if rf is not None else None, be.ptr(rb) if rb is not None else None, int(relu), be.ptr(outb),
                                   be.ptr(outf) if outf is not None else None, be.ptr(sm), be.ptr(si), be.ptr(ws), ws.numel(), None, None, be.stream()), "vdk_bn_act_fwd")
    return outb, outf, sm, si


def bn_act_bwd(x, dout, out_bf16, gamma, save_mean, save_invstd, want_dres=True, backend=None):
    """-> (dy bf16 [R,C], dres f32 | None, dgamma, dbeta)"""
    be = _be(backend)
    R, Cc = x.shape
    ws = _bn_ws(be, R, Cc, x.device)
    dy = torch.empty((R, Cc), dtype=torch.bfloat16, device=x.device)
    dres = torch.empty((R, Cc), dtype=torch.float32, device=x.device) if want_dres else None
    dg = torch.empty(Cc, dtype=torch.float32, device=x.device); db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    be.check(be.lib.vdk_bn_act_bwd(be.ptr(x), be.ptr(dout), be.ptr(out_bf16) if out_bf16 is not None else None, R, Cc, be.ptr(gamma), be.ptr(save_mean),
                                   be.ptr(save_invstd), be.ptr(dy), be.ptr(dres) if dres is not None else None, be.ptr(dg), be.ptr(db), be.ptr(ws), ws.numel(),
                                   None, None, be.stream()), "vdk_bn_act_bwd")
    return dy, dres, dg, db


def maxpool3s2(x_nhwc: torch.Tensor, want_argmax: bool = False, backend=None):
    """nn.MaxPool2d(3, 2, 1) on bf16 NHWC; want_argmax: also the uint8 map of winning window positions (first maximum) for the backward"""
    be = _be(backend)
    B, H, W, Cc = x_nhwc.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=torch.bfloat16, device=x_nhwc.device)
    arg = torch.empty(out.shape, dtype=torch.uint8, device=x_nhwc.device) if want_argmax else None
    be.check(be.lib.vdk_maxpool3s2_fwd(be.ptr(x_nhwc), be.ptr(out), be.ptr(arg), B, H, W, Cc, be.stream()), "vdk_maxpool3s2_fwd")
    return (out, arg) if want_argmax else out


def maxpool3s2_bwd(x_nhwc: torch.Tensor, dout: torch.Tensor, argmax: Optional[torch.Tensor] = None, backend=None) -> torch.Tensor:
    be = _be(backend)
    B, H, W, Cc = x_nhwc.shape
    din = torch.empty((B, H, W, Cc), dtype=torch.float32, device=x_nhwc.device)
    dout = dout.contiguous()      # keep the (possibly new) tensor alive across the launch: a temporary would be freed before the kernel is enqueued
    be.check(be.lib.vdk_maxpool3s2_bwd(be.ptr(x_nhwc), be.ptr(argmax), be.ptr(dout), be.ptr(din), B, H, W, Cc, be.stream()), "vdk_maxpool3s2_bwd")
    return din
