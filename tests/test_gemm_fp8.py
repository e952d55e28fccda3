"""OCP fp8 GEMM (csrc/gemm_fp8.hip: v_mfma_scale_f32_32x32x64_f8f6f4, scales 2^0) and its quantisation kernels -- the operand path of BASELINE.json configs[4].
Oracle: torch's own float8 dtypes (OCP e4m3fn / e5m2, round-to-nearest-even) for the conversion, fp32 matmul of the dequantised bytes for the product: the fp8 values
are exact in fp32, so the GEMM itself is held to summation-order tolerance (2e-5 rel for fp32 outputs, 3e-3 for bf16 outputs = their own final rounding); the
quantisation is bit-exact against torch's cast.  What fp8 costs END TO END against the fp32 oracle is stated in test_fp8_linear_error_against_fp32."""
import math

import pytest
import torch

from visiondk_amd import ops
from visiondk_amd.ops import ACT_DGELU, ACT_GELU, FP8_E4M3, FP8_E5M2


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _deq(u8, fmt):
    return u8.cpu().view(torch.float8_e4m3fn if fmt == FP8_E4M3 else torch.float8_e5m2).float()


@pytest.mark.parametrize("fmt", [FP8_E4M3, FP8_E5M2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantisation_is_torchs_fp8_cast(be, dev, fmt, dtype):
    torch.manual_seed(0)
    x = (torch.randn(4, 1024) * torch.logspace(-4, 3, 1024)).to(dtype)        # subnormals of both formats up to values beyond their range
    x[0, :4] = torch.tensor([0.0, -0.0, 448.0, -57344.0]).to(dtype)
    scale = torch.tensor([0.75], dtype=torch.float32)
    amax = torch.zeros(1, dtype=torch.float32)
    q = ops.quant_fp8(x.to(dev), scale.to(dev), fmt, amax.to(dev) if dev == "cpu" else amax.to(dev), backend=be)
    lim = 448.0 if fmt == FP8_E4M3 else 57344.0
    ref = (x.float() * 0.75).clamp(-lim, lim).to(torch.float8_e4m3fn if fmt == FP8_E4M3 else torch.float8_e5m2)
    assert torch.equal(q.cpu(), ref.view(torch.uint8))


def test_amax_and_delayed_scale_update(be, dev):
    x = torch.randn(64, 256) * 3.0
    amax = torch.zeros(2, dtype=torch.float32, device=dev); scale = torch.ones(2, device=dev); inv = torch.ones(2, device=dev)
    ops.quant_fp8(x.to(dev), None, FP8_E4M3, amax[0:1], backend=be)
    ops.quant_fp8((x * 0.5).to(dev), None, FP8_E4M3, amax[0:1], backend=be)          # the maximum accumulates over calls
    assert abs(amax[0].item() - x.abs().max().item()) < 1e-6 and amax[1].item() == 0.0
    ops.fp8_scale_update(amax, scale, inv, FP8_E4M3, margin=1.0, backend=be)
    assert abs(scale[0].item() - 448.0 / x.abs().max().item()) < 1e-3 and abs(inv[0].item() * scale[0].item() - 1) < 1e-6
    assert scale[1].item() == 1.0 and amax.abs().max().item() == 0.0                 # an unused slot keeps its scale; the window restarts


@pytest.mark.parametrize("M,N,K,afmt", [(256, 256, 128, FP8_E4M3), (512, 256, 384, FP8_E5M2), (300, 264, 256, FP8_E4M3)])
def test_fp8_gemm_is_exact_on_the_quantised_operands(be, dev, M, N, K, afmt):
    torch.manual_seed(1)
    a = ops.quant_fp8(torch.randn(M + (-M) % 16, K).to(dev), None, afmt, None, backend=be)[:M]
    b = ops.quant_fp8((torch.randn(N + (-N) % 16, K) * 0.2).to(dev), None, FP8_E4M3, None, backend=be)[:N]
    sa = torch.tensor([0.37], device=dev); sb = torch.tensor([1.9], device=dev)
    ref = (_deq(a, afmt) @ _deq(b, FP8_E4M3).t()) * (0.37 * 1.9)
    got = ops.gemm_fp8_nt(a.contiguous(), b.contiguous(), sa, sb, a_fmt=afmt, out_dtype=torch.float32, backend=be)
    assert _rel(got, ref) < 2e-5
    bias = torch.randn(N, device=dev)
    got = ops.gemm_fp8_nt(a.contiguous(), b.contiguous(), sa, sb, a_fmt=afmt, bias=bias, backend=be)
    assert _rel(got.float(), ref + bias.cpu()) < 3e-3


def test_fp8_gemm_fused_epilogues(be, dev):
    torch.manual_seed(2)
    M, N, K = 256, 512, 256
    a = ops.quant_fp8(torch.randn(M, K).to(dev), None, FP8_E4M3, None, backend=be)
    b = ops.quant_fp8((torch.randn(N, K) * 0.1).to(dev), None, FP8_E4M3, None, backend=be)
    ref = _deq(a, 0) @ _deq(b, 0).t()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev)
    u = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    g = ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be)
    uref = ref + bias.cpu()
    assert _rel(u.float(), uref) < 3e-3 and _rel(g.float(), torch.nn.functional.gelu(uref)) < 3e-3
    r = ops.gemm_fp8_nt(a, b, bias=bias, residual=res, out_dtype=torch.float32, backend=be)
    assert _rel(r, uref + res.cpu()) < 2e-5
    dy = ops.quant_fp8(torch.randn(M, K).to(dev), None, FP8_E5M2, None, backend=be)
    du = ops.gemm_fp8_nt(dy, b, a_fmt=FP8_E5M2, act=ACT_DGELU, aux=u, backend=be)
    uf = u.float().cpu()
    gp = 0.5 * (1 + torch.erf(uf / math.sqrt(2))) + uf * torch.exp(-0.5 * uf * uf) / math.sqrt(2 * math.pi)
    assert _rel(du.float(), (_deq(dy, 1) @ _deq(b, 0).t()) * gp) < 3e-3


@pytest.mark.parametrize("M", [256, 300])
def test_fp8_copy_of_the_output_rides_with_the_epilogue(be, dev, M):
    """vdk_gemm_fp8_nt_q8: the GELU / dGELU epilogues also write the fp8 quantisation of the bf16 tensor they store (the A operand of the next fp8 GEMM) and accumulate its
    amax -- bit-identical to the separate vdk_quant_fp8 pass over that tensor, which is what the engine's fp8 mode runs without the fusion."""
    torch.manual_seed(5)
    N, K = 512, 256
    a = ops.quant_fp8(torch.randn(M, K).to(dev), None, FP8_E4M3, None, backend=be) if M % 16 == 0 else ops.quant_fp8(torch.randn(304, K).to(dev), None, FP8_E4M3, None, backend=be)[:M]
    b = ops.quant_fp8((torch.randn(N, K) * 0.1).to(dev), None, FP8_E4M3, None, backend=be)
    bias = torch.randn(N, device=dev) * 0.1
    u = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    sc = torch.tensor([37.0], device=dev); am = torch.zeros(1, device=dev)
    g, g8 = ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be, q8={"fmt": FP8_E4M3, "scale": sc, "amax": am})
    g_plain = ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=torch.zeros_like(u), backend=be)
    assert torch.equal(g, g_plain)
    am_ref = torch.zeros(1, device=dev)
    gpad = torch.zeros((M + 15) // 16 * 16, N, dtype=torch.bfloat16, device=dev); gpad[:M] = g
    ref8 = ops.quant_fp8(gpad, sc, FP8_E4M3, am_ref, backend=be)[:M]
    assert torch.equal(g8, ref8) and torch.equal(am, am_ref) and am.item() > 0
    dy = ops.quant_fp8(torch.randn((M + 15) // 16 * 16, K).to(dev), None, FP8_E5M2, None, backend=be)[:M]
    sc2 = torch.tensor([900.0], device=dev); am2 = torch.zeros(1, device=dev)
    du, du8 = ops.gemm_fp8_nt(dy, b, a_fmt=FP8_E5M2, act=ACT_DGELU, aux=u, backend=be, q8={"fmt": FP8_E5M2, "scale": sc2, "amax": am2})
    am2_ref = torch.zeros(1, device=dev)
    dpad = torch.zeros((M + 15) // 16 * 16, N, dtype=torch.bfloat16, device=dev); dpad[:M] = du
    ref8 = ops.quant_fp8(dpad, sc2, FP8_E5M2, am2_ref, backend=be)[:M]
    assert torch.equal(du8, ref8) and torch.equal(am2, am2_ref)
    # column sums of the stored du (the bias gradient of fc1) from the same epilogue, with and without the fp8 copy
    for q8 in (None, {"fmt": FP8_E5M2, "scale": sc2, "amax": torch.zeros(1, device=dev)}):
        part = torch.full((2 * ((M + 255) // 256), N), float("nan"), device=dev)
        r = ops.gemm_fp8_nt(dy, b, a_fmt=FP8_E5M2, act=ACT_DGELU, aux=u, backend=be, q8=q8, c_colsum=part)
        du2 = r[0] if q8 is not None else r
        assert torch.equal(du2, du)
        assert _rel(part.sum(0), du.float().sum(0)) < 1e-5
    with pytest.raises(RuntimeError):        # the plain bias form has no fp8 by-product
        ops.gemm_fp8_nt(a, b, bias=bias, backend=be, q8={"fmt": FP8_E4M3, "scale": sc, "amax": am})


def test_fp8_linear_error_against_fp32(be, dev):
    """the stated fp8 tolerance: a Linear with per-tensor scaled e4m3 operands against the fp32 product -- 2^-4 relative rounding per
    operand element averages to a few percent of the output norm: <= 5e-2 here (measured 3.6e-2), against 4e-3 for bf16 operands"""
    torch.manual_seed(3)
    M, N, K = 256, 256, 1024
    x = torch.randn(M, K); w = torch.randn(N, K) * 0.02
    sx = torch.tensor([448.0 / x.abs().max().item()]); sw = torch.tensor([448.0 / w.abs().max().item()])
    a = ops.quant_fp8(x.to(dev), sx.to(dev), FP8_E4M3, None, backend=be); b = ops.quant_fp8(w.to(dev), sw.to(dev), FP8_E4M3, None, backend=be)
    got = ops.gemm_fp8_nt(a, b, (1 / sx).to(dev), (1 / sw).to(dev), out_dtype=torch.float32, backend=be)
    err = _rel(got, x @ w.t())
    print("fp8 e4m3 linear rel err", err)
    assert err < 5e-2


@pytest.mark.parametrize("T,C", [(37, 256), (1000, 768), (5000, 1024)])
@pytest.mark.parametrize("fmt", [FP8_E4M3, FP8_E5M2])
def test_layernorm_writes_the_fp8_copy_of_its_output(be, dev, T, C, fmt):
    """vdk_layernorm_fwd_q8: y, mean, rstd equal to vdk_layernorm_fwd bit for bit; y8 and amax equal to vdk_quant_fp8 over y.  T = 5000 > 4 * 1024 rows: waves walk
    over several rows of the bounded grid."""
    torch.manual_seed(6)
    x = (torch.randn(T, C) * 2 + 0.3).to(dev); g = (1 + 0.1 * torch.randn(C)).to(dev); b = (0.1 * torch.randn(C)).to(dev)
    sc = torch.tensor([11.0], device=dev); am = torch.zeros(1, device=dev)
    y, y8, mean, rstd = ops.layernorm_fwd_q8(x, g, b, sc, am, fmt, backend=be)
    y0, mean0, rstd0 = ops.layernorm_fwd(x, g, b, backend=be)
    assert torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    pad = (-T * C) % 16
    flat = torch.cat([y0.reshape(-1), torch.zeros(pad, dtype=torch.bfloat16, device=dev)])
    am0 = torch.zeros(1, device=dev)
    ref8 = ops.quant_fp8(flat, sc, fmt, am0, backend=be)[:T * C].reshape(T, C)
    assert torch.equal(y8, ref8) and torch.equal(am, am0) and am.item() > 0
