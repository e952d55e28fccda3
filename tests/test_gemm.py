"""K2 GEMM + transpose on the CPU SIMT emulation vs a plain torch fp32 reference fed the same bf16-rounded
operands (so the only difference is fp32 summation order: rel <= 1e-5)."""
import pytest
import torch

from visiondk_amd import ops


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (34, 192, 64), (257, 264, 200), (1, 8, 8)])
def test_gemm_plain(be, dev, M, N, K):
    torch.manual_seed(0)
    a = torch.randn(M, K).bfloat16().to(dev); b = torch.randn(N, K).bfloat16().to(dev)
    # asymmetric operands catch transposed fragment layouts
    b[:, 0] += 3.0
    ref = a.float() @ b.float().T
    out = ops.gemm_nt(a, b, out_dtype=torch.float32, backend=be)
    assert _rel(out, ref) < 1e-5
    outb = ops.gemm_nt(a, b, out_dtype=torch.bfloat16, backend=be)
    assert _rel(outb.float(), ref.bfloat16().float()) < 3e-3


def test_gemm_epilogues(be, dev):
    torch.manual_seed(1)
    M, N, K = 150, 144, 128
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.1).bfloat16().to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    pre = a.float() @ b.float().T * 0.5 + bias
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, act=ops.ACT_GELU, aux=aux, alpha=0.5,
                      backend=be)
    ref = torch.nn.functional.gelu(pre) + res
    assert _rel(out, ref) < 1e-5
    assert _rel(aux.float(), pre.bfloat16().float()) < 1e-4   # bf16 rounding ties may flip with fp32 summation order
    # dGELU: out = (a @ b^T) * gelu'(u)
    u = torch.randn(M, N).bfloat16().to(dev)
    out2 = ops.gemm_nt(a, b, out_dtype=torch.float32, act=ops.ACT_DGELU, aux=u, backend=be)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    ref2 = (a.float() @ b.float().T) * uu.grad
    assert _rel(out2, ref2) < 1e-5


def test_gemm_strided_and_splitk(be, dev):
    torch.manual_seed(2)
    M, N, K = 96, 136, 1000
    big_a = torch.randn(M, K + 24).bfloat16().to(dev)
    a = big_a[:, 8:8 + K]             # lda > K, 16-B aligned start
    b = torch.randn(N, K).bfloat16().to(dev)
    ref = a.float() @ b.float().T
    out = ops.gemm_nt(a, b, out_dtype=torch.float32, splitk=4, backend=be)
    assert _rel(out, ref) < 1e-5
    out1 = ops.gemm_nt(a, b, out_dtype=torch.float32, splitk=1, backend=be)
    assert _rel(out1, ref) < 1e-5


@pytest.mark.parametrize("R,C", [(64, 64), (70, 130), (197, 8), (3, 6)])
def test_transpose_pad(be, dev, R, C):
    x = torch.randn(R, C).bfloat16().to(dev)
    y = ops.transpose_pad(x, backend=be)
    rp = (R + 63) // 64 * 64
    assert y.shape == (C, rp)
    assert torch.equal(y[:, :R], x.T)
    assert torch.count_nonzero(y[:, R:]) == 0


def test_wgrad_via_transposes(be, dev):
    torch.manual_seed(3)
    T, NO, NI = 150, 72, 136
    dy = torch.randn(T, NO).bfloat16().to(dev); x = torch.randn(T, NI).bfloat16().to(dev)
    dw = ops.gemm_nt(ops.transpose_pad(dy, backend=be), ops.transpose_pad(x, backend=be), out_dtype=torch.float32,
                     splitk=2, backend=be)
    assert _rel(dw, dy.float().T @ x.float()) < 1e-5


def test_transpose_fused_colsum(be, dev):
    torch.manual_seed(4)
    R, C = 300, 72
    x = torch.randn(R, C).bfloat16().to(dev)
    rp = 320
    part = torch.zeros((rp // 64, C), dtype=torch.float32, device=dev)
    y = ops.transpose_pad(x, rpad=rp, colsum_partial=part, backend=be)
    assert torch.equal(y[:, :R], x.T)
    assert _rel(part.sum(0), x.float().sum(0)) < 1e-6


@pytest.mark.parametrize("kern", [2, 5, 6])
@pytest.mark.parametrize("M,N,K,splitk", [(256, 256, 64, 1), (300, 264, 192, 1), (512, 256, 256, 1), (130, 520, 128, 1), (256, 256, 512, 2), (520, 128, 192, 1), (264, 256, 384, 1),
                                          (776, 520, 256, 1), (264, 256, 768, 3)])
def test_gemm256_lds_dma_kernel(be, dev, M, N, K, splitk, kern):
    """the 256x256 LDS-DMA kernels (2: eight waves, 5: four waves / one per SIMD, 6: four waves on 256x128 tiles, two workgroups per CU), forced, vs torch fp32 on the same bf16 operands (incl. ragged M/N, 1..5 k-tiles and split-K)"""
    torch.manual_seed(5)
    a = torch.randn(M, K).bfloat16().to(dev); b = torch.randn(N, K).bfloat16().to(dev)
    b[:, 3] += 2.0
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        if splitk == 1:
            out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
            assert be.lib.vdk_gemm_last_kernel() == (2 if kern == 5 and K % 128 else kern)   # the persistent four-wave kernel multiplies k-tiles in pairs
            assert _rel(out, ref + bias + res) < 1e-5
            outb = ops.gemm_nt(a, b, out_dtype=torch.bfloat16, act=ops.ACT_GELU, backend=be)
            assert _rel(outb.float(), torch.nn.functional.gelu(ref).bfloat16().float()) < 4e-3
        else:
            out = ops.gemm_nt(a, b, out_dtype=torch.float32, splitk=splitk, backend=be)
            assert _rel(out, ref) < 1e-5
    finally:
        be.lib.vdk_gemm_force_kernel(0)


@pytest.mark.parametrize("kern,cw", [(5, 1), (5, 2), (5, 3), (6, 2), (6, 3), (6, 5)])
def test_gemm_column_band_tile_order(be, dev, kern, cw):
    """the one-wave-per-SIMD kernels walk their tiles in column bands (an XCD stays inside a slice of B that fits its L2): every (row, column) tile is still served exactly
    once, in bands of equal and of ragged width -- bit-equal to the plain order"""
    torch.manual_seed(11)
    M, N, K = 1300, 776, 128                     # 6 x 4 tiles of 256 x 256 (persistent walk: more tiles than the emulated CUs / forced) or 6 x 7 of 256 x 128
    a = torch.randn(M, K).bfloat16().to(dev); b = torch.randn(N, K).bfloat16().to(dev); bias = torch.randn(N).to(dev)
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        ref = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == kern
        be.lib.vdk_gemm_force_band_cw(cw)
        out = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, backend=be)
    finally:
        be.lib.vdk_gemm_force_kernel(0); be.lib.vdk_gemm_force_band_cw(-1)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("kern", [2, 5, 6])
@pytest.mark.parametrize("K,M,N,splitk,rg", [(64, 256, 256, 1, 0), (192, 264, 136, 1, 0), (256, 72, 520, 2, 0), (128, 128, 192, 1, 16), (384, 520, 264, 1, 0), (512, 264, 136, 2, 0),
                                             (1536, 520, 264, 3, 0)])      # 6 tiles x 3 splits = 18 items on the 1-D split-K grid: shares of 3, 3, 2, 2, ... per XCD
def test_gemm_tn_from_kmajor_operands(be, dev, K, M, N, splitk, rg, kern):
    """wgrad form C = A^T B with A [K, M], B [K, N] read as they lie (ds_read_b64_tr_b16 fragments), incl. the token-row remap"""
    torch.manual_seed(6)
    phys = K + K // rg + 1 if rg else K
    a_full = torch.randn(phys, M).bfloat16().to(dev); b = torch.randn(K, N).bfloat16().to(dev)
    b[:, 2] += 1.5
    if rg:
        rows = torch.tensor([t + t // rg + 1 for t in range(K)], device=dev)
        a_log = a_full[rows]
    else:
        a_log = a_full
    ref = a_log.float().T @ b.float()
    be.lib.vdk_gemm_force_kernel(kern)   # (the token-row remap is served by the eight-wave kernel either way)
    try:
        out = ops.gemm_nt(a_full, b, out_dtype=torch.float32, splitk=splitk, trans=True, a_row_group=rg, a_rows=K, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == (2 if rg or (kern == 5 and K % 128) else kern)
    finally:
        be.lib.vdk_gemm_force_kernel(0)
    assert out.shape == (M, N)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(70, 40, 64), (130, 200, 100), (5, 64, 200)])
def test_gemm_f32_precise(be, dev, M, N, K):
    """fp32 MFMA GEMM (precise inference path) vs float64 numpy: every element within fp32 round-off of the exact product"""
    torch.manual_seed(M)
    a, b = torch.randn(M, K), torch.randn(N, K)
    bias, res = torch.randn(N), torch.randn(M, N)
    ref = (a.double() @ b.double().t())
    out = ops.gemm_f32(a.to(dev), b.to(dev), backend=be)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-5)
    out = ops.gemm_f32(a.to(dev), b.to(dev), bias=bias.to(dev), residual=res.to(dev), act=ops.ACT_GELU, alpha=0.5, backend=be)
    ref2 = torch.nn.functional.gelu(0.5 * ref + bias.double()) + res.double()
    torch.testing.assert_close(out.cpu().double(), ref2, rtol=1e-5, atol=2e-5)
    if N % 4 == 0:
        out = ops.gemm_f32(a.to(dev), b.t().contiguous().to(dev), b_kmajor=True, backend=be)
        torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-5)


def test_softmax_rows_f32(be, dev):
    x = torch.randn(37, 200) * 3
    y = ops.softmax_rows_f32(x.clone().to(dev), cols=197, scale=0.125, backend=be).cpu()
    torch.testing.assert_close(y[:, :197], torch.softmax(x[:, :197] * 0.125, dim=1), rtol=1e-5, atol=1e-7)
    assert (y[:, 197:] == 0).all()


def test_gemm256_a_colsum_byproduct(be, dev):
    """bias gradient fused into the dgrad GEMM: the 256x256 NT kernel's tn == 0 workgroups also emit column sums of the A operand"""
    torch.manual_seed(5)
    M, N, K = 512, 512, 192
    a = torch.randn(M, K).bfloat16(); b = torch.randn(N, K).bfloat16()
    be.lib.vdk_gemm_force_kernel(2)
    try:
        rows = be.lib.vdk_gemm_a_colsum_rows(M, N, K)
        assert rows == M // 256
        part = torch.full((rows, K), float("nan"), dtype=torch.float32, device=dev)
        out = ops.gemm_nt(a.to(dev), b.to(dev), out_dtype=torch.float32, a_colsum=part, backend=be)
    finally:
        be.lib.vdk_gemm_force_kernel(0)
    torch.testing.assert_close(out.cpu(), a.float() @ b.float().t(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(part.sum(0).cpu(), a.float().sum(0), rtol=1e-5, atol=1e-4)
    assert be.lib.vdk_gemm_a_colsum_rows(300, 512, 192) == 0        # ragged M: by-product not available


@pytest.mark.parametrize("M,N,K,grid", [(768, 512, 256, 8), (768, 512, 256, 16), (600, 776, 384, 8), (512, 512, 1024, 24)])
def test_gemm256_stream_k(be, dev, M, N, K, grid):
    """stream-K launch of the 256x256 kernel (persistent workgroups over (tile, k-tile) units, partial tiles combined by the last arriver in K order), forced, with a
    small grid so that every workgroup owns partial tiles: all fused epilogues + the a_colsum by-product; equal to the whole-tile launch to fp32 summation order,
    bit-identical from launch to launch, tile counters left at zero"""
    torch.manual_seed(6)
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.1).bfloat16().to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    ref = a.float() @ b.float().T
    ws = ops.streamk_workspace(dev, backend=be)
    be.lib.vdk_gemm_streamk_grid(grid); be.lib.vdk_gemm_force_kernel(3)
    try:
        ws = ops.streamk_workspace(dev, backend=be)
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, streamk_ws=ws, backend=be)
        assert _rel(out, ref + bias + res) < 1e-5
        out2 = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, streamk_ws=ws, backend=be)
        assert torch.equal(out, out2)
        u = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        g = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=u, streamk_ws=ws, backend=be)
        assert _rel(u.float(), ref + bias) < 4e-3 and _rel(g.float(), torch.nn.functional.gelu(ref + bias)) < 4e-3
        plain = ops.gemm_nt(a, b, streamk_ws=ws, backend=be)
        be.lib.vdk_gemm_force_kernel(2)
        assert _rel(plain.float(), ops.gemm_nt(a, b, backend=be).float()) < 1e-3
        be.lib.vdk_gemm_force_kernel(3)
        if M % 256 == 0:
            rows = be.lib.vdk_gemm_a_colsum_rows(M, N, K)
            part = torch.full((rows, K), float("nan"), dtype=torch.float32, device=dev)
            o3 = ops.gemm_nt(a, b, out_dtype=torch.float32, a_colsum=part, streamk_ws=ws, backend=be)
            assert _rel(o3, ref) < 1e-5
            torch.testing.assert_close(part.sum(0).cpu(), a.float().sum(0).cpu(), rtol=1e-5, atol=1e-4)
        assert int(ws[:65536].view(torch.int32).abs().sum()) == 0
    finally:
        be.lib.vdk_gemm_force_kernel(0); be.lib.vdk_gemm_streamk_grid(0)


@pytest.mark.parametrize("kern", [2, 5, 6])
@pytest.mark.parametrize("M", [512, 300])
def test_gemm256_c_colsum_byproduct(be, dev, M, kern):
    """bias gradient fused into the PRODUCER of dY: the 256x256 NT kernel's plain / dGELU bf16 epilogues also emit column sums of what they store"""
    torch.manual_seed(7)
    N, K = 512, 256
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.2).bfloat16().to(dev); u = torch.randn(M, N).bfloat16().to(dev)
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
        assert rows == 2 * ((M + 255) // 256)
        for act in (ops.ACT_NONE, ops.ACT_DGELU):
            part = torch.full((rows, N), float("nan"), dtype=torch.float32, device=dev)
            out = ops.gemm_nt(a, b, act=act, aux=u if act else None, c_colsum=part, backend=be)
            ref = ops.gemm_nt(a, b, act=act, aux=u if act else None, backend=be)
            assert torch.equal(out, ref)
            torch.testing.assert_close(part.sum(0).cpu(), out.float().sum(0).cpu(), rtol=1e-5, atol=1e-4)
    finally:
        be.lib.vdk_gemm_force_kernel(0)


@pytest.mark.parametrize("kern", [5, 6])
@pytest.mark.parametrize("M,N,K", [(1300, 776, 256), (2050, 520, 128)])
def test_gemm_w4_persistent_walk(be, dev, M, N, K, kern):
    """more output tiles than CUs (the emulated device has 16): the four-wave kernel walks its tiles with the DMA cursor running into the next tile;
    every fused epilogue form against torch fp32 on the same bf16 operands, ragged M / N"""
    torch.manual_seed(11)
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.3).bfloat16().to(dev)
    b[:, 5] += 1.0
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev); u = torch.randn(M, N).bfloat16().to(dev)
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        out = ops.gemm_nt(a, b, bias=bias, backend=be)                                                    # bias -> bf16 (row-staged form)
        assert be.lib.vdk_gemm_last_kernel() == kern
        assert _rel(out.float(), (ref + bias).bfloat16().float()) < 3e-3
        out = ops.gemm_nt(a, b, backend=be)                                                               # plain bf16
        assert _rel(out.float(), ref.bfloat16().float()) < 3e-3
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=aux, backend=be)                          # bias + GELU + saved pre-activation
        assert _rel(aux.float(), (ref + bias).bfloat16().float()) < 3e-3
        assert _rel(out.float(), torch.nn.functional.gelu(ref + bias).bfloat16().float()) < 4e-3
        rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
        part = torch.full((rows, N), float("nan"), dtype=torch.float32, device=dev)
        out = ops.gemm_nt(a, b, act=ops.ACT_DGELU, aux=u, c_colsum=part, backend=be)                       # dGELU + column sums of the stored output
        uu = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uu).sum().backward()
        assert _rel(out.float(), (ref * uu.grad).bfloat16().float()) < 4e-3
        torch.testing.assert_close(part.sum(0).cpu(), out.float().sum(0).cpu(), rtol=1e-4, atol=1e-3)
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)              # fp32 + residual (slab form)
        assert _rel(out, ref + bias + res) < 1e-5
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, backend=be)                            # fp32 + bias without residual (ConvNeXt stem / downsample): row-staged form
        assert _rel(out, ref + bias) < 1e-5
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, alpha=0.5, backend=be)                            # run-time-flag form
        assert _rel(out, 0.5 * ref) < 1e-5
        assert be.lib.vdk_gemm_last_kernel() == kern
    finally:
        be.lib.vdk_gemm_force_kernel(0)


def test_gemm_w4_row_split_for_the_ragged_round(be, dev):
    """18 tiles on the emulated 16 CUs, K = 1536: the persistent four-wave kernel takes the 8 tile rows of the whole round, the 256x128 kernel the last row
    (two launches over row ranges of the same operands); outputs, saved pre-activation and column-sum partials against torch on the same bf16 operands"""
    torch.manual_seed(13)
    M, N, K = 2200, 300 // 4 * 4 + 4, 1536       # N = 304 -> 2 column tiles
    a = (torch.randn(M, K) * 0.5).bfloat16().to(dev); b = (torch.randn(N, K) * 0.1).bfloat16().to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev); u = torch.randn(M, N).bfloat16().to(dev)
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(5)
    try:
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == 5
        assert _rel(out, ref + bias + res) < 1e-5
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=aux, backend=be)
        assert _rel(aux.float(), (ref + bias).bfloat16().float()) < 3e-3
        assert _rel(out.float(), torch.nn.functional.gelu(ref + bias).bfloat16().float()) < 4e-3
        rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
        part = torch.full((rows, N), float("nan"), dtype=torch.float32, device=dev)
        out = ops.gemm_nt(a, b, act=ops.ACT_DGELU, aux=u, c_colsum=part, backend=be)
        uu = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uu).sum().backward()
        assert _rel(out.float(), (ref * uu.grad).bfloat16().float()) < 4e-3
        torch.testing.assert_close(part.sum(0).cpu(), out.float().sum(0).cpu(), rtol=1e-4, atol=2e-3)
    finally:
        be.lib.vdk_gemm_force_kernel(0)


@pytest.mark.parametrize("kern", [1, 2, 5, 6])
def test_gemm_gelu_with_saved_derivative(be, dev, kern):
    """VDK_ACT_GELU_SAVE_GRAD / VDK_ACT_MUL_AUX on every kernel structure: the forward epilogue stores fp16 GELU'(pre-activation) beside bf16 GELU(pre-activation)
    (bit-equal to what VDK_ACT_GELU stores), the backward epilogue is one multiplication -- with and without the column sums of what it stores; against torch on the same
    bf16 operands and against the two-evaluation pair (VDK_ACT_GELU + VDK_ACT_DGELU), whose derivative is evaluated from the bf16-rounded pre-activation"""
    torch.manual_seed(17)
    M, N, K = 1300, 520, 256
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.1).bfloat16().to(dev)
    bias = torch.randn(N).to(dev)
    ref = a.float() @ b.float().T + bias
    rr = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(rr).sum().backward()
    dref = rr.grad
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        u = torch.empty(M, N, dtype=torch.bfloat16, device=dev); dsv = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        g_old = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=u, backend=be)
        g_new = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU_SAVE_GRAD, aux=dsv, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == kern
        assert torch.equal(g_old.view(torch.int16), g_new.view(torch.int16))
        assert (dsv.float() - dref).abs().max().item() < 1.5e-3                             # fp16 spacing at 1 is 9.8e-4 (|GELU'| <= 1.13), fp32 epilogue arithmetic below it
        assert _rel(dsv.float(), dref) < 4e-4
        # backward: dY [M, N2] x W^T -> [M, N] scaled by the saved derivative
        dy = torch.randn(M, K).bfloat16().to(dev)
        acc = dy.float() @ b.float().T
        out = ops.gemm_nt(dy, b, act=ops.ACT_MUL_AUX, aux=dsv, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == kern
        assert _rel(out.float(), (acc * dsv.float()).bfloat16().float()) < 3e-3
        out_old = ops.gemm_nt(dy, b, act=ops.ACT_DGELU, aux=u, backend=be)
        assert _rel(out.float(), out_old.float()) < 6e-3                                     # the two pairs differ by the bf16 rounding of the derivative
        rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
        if rows > 0:
            part = torch.full((rows, N), float("nan"), dtype=torch.float32, device=dev)
            out2 = ops.gemm_nt(dy, b, act=ops.ACT_MUL_AUX, aux=dsv, c_colsum=part, backend=be)
            assert torch.equal(out2.view(torch.int16), out.view(torch.int16))
            torch.testing.assert_close(part.sum(0).cpu(), out2.float().sum(0).cpu(), rtol=1e-4, atol=2e-3)
    finally:
        be.lib.vdk_gemm_force_kernel(0)


def test_gemm_narrow_output_goes_to_the_256x128_kernel(be, dev, monkeypatch):
    """128 <= N < 256 with many rows (ConvNeXt's first stage: C = 128) is one column of 256x128 tiles on the two-workgroups-per-CU kernel; below the row threshold the 128x128
    kernel keeps it.  (The threshold is read once per process: this test only runs the narrow case when the environment of the process already lowered it.)"""
    import os
    torch.manual_seed(19)
    M, N, K = 1100, 136, 192
    a = torch.randn(M, K).bfloat16().to(dev); b = (torch.randn(N, K) * 0.2).bfloat16().to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    ref = a.float() @ b.float().T
    out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
    assert _rel(out, ref + bias + res) < 1e-5
    want = 6 if int(os.environ.get("VDK_GEMM_NARROW_MIN_M", "65536")) <= M else 1
    assert be.lib.vdk_gemm_last_kernel() == want
    out = ops.gemm_nt(a, b, backend=be)
    assert _rel(out.float(), ref.bfloat16().float()) < 3e-3
    assert be.lib.vdk_gemm_last_kernel() == want


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kern", [1, 2, 5, 6])
def test_gemm_col_scale_before_bias_and_residual(be, dev, kern, dtype):
    """VdkGemmDesc.col_scale: C = residual + (acc * col_scale + bias) -- ConvNeXt's layer scale in the fc2 epilogue under fp16 operands (gamma = 1e-6 cannot be folded into
    an fp16 weight).  Every kernel structure that can serve the problem, ragged M / N, with and without the residual; a NULL col_scale leaves the epilogue bit-identical."""
    if kern == 2 and dtype == torch.float16:
        pytest.skip("the eight-wave kernel is bf16 only")
    M, N, K = 1300, 776, 256
    torch.manual_seed(12)
    a = torch.randn(M, K).to(dtype).to(dev); b = (torch.randn(N, K) * 0.3).to(dtype).to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    cs = (torch.rand(N) * 2e-6).to(dev); cs[::7] = 0.5
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, col_scale=cs, backend=be)
        assert be.lib.vdk_gemm_last_kernel() == kern
        assert _rel(out - res, ref * cs + bias) < 2e-5
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, col_scale=cs, backend=be)
        assert _rel(out, ref * cs + bias) < 1e-5
        one = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, col_scale=torch.ones(N, device=dev), backend=be)
        none = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
        assert torch.equal(one, none)
    finally:
        be.lib.vdk_gemm_force_kernel(0)
    with pytest.raises(Exception):      # 16-bit outputs / activations / split-K do not take it
        ops.gemm_nt(a, b, bias=bias, col_scale=cs, backend=be)


def test_gemm_fp16_row_blocks_for_tensors_beyond_32_bit_offsets(be, dev, monkeypatch):
    """fp16 problems whose output / residual / aux rows would exceed the four-wave kernels' 32-bit buffer offsets (the margin head at C = 10^6: [512, 10^6] fp32) run as row
    blocks; the limit is lowered here so that small problems take the path.  NT with bias + residual, NT with GELU + aux, TN (weight-gradient form): bit-equal to one launch."""
    torch.manual_seed(13)
    M, N, K = 1100, 520, 128
    a = torch.randn(M, K).half().to(dev); b = (torch.randn(N, K) * 0.3).half().to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    at = torch.randn(K * 2, 776).half().to(dev); bt = torch.randn(K * 2, N).half().to(dev)
    def run():
        o1 = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
        aux = torch.empty(M, N, dtype=torch.float16, device=dev)
        o2 = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=aux, backend=be)
        o3 = ops.gemm_nt(at, bt, out_dtype=torch.float32, trans=True, backend=be)
        return o1, o2, aux, o3
    whole = run()
    monkeypatch.setenv("VDK_GEMM_ROWBLOCK_LIMIT", str((256 + 256) * N * 4 + 1024))      # fp32 rows of N: blocks of 256 rows
    parts = run()
    for w, q in zip(whole, parts):
        assert torch.equal(w, q)
    ref = at.float().T @ bt.float()
    assert _rel(parts[3], ref) < 1e-5
