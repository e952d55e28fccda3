"""csrc/conv.hip vs plain torch fp32 (the op each kernel replaces), on the CPU SIMT emulation and (-m gpu) on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

from visiondk_amd import ops


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 7, 7, 16), (2, 14, 10, 72), (1, 20, 17, 8), (1, 14, 14, 64), (1, 7, 28, 8), (2, 56, 56, 24), (2, 28, 28, 40), (3, 14, 14, 72), (2, 7, 7, 136), (1, 14, 28, 8)])   # square 56/28/14/7: the row-streaming kernel
def test_dwconv7_fwd_dgrad_wgrad(be, dev, B, H, W, C):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W)
    w = torch.randn(C, 1, 7, 7) * 0.1
    b = torch.randn(C)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, br, padding=3, groups=C)
    dy = torch.randn_like(y)
    y.backward(dy)
    skip = torch.randn_like(x)

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    wt = ops.dwconv7_weight_prep(w.to(dev), backend=be)
    yk = ops.dwconv7(nhwc(x), wt, b.to(dev), backend=be)
    torch.testing.assert_close(yk.cpu().permute(0, 3, 1, 2), y.detach(), rtol=1e-5, atol=1e-5)
    # input gradient: same kernel with flipped taps, shortcut gradient added, bf16 copy
    dxk, dxb = ops.dwconv7(nhwc(dy), wt, None, nhwc(skip), flip=True, want_bf16=True, backend=be)
    ref = xr.grad + skip
    torch.testing.assert_close(dxk.cpu().permute(0, 3, 1, 2), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dxb.float().cpu().permute(0, 3, 1, 2), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    dw, db = ops.dwconv7_wgrad(nhwc(x), nhwc(dy), backend=be)
    torch.testing.assert_close(dw.cpu().reshape(C, 1, 7, 7), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,W,C,per_chunk", [(5, 14, 72, 2), (4, 7, 136, 1), (3, 28, 40, 1), (3, 56, 24, 2)])
def test_dwconv7_persistent_workgroups_walk_over_images(be, dev, B, W, C, per_chunk, monkeypatch):
    """The row-streaming kernel is persistent: a workgroup keeps its channel chunk and walks over images, prefetching the next image's first rows under
    the last strip of the current one.  Real launches give a workgroup several images only for B > 512 / chunks, so the tests force it."""
    monkeypatch.setenv("VDK_DW_ROWS_PER_CHUNK", str(per_chunk))
    torch.manual_seed(1)
    x = torch.randn(B, C, W, W)
    w = torch.randn(C, 1, 7, 7) * 0.1
    b = torch.randn(C)
    skip = torch.randn_like(x)
    y = F.conv2d(x, w, b, padding=3, groups=C)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    wt = ops.dwconv7_weight_prep(w.to(dev), backend=be)
    yk = ops.dwconv7(nhwc(x), wt, b.to(dev), nhwc(skip), backend=be)
    torch.testing.assert_close(yk.cpu().permute(0, 3, 1, 2), y + skip, rtol=1e-5, atol=1e-5)


def test_conv2x2_stride2_as_gemm(be, dev):
    torch.manual_seed(1)
    B, H, W, Ci, Co = 2, 8, 4, 16, 24
    x = torch.randn(B, Ci, H, W).bfloat16().float()
    w = (torch.randn(Co, Ci, 2, 2) * 0.2)
    bias = torch.randn(Co)
    xr = x.clone().requires_grad_(True); wr = w.bfloat16().float().requires_grad_(True)
    y = F.conv2d(xr, wr, bias, stride=2)
    dy = torch.randn_like(y).bfloat16().float()
    y.backward(dy)

    xb = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)
    a = ops.space_to_depth2(xb, backend=be)                                  # [B, H/2, W/2, 4Ci]
    wb, wtb = ops.conv2x2_weight_prep(w.to(dev), backend=be)
    yk = ops.gemm_nt(a.reshape(-1, 4 * Ci), wb, out_dtype=torch.float32, bias=bias.to(dev), backend=be)
    torch.testing.assert_close(yk.cpu().reshape(B, H // 2, W // 2, Co).permute(0, 3, 1, 2), y.detach(), rtol=1e-4, atol=1e-4)
    # input gradient: dA = dY @ W (NT with the transposed copy), then depth-to-space
    dyb = dy.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)
    da = ops.gemm_nt(dyb.reshape(-1, Co), wtb, out_dtype=torch.bfloat16, backend=be)
    dx = ops.space_to_depth2(da.reshape(B, H // 2, W // 2, 4 * Ci), inverse=True, backend=be)
    torch.testing.assert_close(dx.float().cpu().permute(0, 3, 1, 2), xr.grad, rtol=2e-2, atol=2e-2)
    # weight gradient: dWp = dY^T A, back to [Co, Ci, 2, 2]
    dwp = ops.gemm_nt(dyb.reshape(-1, Co).t().contiguous(), a.reshape(-1, 4 * Ci).t().contiguous(), out_dtype=torch.float32, backend=be)
    dw = ops.conv2x2_wgrad_unpermute(dwp, Ci, backend=be)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-3)
    # space-to-depth round trip
    assert torch.equal(ops.space_to_depth2(a, inverse=True, backend=be).cpu(), xb.cpu())


def test_layerscale_fold(be, dev):
    torch.manual_seed(2)
    C, M, R = 16, 64, 40
    w2 = torch.randn(C, M) * 0.1; b2 = torch.randn(C); gamma = torch.rand(C) + 0.5
    g = torch.randn(R, M)
    w2r = w2.clone().requires_grad_(True); b2r = b2.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True)
    y = (g @ w2r.t() + b2r) * gr
    dy = torch.randn_like(y)
    y.backward(dy)
    w2p, w2pt, b2p = ops.layerscale_weight_prep(w2.to(dev), b2.to(dev), gamma.to(dev), backend=be)
    torch.testing.assert_close(w2p.float().cpu(), (gamma[:, None] * w2).bfloat16().float())
    torch.testing.assert_close(w2pt.float().cpu(), (gamma[:, None] * w2).bfloat16().float().t())
    torch.testing.assert_close(b2p.cpu(), gamma * b2)
    # gradients of the folded parameters (fp32 here), then the chain rule kernel
    dw2p = dy.t() @ g; db2p = dy.sum(0)
    dw2, db2, dg = ops.layerscale_grad(dw2p.to(dev), db2p.to(dev), w2.to(dev), b2.to(dev), gamma.to(dev), backend=be)
    torch.testing.assert_close(dw2.cpu(), w2r.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db2.cpu(), b2r.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
