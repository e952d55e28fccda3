"""csrc/resnet_ops.hip + the implicit-GEMM convolution of csrc/gemm.hip vs plain torch (the ops they replace), emulator and MI355X."""
import pytest
import torch
import torch.nn.functional as F

from visiondk_amd import ops

nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,Ci,Co,H,W,k,s,p", [(2, 8, 16, 9, 7, 3, 1, 1), (2, 16, 24, 8, 8, 3, 2, 1), (3, 8, 16, 6, 6, 1, 2, 0), (1, 3, 8, 16, 16, 7, 2, 3)])
def test_implicit_conv_fwd_dgrad_wgrad(be, dev, B, Ci, Co, H, W, k, s, p):
    torch.manual_seed(0)
    x = torch.randn(B, Ci, H, W).bfloat16().float()
    w = (torch.randn(Co, Ci, k, k) * 0.2).bfloat16().float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride=s, padding=p)
    OH, OW = y.shape[2:]
    dy = torch.randn_like(y).bfloat16().float()
    y.backward(dy)
    cip = (Ci + 7) // 8 * 8
    a = ops.nchw_to_nhwc_bf16(x.to(dev), cip, backend=be)                                 # [B,H,W,Cip] bf16 (zero-padded channels)
    wf, wd = ops.conv_weight_prep(w.to(dev), cip, backend=be)
    yk = ops.conv_gemm(a, wf, oh=OH, ow=OW, kh=k, kw=k, stride=s, pad=p, backend=be)      # f32 [B*OH*OW, Co]
    torch.testing.assert_close(yk.cpu().view(B, OH, OW, Co).permute(0, 3, 1, 2), y.detach(), rtol=1e-4, atol=1e-4)
    # input gradient: rows = input pixels, gathered tensor = dY
    dyb = nhwc(dy).bfloat16().to(dev)
    dx = ops.conv_gemm(dyb, wd, oh=H, ow=W, kh=k, kw=k, stride=s, pad=p, transposed=True, backend=be)   # f32 [B*H*W, Cip]
    torch.testing.assert_close(dx.cpu().view(B, H, W, cip)[..., :Ci].permute(0, 3, 1, 2), xr.grad, rtol=1e-4, atol=1e-4)
    # weight gradient: explicit im2col + dY^T . col
    col = ops.im2col(a, OH, OW, k, k, s, p, backend=be)
    dwp = (dyb.view(-1, Co).float().t() @ col.float())                                   # the GEMM itself is covered by test_gemm; here the layouts
    dw = ops.conv_wgrad_unpermute(dwp.contiguous(), Ci, k, k, backend=be)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("R,C,res_dtype,relu", [(300, 16, None, True), (1000, 72, torch.bfloat16, True), (520, 8, torch.float32, True), (257, 64, None, False)])
def test_bn_act_fwd_bwd(be, dev, R, C, res_dtype, relu):
    torch.manual_seed(R)
    x = torch.randn(R, C) * 2 + 0.3
    res = None if res_dtype is None else torch.randn(R, C).to(res_dtype)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    rm, rv = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev)
    xr = x.clone().requires_grad_(True)
    rr = None if res is None else res.float().clone().requires_grad_(True)
    pre = bn(xr) + (rr if rr is not None else 0)
    out = F.relu(pre) if relu else pre
    dout = torch.randn_like(out)
    out.backward(dout)
    ob, of, sm, si = ops.bn_act_fwd(x.to(dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev), rm, rv, res=None if res is None else res.to(dev), relu=relu,
                                    want_f32=True, backend=be)
    torch.testing.assert_close(of.cpu(), out.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ob.float().cpu(), out.detach().bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    # backward: the ReLU mask comes from the saved bf16 output; use the fp32-exact positions (avoid values that round to 0 in bf16: none here)
    dy, dres, dg, db = ops.bn_act_bwd(x.to(dev), dout.to(dev), ob if relu else None, bn.weight.detach().to(dev), sm, si, backend=be)
    torch.testing.assert_close(dy.float().cpu(), xr.grad.bfloat16().float(), rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(dg.cpu(), bn.weight.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.cpu(), bn.bias.grad, rtol=1e-3, atol=1e-3)
    if rr is not None:
        torch.testing.assert_close(dres.cpu(), rr.grad, rtol=1e-6, atol=1e-6)


def test_maxpool_first_max_rule_and_avgpool(be, dev):
    torch.manual_seed(3)
    B, C, H, W = 2, 8, 9, 12
    x = F.relu(torch.randn(B, C, H, W)).bfloat16().float()          # many exact zeros -> ties inside windows
    x[0, :, 2:5, 2:5] = 1.5                                           # a plateau: every window over it is a tie
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    a = nhwc(x).bfloat16().to(dev)
    yk = ops.maxpool3s2(a, backend=be)
    assert torch.equal(yk.float().cpu().permute(0, 3, 1, 2), y.detach())
    din = ops.maxpool3s2_bwd(a, nhwc(dy).to(dev), backend=be)
    torch.testing.assert_close(din.cpu().permute(0, 3, 1, 2), xr.grad, rtol=1e-6, atol=1e-6)
    # the forward's argmax map (what the engine keeps) routes the gradient to the same winners, bit for bit, without re-reading the windows
    yk2, arg = ops.maxpool3s2(a, want_argmax=True, backend=be)
    assert torch.equal(yk2, yk) and arg.dtype == torch.uint8 and int(arg.max()) <= 8
    din2 = ops.maxpool3s2_bwd(a, nhwc(dy).to(dev), argmax=arg, backend=be)
    assert torch.equal(din2, din)
