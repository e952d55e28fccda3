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
    # the engine's form: no im2col matrix, the TN kernel gathers its B tiles from the NHWC input (VdkConvGeom.rows) -- whole and split along the pixels
    for sk in (1, 2):
        dwi = ops.conv_wgrad_implicit(dyb.view(-1, Co), a, oh=OH, ow=OW, kh=k, kw=k, stride=s, pad=p, splitk=sk, backend=be)
        torch.testing.assert_close(dwi.cpu(), dwp.cpu(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,Ci,Co,H,k,s,p", [(5, 40, 264, 13, 3, 1, 1), (3, 72, 16, 21, 3, 2, 1)])
def test_implicit_wgrad_several_tiles_and_ragged_rows(be, dev, B, Ci, Co, H, k, s, p):
    """More than one 256-column tile of taps x channels (a tile boundary inside a tap), more than one 256-row tile of output channels, several k-tiles with a ragged last one
    (rows not a multiple of 64), split-K: the gathered weight gradient equals dY^T . im2col(x) of the explicit matrix."""
    torch.manual_seed(3)
    OH = (H + 2 * p - k) // s + 1
    a = torch.randn(B, H, H, Ci).bfloat16().to(dev)
    dyb = torch.randn(B * OH * OH, Co).bfloat16().to(dev)
    col = ops.im2col(a, OH, OH, k, k, s, p, backend=be)
    want = dyb.float().cpu().t() @ col.float().cpu()
    for sk in (1, 3):
        got = ops.conv_wgrad_implicit(dyb, a, oh=OH, ow=OH, kh=k, kw=k, stride=s, pad=p, splitk=sk, backend=be)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=2e-3)


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


RESNET18_CONVS = [("conv1 7x7/2", 3, 64, 224, 7, 2, 3), ("layer1 3x3", 64, 64, 56, 3, 1, 1), ("layer2.0 3x3/2", 64, 128, 56, 3, 2, 1), ("layer2 3x3", 128, 128, 28, 3, 1, 1),
                  ("layer2 downsample 1x1/2", 64, 128, 56, 1, 2, 0), ("layer3.0 3x3/2", 128, 256, 28, 3, 2, 1), ("layer3 3x3", 256, 256, 14, 3, 1, 1),
                  ("layer3 downsample 1x1/2", 128, 256, 28, 1, 2, 0), ("layer4.0 3x3/2", 256, 512, 14, 3, 2, 1), ("layer4 3x3", 512, 512, 7, 3, 1, 1),
                  ("layer4 downsample 1x1/2", 256, 512, 14, 1, 2, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,Ci,Co,H,k,s,p", RESNET18_CONVS)
def test_resnet18_convs_at_bench_shapes_on_identical_bf16_operands(hip, name, Ci, Co, H, k, s, p):
    """Every convolution of ResNet-18 at BASELINE.json configs[0]'s shapes (batch 32, 224 x 224): forward, input gradient and weight gradient of the implicit-GEMM
    kernels against torch's fp32 convolution fed the SAME bf16-rounded operands, so what is left is fp32 summation order (1e-5 Frobenius-relative).  The end-to-end
    ResNet-18 parity test sits on a floor of 0.18 (train-mode BatchNorm + ReLU at random init in bf16 storage) and can only catch O(1) errors in the early layers;
    this one pins each layer's three products by themselves."""
    be, dev = hip, "cuda:0"
    torch.manual_seed(17)
    B, W = 32, H
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(B, Ci, H, W, device=dev).bfloat16().float()
    w = (torch.randn(Co, Ci, k, k, device=dev) * (2.0 / (Ci * k * k)) ** 0.5).bfloat16().float()
    dy = torch.randn(B, Co, OH, OH, device=dev).bfloat16().float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xr, wr, stride=s, padding=p)
    y.backward(dy)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    cip = (Ci + 7) // 8 * 8
    a = ops.nchw_to_nhwc_bf16(x, cip, backend=be)
    wf, wd = ops.conv_weight_prep(w, cip, backend=be)
    yk = ops.conv_gemm(a, wf, oh=OH, ow=OH, kh=k, kw=k, stride=s, pad=p, backend=be)
    e_fwd = rel(yk.view(B, OH, OH, Co).permute(0, 3, 1, 2), y.detach())
    dyb = dy.permute(0, 2, 3, 1).contiguous().bfloat16()
    dx = ops.conv_gemm(dyb, wd, oh=H, ow=W, kh=k, kw=k, stride=s, pad=p, transposed=True, backend=be)
    e_dgrad = rel(dx.view(B, H, W, cip)[..., :Ci].permute(0, 3, 1, 2), xr.grad)
    col = ops.im2col(a, OH, OH, k, k, s, p, backend=be)                                      # the engine's weight gradient: dY^T . im2col(x) on the TN kernel
    rows = B * OH * OH
    sk = max(1, min(16, rows // 4096))
    if rows % 64 == 0:
        dwp = ops.gemm_nt(dyb.view(rows, Co), col, out_dtype=torch.float32, trans=True, splitk=sk, backend=be)
    else:                                                                                    # ragged pixel counts (7 x 7 maps): explicit transposes + the NT kernel, like the engine
        rp = (rows + 63) // 64 * 64
        dwp = ops.gemm_nt(ops.transpose_pad(dyb.view(rows, Co), rpad=rp, backend=be), ops.transpose_pad(col, rpad=rp, backend=be), out_dtype=torch.float32, splitk=sk, backend=be)
    dw = ops.conv_wgrad_unpermute(dwp, Ci, k, k, backend=be)
    e_wgrad = rel(dw, wr.grad)
    # the form the engine runs: the im2col operand gathered inside the TN kernel
    dwi = ops.conv_wgrad_implicit(dyb.view(rows, Co), a, oh=OH, ow=OH, kh=k, kw=k, stride=s, pad=p, splitk=max(1, min(16, rows // 4096)), backend=be)
    e_wgrad_i = rel(ops.conv_wgrad_unpermute(dwi, Ci, k, k, backend=be), wr.grad)
    print(name, {"fwd": e_fwd, "dgrad": e_dgrad, "wgrad": e_wgrad, "wgrad_implicit": e_wgrad_i})
    assert e_fwd < 1e-5 and e_dgrad < 1e-5 and e_wgrad < 1e-5 and e_wgrad_i < 1e-5, (name, e_fwd, e_dgrad, e_wgrad, e_wgrad_i)
