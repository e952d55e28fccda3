"""LayerNorm / reductions / losses / optimizer kernels on the CPU SIMT emulation vs plain torch fp32."""
import math

import pytest
import torch

from visiondk_amd import ops


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("T,C", [(10, 768), (7, 64), (33, 1024), (5, 1536), (21, 128), (9, 96), (17, 256), (150, 128), (77, 512), (210, 256), (300, 384)])
# (C <= 128: a row is half a wave; C <= 512: two rows per wave and pass in the backward; T > 64: several blocks, ragged last pass)
def test_layernorm_fwd_bwd(be, dev, T, C):
    torch.manual_seed(0)
    x = (torch.randn(T, C) * 2 + 0.5).to(dev)
    g = torch.randn(C).to(dev); b = torch.randn(C).to(dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, eps=1e-6, out_dtype=torch.float32, backend=be)
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (C,), gr, br, eps=1e-6)
    assert _rel(y, ref) < 1e-6
    yb, _, _ = ops.layernorm_fwd(x, g, b, eps=1e-6, out_dtype=torch.bfloat16, backend=be)
    assert torch.equal(yb, y.bfloat16())
    dy = torch.randn(T, C).bfloat16().to(dev)
    dres = torch.randn(T, C).to(dev)
    ref.backward(dy.float())
    dx, dxb, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres, backend=be)
    assert _rel(dx, xr.grad + dres) < 2e-6
    assert torch.equal(dxb, dx.bfloat16())
    assert _rel(dg, gr.grad) < 2e-6 and _rel(db, br.grad) < 2e-6
    # f32 dy, no residual
    dx2, _, dg2, _ = ops.layernorm_bwd(dy.float(), x, mean, rstd, g, dres=None, want_bf16=False, backend=be)
    assert _rel(dx2, xr.grad) < 2e-6


def test_layernorm_strided_rows(be, dev):
    # final norm applied to the cls rows only: row stride = N*C
    B, N, C = 3, 5, 64
    x = torch.randn(B, N, C).to(dev)
    g = torch.randn(C).to(dev); b = torch.randn(C).to(dev)
    y, _, _ = ops.layernorm_fwd(x, g, b, out_dtype=torch.float32, rows=B, ldx=N * C, backend=be)
    assert _rel(y, torch.nn.functional.layer_norm(x[:, 0], (C,), g, b, 1e-6)) < 1e-6


def test_reductions(be, dev):
    x = torch.randn(9, 5, 12).to(dev)
    assert _rel(ops.reduce_rows(x, 0.5, backend=be), x.sum(0) * 0.5) < 1e-6
    y = torch.randn(300, 72).bfloat16().to(dev)
    assert _rel(ops.colsum_bf16(y, backend=be), y.float().sum(0)) < 1e-5


@pytest.mark.parametrize("B,C,eps", [(6, 1000, 0.05), (3, 5, 0.0), (4, 257, 0.1)])
def test_softmax_ce(be, dev, B, C, eps):
    torch.manual_seed(1)
    logits = (torch.randn(B, C) * 3).to(dev)
    ya = torch.randint(0, C, (B,)).to(dev)
    lr = logits.clone().requires_grad_(True)
    ref_rows = torch.nn.functional.cross_entropy(lr, ya, label_smoothing=eps, reduction="none")
    ref_rows.mean().backward()
    loss, dlb, dlf = ops.softmax_ce(logits, ya, label_smoothing=eps, grad_scale=1.0 / B, backend=be)
    assert _rel(loss, ref_rows) < 1e-6
    assert _rel(dlf, lr.grad) < 1e-5
    assert torch.equal(dlb[:, :C], dlf.bfloat16()) and torch.count_nonzero(dlb[:, C:]) == 0
    # mixup pair: lam*CE(a) + (1-lam)*CE(b)   (train.py:34-35)
    yb = torch.randint(0, C, (B,)).to(dev); lam = 0.3
    lr2 = logits.clone().requires_grad_(True)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=eps, reduction="none")
    rows2 = lam * crit(lr2, ya) + (1 - lam) * crit(lr2, yb)
    rows2.mean().backward()
    loss2, _, dlf2 = ops.softmax_ce(logits, ya, yb, lam, eps, 1.0 / B, backend=be)
    assert _rel(loss2, rows2) < 1e-6 and _rel(dlf2, lr2.grad) < 1e-5


def test_bce(be, dev):
    torch.manual_seed(2)
    B, C = 8, 5
    logits = (torch.randn(B, C) * 2).to(dev)
    t = (torch.rand(B, C) > 0.5).float().to(dev)
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(lr, t)
    ref.backward()
    loss, _, dlf = ops.bce_logits(logits, t, grad_scale=1.0 / (B * C), backend=be)
    assert abs(loss.sum().item() / (B * C) - ref.item()) < 1e-6
    assert _rel(dlf, lr.grad) < 1e-5


@pytest.mark.parametrize("ps,H", [(16, 32), (14, 28), (4, 8)])
def test_patchify(be, dev, ps, H):
    x = torch.randn(2, 3, H, H).to(dev)
    p = ops.patchify(x, ps, backend=be)
    ref = torch.nn.functional.unfold(x, ps, stride=ps).transpose(1, 2).reshape(-1, 3 * ps * ps)  # k = c*ps*ps + ky*ps + kx
    K = 3 * ps * ps
    assert torch.equal(p[:, :K], ref.bfloat16()) and torch.count_nonzero(p[:, K:]) == 0


def test_casts(be, dev):
    w = torch.randn(20, 12).to(dev)
    assert torch.equal(ops.cast_bf16(w, backend=be), w.bfloat16())
    wt = ops.transpose_cast(w, backend=be)
    assert wt.shape == (12, 24) and torch.equal(wt[:, :20], w.T.bfloat16()) and torch.count_nonzero(wt[:, 20:]) == 0


def test_sgd_clip_ema_step_matches_torch(be, dev):
    torch.manual_seed(3)
    n = 1003
    p0 = torch.randn(n).to(dev); ema0 = p0.clone()
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p_ref], lr=0.006, momentum=0.937, weight_decay=5e-4)
    p = p0.clone(); m = torch.zeros(n, device=dev); ema = ema0.clone(); pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ema_ref = ema0.clone()
    for step in range(3):
        g = (torch.randn(n) * (5.0 if step == 1 else 0.1)).to(dev)   # step 1 triggers clipping (norm > 10)
        p_ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], max_norm=10.0)
        opt.step()
        d = 0.9999 * (1 - math.exp(-(step + 1) / 2000))
        ema_ref = ema_ref * d + (1 - d) * p_ref.detach()
        nsq = ops.sumsq(g, backend=be)
        assert abs(nsq.item() - (g.double() ** 2).sum().item()) / (g.double() ** 2).sum().item() < 1e-6
        ops.sgd_step(p, g, m, lr=0.006, momentum=0.937, weight_decay=5e-4, ema=ema, p_bf16=pb, normsq=nsq, max_norm=10.0,
                     ema_decay=d, first_step=(step == 0), backend=be)
        assert _rel(p, p_ref.detach()) < 1e-6
        assert _rel(ema, ema_ref) < 1e-6
        assert torch.equal(pb, p.bfloat16())


def test_mixup(be, dev):
    x = torch.randn(4, 3, 8, 8).to(dev)
    perm = torch.tensor([2, 0, 3, 1]).to(dev)
    out = ops.mixup(x, perm, 0.3, backend=be)
    assert _rel(out, 0.3 * x + 0.7 * x[perm]) < 1e-6


def test_ohem_mask_matches_reference_sampler(be, dev):
    import importlib.util
    from pathlib import Path
    ref_path = Path("/root/reference/structure/sampler.py")
    torch.manual_seed(4)
    logits = torch.randn(37, 10) * 2; labels = torch.randint(0, 10, (37,)); labels[5] = 255
    for min_kept, thresh in [(8, 0.2), (100, 0.05), (0, 0.5)]:
        got = ops.ohem_mask(logits.to(dev), labels.to(dev), min_kept, thresh, backend=be).cpu()
        # restatement of OHEMImageSampler.sample (structure/sampler.py:11-31)
        prob = torch.softmax(logits, 1); v1 = labels != 255
        tp = prob[v1].gather(1, labels[v1].unsqueeze(1)).squeeze(1)
        sp, si = tp.sort()
        thr = max(sp[min(min_kept, sp.numel() - 1)].item(), thresh)
        v2 = torch.zeros_like(labels, dtype=torch.bool)
        v2[torch.nonzero(v1).squeeze(1)[si[sp < thr]]] = True
        exp = v1 & v2
        if ref_path.exists():   # in the build container also run the reference's own class
            spec = importlib.util.spec_from_file_location("ref_sampler", ref_path); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
            if not (labels == 255).any():
                exp = mod.OHEMImageSampler(min_kept, thresh).sample(logits, labels)
        assert torch.equal(got, exp), (min_kept, thresh)


def test_topk_rows(be, dev):
    torch.manual_seed(5)
    x = torch.randn(9, 1000); x[2, 7] = x[2, 3] = 9.0      # tie -> lower index first
    val, idx = ops.topk_rows(x.to(dev), 5, backend=be)
    rv, ri = torch.topk(x, 5, dim=1)
    assert torch.equal(val.cpu(), rv)
    assert idx[2, 0].item() == 3 and idx[2, 1].item() == 7
    assert torch.equal(idx.cpu()[[0, 1, 3, 4, 5, 6, 7, 8]], ri[[0, 1, 3, 4, 5, 6, 7, 8]])


@pytest.mark.parametrize("B,F", [(40, 72), (300, 100), (1000, 33)])   # small-batch kernel and the row-parallel kernel (B >= 256)
def test_batchnorm_rows_fwd_bwd(be, dev, B, F):
    torch.manual_seed(B)
    x = torch.randn(B, F) * 2 + 0.5
    bn = torch.nn.BatchNorm1d(F)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(F) + 0.5); bn.bias.copy_(torch.randn(F) * 0.1)
        bn.running_mean.copy_(torch.randn(F) * 0.1); bn.running_var.copy_(torch.rand(F) + 0.5)
    rm, rv = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev)
    xr = x.clone().requires_grad_(True)
    y = bn(xr)
    dy = torch.randn_like(y)
    y.backward(dy)
    yk, sm, si = ops.batchnorm_fwd(x.to(dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev), rm, rv, training=True, backend=be)
    torch.testing.assert_close(yk.cpu(), y.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    dx, dg, db = ops.batchnorm_bwd(dy.to(dev), x.to(dev), bn.weight.detach().to(dev), sm, si, backend=be)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), bn.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), bn.bias.grad, rtol=1e-4, atol=1e-4)
    bn.eval()
    ye, _, _ = ops.batchnorm_fwd(x.to(dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev), rm, rv, training=False, backend=be)
    torch.testing.assert_close(ye.cpu(), bn(x).detach(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("S,n", [(37, 130), (256, 40), (1004, 1024), (1024, 200), (2047, 50), (2500, 72)])
def test_batch_reduction_folds_tall_jobs(be, dev, S, n):
    """the engines' batched row reduction: 256 .. 2047 partial rows go to the 16-column x 32-row-group form, from 2048 on the rows are folded onto the first 64 in place
    (more workgroups than the n / 64 of the final sum); against a float64 sum, and bit-reproducible"""
    torch.manual_seed(S)
    ld = n + 6
    src = torch.randn(S, ld)
    want = src[:, :n].double().sum(0) * 0.5
    outs = []
    for _ in range(2):
        buf = src.clone().to(dev); out = torch.empty(n, device=dev)
        be.check(be.lib.vdk_debug_reduce_rows_job(be.ptr(buf), ld, S, n, be.ptr(out), 0.5, be.stream()), "vdk_debug_reduce_rows_job")
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    torch.testing.assert_close(outs[0].double(), want, rtol=1e-5, atol=1e-4)

