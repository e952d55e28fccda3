"""Native ConvNeXt engine (csrc/convnext_engine.hip) vs the pinned oracle (oracle/convnext_ref.py): forward map and every
parameter gradient, on the CPU SIMT emulation and (-m gpu) on the MI355X through the same C ABI."""
import pytest
import torch

from oracle.convnext_ref import ConvNeXtRef
from visiondk_amd import convnext


def _pair(be, dev, depths, dims, img, seed=0, num_classes=0, operand="bf16", gamma=None):
    spec = convnext.ConvNeXtSpec(img_size=img, depths=depths, dims=dims, num_classes=num_classes)
    model = convnext.ConvNeXt(spec, device=dev, backend=be, seed=seed, operand=operand)
    ref = ConvNeXtRef(3, depths, dims, num_classes=num_classes)
    torch.manual_seed(seed)
    with torch.no_grad():   # non-trivial values everywhere (timm's init leaves biases at 0 and gamma at 1e-6: gradients would hide bugs)
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) * 0.5 + 0.25 if gamma is None else gamma * (1.0 + torch.rand_like(p)))
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
            else:
                p.copy_(torch.randn_like(p) * (0.5 / (p[0].numel() ** 0.5)))
    missing, unexpected = model.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    return model, ref


def test_state_dict_names_match_timm_layout(be, dev):
    model, ref = _pair(be, dev, (1, 1, 2, 1), (8, 16, 24, 32), 32)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert model.state_dict()[k].shape == v.shape, k


@pytest.mark.parametrize("B,img,depths,dims", [(2, 32, (1, 1, 2, 1), (8, 16, 24, 32)), (4, 64, (2, 1, 1, 1), (16, 32, 64, 72))])
def test_forward_backward_vs_oracle(be, dev, B, img, depths, dims):
    model, ref = _pair(be, dev, depths, dims, img)
    torch.manual_seed(3)
    x = torch.randn(B, 3, img, img)
    y = model(x.to(dev))
    yr = ref(x)
    assert y.shape == yr.shape
    rel = ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item()
    assert rel < 2e-2, rel            # bf16 GEMM operands vs the fp32 oracle
    dy = torch.randn_like(yr)
    y.backward(dy.to(dev))
    yr.backward(dy)
    worst = []
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        g, gr = p.grad.detach().cpu(), pr.grad
        r = ((g - gr).norm() / (gr.norm() + 1e-12)).item()
        worst.append((r, n))
        assert r < 6e-2, (n, r)
    # the bulk is far better than the bound
    worst.sort()
    assert worst[len(worst) // 2][0] < 2e-2, worst[len(worst) // 2]


@pytest.mark.parametrize("B,img,depths,dims,ncls", [(3, 32, (1, 1, 2, 1), (8, 16, 24, 32), 5), (6, 64, (1, 1, 1, 2), (16, 32, 64, 72), 37)])
def test_classifier_head_vs_oracle(be, dev, B, img, depths, dims, ncls):
    """timm.create_model('convnext_*', num_classes=N) as VisionWrapper builds it: global average pool -> head.norm -> head.fc; logits, CE loss and every
    parameter gradient (incl. the padded fc rows staying out of the state_dict) against the oracle."""
    model, ref = _pair(be, dev, depths, dims, img, num_classes=ncls)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    assert model.state_dict()["head.fc.weight"].shape == (ncls, dims[3])
    torch.manual_seed(4)
    x = torch.randn(B, 3, img, img)
    t = torch.randint(0, ncls, (B,))
    y = model(x.to(dev))
    yr = ref(x)
    assert y.shape == (B, ncls)
    assert ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item() < 2e-2
    loss = torch.nn.functional.cross_entropy(y, t.to(dev))
    loss_r = torch.nn.functional.cross_entropy(yr, t)
    assert abs(loss.item() - loss_r.item()) < 2e-2 * abs(loss_r.item())
    loss.backward(); loss_r.backward()
    worst = []
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        r = ((p.grad.detach().cpu() - pr.grad).norm() / (pr.grad.norm() + 1e-12)).item()
        worst.append((r, n))
        assert r < 8e-2, (n, r)
    worst.sort()
    assert worst[len(worst) // 2][0] < 3e-2, worst[len(worst) // 2]


def test_classifier_train_step_matches_reference_update(be, dev):
    """get_model(task=classification, name=timm-convnext_*) -> VisionWrapper -> ClassifierTrainStep: CE(label_smoothing) -> backward -> clip_grad_norm_ ->
    SGD(momentum, wd) -> EMA against the same sequence on the oracle (one step, clip active)."""
    import math
    from visiondk_amd import resnet
    ncls, img = 6, 32
    model, ref = _pair(be, dev, (1, 1, 1, 1), (8, 16, 24, 32), img, num_classes=ncls)
    lr, mom, wd, max_norm = 0.05, 0.9, 5e-4, 0.05
    step = resnet.ClassifierTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, loss="ce", label_smoothing=0.1, max_norm=max_norm, ema=True)
    opt = torch.optim.SGD(ref.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    torch.manual_seed(5)
    x = torch.randn(5, 3, img, img); t = torch.randint(0, ncls, (5,))
    loss_r = torch.nn.functional.cross_entropy(ref(x), t, label_smoothing=0.1)
    loss_r.backward()
    assert torch.nn.utils.clip_grad_norm_(list(ref.parameters()), max_norm=max_norm) > max_norm
    opt.step()
    rows = step.step(x.to(dev), t.to(dev))
    assert abs(rows.mean().item() - loss_r.item()) < 2e-2 * abs(loss_r.item())
    got = dict(model.named_parameters())
    for n, p in ref.named_parameters():
        upd_ref, upd = p.detach() - start[n], got[n].detach().cpu() - start[n]
        r = ((upd - upd_ref).norm() / (upd_ref.norm() + 1e-12)).item()
        assert r < 0.12, (n, r)
    d = 0.9999 * (1 - math.exp(-1 / 2000))
    name, off, numel, shape = model.engine.entries[-2]                      # head.fc.weight: EMA = d * start + (1 - d) * updated
    exp = d * start[name] + (1 - d) * ref.state_dict()[name]
    got_ema = step.ema[off:off + numel].view(shape).cpu()
    assert ((got_ema - exp).norm() / exp.norm()).item() < 1e-4


def test_vision_wrapper_routes_convnext_classifier(be, dev, monkeypatch):
    from visiondk_amd import face
    monkeypatch.setitem(convnext.TIMM_CONVNEXTS, "convnext_tiny", dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32)))   # a miniature under the timm id
    cfg = {"task": "classification", "name": "timm-convnext_tiny.in12k_ft_in1k", "image_size": 32, "num_classes": 4, "pretrained": False, "kwargs": {}}
    wrap = face.VisionWrapper(cfg, None, 0, backend=be, device=dev)
    y = wrap.model(torch.randn(2, 3, 32, 32).to(dev))
    assert y.shape == (2, 4) and "head.fc.weight" in wrap.model.state_dict()
    with pytest.raises(NotImplementedError):       # options of the reference's model config that are not built fail loudly instead of being ignored
        face.VisionWrapper(dict(cfg, backbone_freeze=True), None, 0, backend=be, device=dev)


def test_classifier_sam_and_mixup_step_match_reference_sequence(be, dev):
    """update_sam (train.py:150-175) with a CE mixup pair: loss at w -> e(w) = rho * w^2 * g / ||w * g|| -> loss at w + e(w) -> back to w -> SGD with the second
    gradient (no clipping) -> the FIRST loss is returned."""
    from visiondk_amd import resnet
    ncls, img = 6, 32
    model, ref = _pair(be, dev, (1, 1, 1, 1), (8, 16, 24, 32), img, num_classes=ncls)
    lr, mom, wd, rho = 0.05, 0.9, 5e-4, 0.05
    step = resnet.ClassifierTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, loss="ce", label_smoothing=0.05, ema=False, sam=True, sam_rho=rho)
    params = list(ref.parameters())
    start = [p.detach().clone() for p in params]
    opt = torch.optim.SGD(params, lr=lr, momentum=mom, weight_decay=wd)
    torch.manual_seed(6)
    x = torch.randn(5, 3, img, img); ya = torch.randint(0, ncls, (5,)); yb = torch.randint(0, ncls, (5,)); lam = 0.3
    ce = lambda out: lam * torch.nn.functional.cross_entropy(out, ya, label_smoothing=0.05) + (1 - lam) * torch.nn.functional.cross_entropy(out, yb, label_smoothing=0.05)
    loss1 = ce(ref(x)); loss1.backward()
    with torch.no_grad():
        norm = torch.stack([(p.abs() * p.grad).norm(2) for p in params]).norm(2)
        for p in params:
            p.add_(p.pow(2) * p.grad * (rho / (norm + 1e-12)))
    opt.zero_grad()
    ce(ref(x)).backward()
    with torch.no_grad():
        for p, s0 in zip(params, start):
            p.copy_(s0)
    opt.step()
    rows = step.step(x.to(dev), ya.to(dev), yb.to(dev), lam)
    assert abs(rows.mean().item() - loss1.item()) < 2e-2 * abs(loss1.item())
    got = dict(model.named_parameters())
    for (n, p), s0 in zip(ref.named_parameters(), start):
        upd_ref, upd = p.detach() - s0, got[n].detach().cpu() - s0
        r = ((upd - upd_ref).norm() / (upd_ref.norm() + 1e-12)).item()
        assert r < 0.12, (n, r)


def test_classifier_takes_other_resolutions(be, dev):
    """progressive resizing (vision_engine.py:181-222): the classifier's global-average-pool head is resolution-agnostic, so the engine follows the input size"""
    model, ref = _pair(be, dev, (1, 1, 1, 1), (8, 16, 24, 32), 64, num_classes=5)
    for img in (32, 96, 64):
        x = torch.randn(2, 3, img, img)
        y = model(x.to(dev))
        yr = ref(x)
        assert ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item() < 2e-2, img
    y.sum().backward()
    with pytest.raises(ValueError):
        model(torch.randn(2, 3, 48, 48).to(dev))


@pytest.mark.parametrize("B,img,depths,dims", [(2, 32, (1, 1, 2, 1), (8, 16, 24, 32)), (4, 64, (2, 1, 1, 1), (16, 32, 64, 72))])
def test_fp32_training_mode_vs_oracle(be, dev, B, img, depths, dims):
    """engine.precision = "fp32" (vdk_convnext_forward_train_f32 / vdk_convnext_backward_train_f32): fp32 activations, every contraction on the fp32 MFMA -- the arithmetic of the
    reference's face / CBIR loop (no autocast, engine/procedure/train.py:217-227).  Forward map and EVERY parameter gradient against the fp32 oracle at fp32 tolerances."""
    model, ref = _pair(be, dev, depths, dims, img)
    model.engine.precision = "fp32"
    torch.manual_seed(3)
    x = torch.randn(B, 3, img, img)
    y = model(x.to(dev))
    yr = ref(x)
    rel = ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item()
    assert rel < 2e-6, rel
    dy = torch.randn_like(yr)
    y.backward(dy.to(dev))
    yr.backward(dy)
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        g, gr = p.grad.detach().cpu(), pr.grad
        r = ((g - gr).norm() / (gr.norm() + 1e-12)).item()
        assert r < 2e-5, (n, r)


@pytest.mark.parametrize("gamma", [None, 1e-6])
def test_fp16_operands_forward_backward_vs_oracle(be, dev, gamma):
    """VdkConvNextConfig.operand = VDK_F16: 8x less operand rounding than bf16 -- map 6e-4 / gradients <= 2e-3 from the fp32 oracle here (bf16: 4e-3 / 1.2e-2) -- also at
    timm's layer-scale init gamma = 1e-6, where gamma (.) W2 would underflow fp16: the forward applies gamma in the fc2 epilogue (VdkGemmDesc.col_scale), the backward runs the
    branch at the block's power-of-two scale r and un-scales in LayerNorm backward (dy_scale) and over the fc1 gradients.  The output gradient carries a loss scale as
    under GradScaler; the engine's gradients carry it too."""
    model, ref = _pair(be, dev, (1, 1, 2, 1), (8, 16, 24, 32), 32, operand="fp16", gamma=gamma)
    assert model.engine.wb16.dtype == torch.float16
    torch.manual_seed(3)
    x = torch.randn(2, 3, 32, 32)
    y = model(x.to(dev)); yr = ref(x)
    assert ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item() < 1.5e-3
    dy = torch.randn_like(yr)
    S = 1024.0
    y.backward((dy * S).to(dev)); yr.backward(dy)
    worst = []
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        r = ((p.grad.detach().cpu() / S - pr.grad).norm() / (pr.grad.norm() + 1e-30)).item()
        worst.append((r, n))
        assert r < 5e-3, (n, r)
    worst.sort()
    assert worst[len(worst) // 2][0] < 2e-3, worst[len(worst) // 2]


def test_fp16_classifier_train_step_runs_the_grad_scaler_protocol(be, dev):
    """ClassifierTrainStep over an fp16 ConvNeXt classifier (the reference's autocast dtype, train.py:118) == CE -> scaled backward -> unscale -> clip -> SGD on the fp32
    oracle, updates within 2e-2 (bf16: 0.12); an overflowing scale skips the step and halves the scale (scaler.step / scaler.update, train.py:209-211)."""
    from visiondk_amd import resnet
    ncls, img = 6, 32
    model, ref = _pair(be, dev, (1, 1, 1, 1), (8, 16, 24, 32), img, num_classes=ncls, operand="fp16")
    lr, mom, wd, max_norm = 0.05, 0.9, 5e-4, 0.05
    step = resnet.ClassifierTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, loss="ce", label_smoothing=0.1, max_norm=max_norm, ema=False, init_scale=4096.0)
    assert step.amp
    opt = torch.optim.SGD(ref.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    torch.manual_seed(5)
    x = torch.randn(5, 3, img, img); t = torch.randint(0, ncls, (5,))
    loss_r = torch.nn.functional.cross_entropy(ref(x), t, label_smoothing=0.1)
    loss_r.backward()
    assert torch.nn.utils.clip_grad_norm_(list(ref.parameters()), max_norm=max_norm) > max_norm
    opt.step()
    rows = step.step(x.to(dev), t.to(dev))
    assert abs(rows.mean().item() - loss_r.item()) < 2e-3 * abs(loss_r.item())
    assert step.skipped_steps() == 0
    got = dict(model.named_parameters())
    for n, p in ref.named_parameters():
        upd_ref, upd = p.detach() - start[n], got[n].detach().cpu() - start[n]
        r = ((upd - upd_ref).norm() / (upd_ref.norm() + 1e-12)).item()
        assert r < 2e-2, (n, r)
    step.load_scaler_state_dict(dict(step.scaler_state_dict(), scale=2.0 ** 40))
    before = model.engine.params.clone()
    step.step(x.to(dev), t.to(dev))
    assert step.skipped_steps() == 1 and step.loss_scale() == 2.0 ** 39 and torch.equal(model.engine.params, before)
