"""faceX / CBIR model path: TimmWrapper (ViT features + neck) + margin head vs the oracle restatement (same weights)."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle.vit_ref import TimmWrapperRef
from visiondk_amd import face


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CFG = {"task": "cbir", "image_size": 32,
       "backbone": {"timm-vit_test_patch16": {"pretrained": False, "image_size": 32, "feat_dim": 64}},
       "head": {"arcface": {"feat_dim": 64, "num_class": 40, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}


class _RefArcFace(torch.nn.Module):   # arcface.py:20-36 restated (the golden test pins the HIP head against the real module)
    def __init__(self, w, m=0.35, s=32.0):
        super().__init__()
        self.weight = torch.nn.Parameter(w.clone()); self.m, self.s = m, s

    def forward(self, f, y):
        import math
        kn = torch.nn.functional.normalize(self.weight, dim=0); f = torch.nn.functional.normalize(f)
        c = (f @ kn).clamp(-1, 1)
        cm = c * math.cos(self.m) - torch.sqrt(1 - c ** 2) * math.sin(self.m)
        cm = torch.where(c > math.cos(math.pi - self.m), cm, c)
        idx = torch.zeros_like(c).scatter_(1, y.view(-1, 1), 1).bool()
        out = c * 1.0
        out[idx] = cm[idx]
        return out * self.s


def _build(be, dev):
    from visiondk_amd import vit
    vit.TIMM_VITS.setdefault("vit_test_patch16", dict(dim=128, depth=3, heads=2, mlp_dim=512))   # a 3-block ViT keeps the emulated run short
    torch.manual_seed(0)
    wrap = face.get_model(CFG, None, 0, backend=be, device=dev)
    model = wrap.model
    bb = model.trainingwrapper["backbone"]
    spec = bb.model.spec
    ref = TimmWrapperRef(64, spec.img_size, spec.patch_size, spec.dim, spec.depth, spec.heads, spec.mlp_dim)
    with torch.no_grad():
        for blk in ref.model.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                lin.weight.mul_(3.0)
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    sd = {k: v for k, v in ref.state_dict().items()}
    missing = bb.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    return model, ref


def test_state_dict_keys_match_reference_layout(be, dev):
    model, ref = _build(be, dev)
    keys = list(model.trainingwrapper["backbone"].state_dict().keys())
    assert keys == list(ref.state_dict().keys())
    assert "model.blocks.0.attn.qkv.weight" in keys and "output_layer.2.weight" in keys and "output_layer.3.running_var" in keys
    assert "trainingwrapper.head.weight" in model.state_dict() and "trainingwrapper.backbone.model.cls_token" in model.state_dict()


def test_face_forward_backward_vs_oracle(be, dev):
    model, ref = _build(be, dev)
    head = model.trainingwrapper["head"]
    rhead = _RefArcFace(head.weight.detach().cpu())
    torch.manual_seed(1)
    x = torch.randn(6, 3, 32, 32); y = torch.randint(0, 40, (6,))
    model.train(); ref.train()
    emb_ref = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(rhead(emb_ref, y), y)
    loss_ref.backward()
    logits = model(x.to(dev), y.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    bb = model.trainingwrapper["backbone"]
    got = dict(bb.named_parameters()); exp = dict(ref.named_parameters())
    assert set(got) == set(exp)
    gmax = max(exp[n].grad.norm().item() for n in exp)
    for n in exp:
        if exp[n].grad.norm().item() < 1e-5 * gmax:
            # analytically zero gradient (the neck LayerNorm bias: train-mode BatchNorm cancels any per-channel shift):
            # the reference holds fp32 round-off, we hold bf16 round-off; both must be negligible in absolute terms
            assert got[n].grad.norm().item() < 1e-3 * gmax, n
            continue
        r = _rel(got[n].grad, exp[n].grad)
        assert r < 8e-2, (n, r)
    assert _rel(head.weight.grad, rhead.weight.grad) < 8e-2
    # BatchNorm running statistics were updated like torch's
    assert _rel(bb.output_layer[3].running_mean, ref.output_layer[3].running_mean) < 2e-2
    assert _rel(bb.output_layer[3].running_var, ref.output_layer[3].running_var) < 2e-2


def test_extract_cbir_eval_embeddings(be, dev):
    model, ref = _build(be, dev)
    bb = model.trainingwrapper["backbone"]
    x = torch.randn(5, 3, 32, 32)
    ref.eval()
    with torch.no_grad():
        exp = torch.nn.functional.normalize(ref(x)).numpy()
    got = face.FeatureExtractor(bb).extract_cbir([x[:3], x[3:]], dev)
    assert got.shape == (5, 64) and got.dtype == np.float32
    assert _rel(got, exp) < 2e-2
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, rtol=1e-5)


# ---- CNN backbone (timm ConvNeXt) + BatchNorm2d neck: timm_wrapper.py:30-37 ----------------------------------------------------------
def _build_cnn(be, dev, monkeypatch, dims=(8, 16, 24, 32), img=32, operand="bf16"):
    from oracle.convnext_ref import TimmWrapperCNNRef
    from visiondk_amd import convnext
    depths = (1, 1, 2, 1)
    monkeypatch.setitem(convnext.TIMM_CONVNEXTS, "convnext_test", dict(depths=depths, dims=dims))
    cfg = {"task": "cbir", "image_size": img,
           "backbone": {"timm-convnext_test": {"pretrained": False, "image_size": img, "feat_dim": 64, "operand": operand}},
           "head": {"arcface": {"feat_dim": 64, "num_class": 40, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    torch.manual_seed(0)
    model = face.get_model(cfg, None, 0, backend=be, device=dev).model
    ref = TimmWrapperCNNRef(64, img, 3, depths, dims)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) * 0.5 + 0.25)
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
            elif "model." in n:
                p.copy_(torch.randn_like(p) * (0.5 / (p[0].numel() ** 0.5)))
    bb = model.trainingwrapper["backbone"]
    bb.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()}, strict=True)
    return model, ref, img


def test_cnn_backbone_state_dict_and_training_step(be, dev, monkeypatch):
    model, ref, img = _build_cnn(be, dev, monkeypatch)
    bb = model.trainingwrapper["backbone"]
    assert list(bb.state_dict().keys()) == list(ref.state_dict().keys())
    assert "model.stages.2.blocks.1.conv_dw.weight" in bb.state_dict() and "output_layer.0.running_var" in bb.state_dict()
    head = model.trainingwrapper["head"]
    rhead = _RefArcFace(head.weight.detach().cpu())
    torch.manual_seed(1)
    x = torch.randn(8, 3, img, img); y = torch.randint(0, 40, (8,))
    model.train(); ref.train()
    loss_ref = torch.nn.functional.cross_entropy(rhead(ref(x), y), y)
    loss_ref.backward()
    loss = torch.nn.functional.cross_entropy(model(x.to(dev), y.to(dev)), y.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    got = dict(bb.named_parameters()); exp = dict(ref.named_parameters())
    assert set(got) == set(exp)
    gmax = max(exp[n].grad.norm().item() for n in exp)
    for n in exp:
        if exp[n].grad.norm().item() < 1e-5 * gmax:
            assert got[n].grad.norm().item() < 1e-3 * gmax, n     # analytically zero (e.g. the Linear bias in front of a train-mode BatchNorm1d)
            continue
        r = _rel(got[n].grad, exp[n].grad)
        assert r < 8e-2, (n, r)
    assert _rel(head.weight.grad, rhead.weight.grad) < 8e-2
    for i in (0, 3):   # BatchNorm2d and BatchNorm1d running statistics follow torch's update
        assert _rel(bb.output_layer[i].running_mean, ref.output_layer[i].running_mean) < 2e-2
        assert _rel(bb.output_layer[i].running_var, ref.output_layer[i].running_var) < 2e-2


def test_cnn_neck_tall_batchnorm_path(be, dev, monkeypatch):
    """the neck's BatchNorm2d over many rows (25 088 at cfg3) runs as the slab-parallel kernel sequence: same loss, gradients and running statistics as the one-kernel form's test"""
    monkeypatch.setattr(face, "_BN_TALL_ROWS", 4)
    test_cnn_backbone_state_dict_and_training_step(be, dev, monkeypatch)
    test_cnn_extract_cbir_eval_embeddings(be, dev, monkeypatch)


def test_cnn_neck_wide_contraction(be, dev, monkeypatch):
    """K = HW * C = 1024 columns into the neck's Linear: it runs split over the contraction with the bias added after the combine (at cfg3's K = 50 176 the 4 output tiles
    would otherwise leave the whole contraction to 4 CUs).  Eval mode: BatchNorm1d does not cancel the bias there."""
    model, ref, img = _build_cnn(be, dev, monkeypatch, dims=(8, 16, 24, 256), img=64)
    bb = model.trainingwrapper["backbone"]
    with torch.no_grad():
        ref.output_layer[2].bias.add_(torch.randn(64) * 0.5)
    bb.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()}, strict=True)
    x = torch.randn(4, 3, img, img)
    ref.eval(); model.eval()
    with torch.no_grad():
        exp = ref(x)
        got = bb(x.to(dev))
    assert _rel(got, exp) < 2e-2
    ref.train(); model.train()
    exp = ref(x); exp.square().sum().backward()
    got = bb(x.to(dev)); got.square().sum().backward()
    assert _rel(got, exp) < 2e-2
    assert _rel(bb.output_layer[2].weight.grad, ref.output_layer[2].weight.grad) < 8e-2


def test_cnn_extract_cbir_eval_embeddings(be, dev, monkeypatch):
    model, ref, img = _build_cnn(be, dev, monkeypatch)
    bb = model.trainingwrapper["backbone"]
    x = torch.randn(5, 3, img, img)
    ref.eval()
    with torch.no_grad():
        exp = torch.nn.functional.normalize(ref(x)).numpy()
    got = face.FeatureExtractor(bb).extract_cbir([x[:3], x[3:]], dev)
    assert got.shape == (5, 64) and _rel(got, exp) < 2e-2


@pytest.mark.parametrize("layer_wise", [False, True])
def test_face_train_step_matches_reference_update(be, dev, monkeypatch, layer_wise):
    """layer_wise: the head's parameter group runs at 10 x lr (built/layer_optimizer.py:26-29, cbir.yaml:113).  FaceTrainStep == compute_loss(face=True) + Trainer.update on the oracle (CE -> backward -> clip_grad_norm_ -> SGD -> EMA), 2 steps,
    with the clip active (max_norm below the gradient norm)."""
    import math
    model, ref, img = _build_cnn(be, dev, monkeypatch)
    head = model.trainingwrapper["head"]
    rhead = _RefArcFace(head.weight.detach().cpu())
    bb = model.trainingwrapper["backbone"]
    lr, mom, wd, max_norm = 0.05, 0.9, 5e-4, 0.5
    step = face.FaceTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, max_norm=max_norm, ema=True, layer_wise=layer_wise)
    params = list(ref.parameters()) + [rhead.weight]
    opt = torch.optim.SGD([{"params": list(ref.parameters()), "lr": lr}, {"params": [rhead.weight], "lr": lr * 10 if layer_wise else lr}],
                          lr=lr, momentum=mom, weight_decay=wd)
    assert [g["lr"] for g in step.param_groups] == [g["lr"] for g in opt.param_groups][:len(step.param_groups)]
    ema_ref = {k: v.clone() for k, v in ref.state_dict().items() if v.dtype.is_floating_point}
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    model.train(); ref.train()
    torch.manual_seed(5)
    for it in range(2):
        x = torch.randn(8, 3, img, img); y = torch.randint(0, 40, (8,))
        opt.zero_grad()
        loss_ref = torch.nn.functional.cross_entropy(rhead(ref(x), y), y)
        loss_ref.backward()
        total = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
        assert total > max_norm          # the clip is active
        opt.step()
        d = 0.9999 * (1 - math.exp(-(it + 1) / 2000))
        for k, v in ref.state_dict().items():
            if v.dtype.is_floating_point:
                ema_ref[k].mul_(d).add_(v.detach(), alpha=1 - d)
        rows = step.step(x.to(dev), y.to(dev))
        assert abs(rows.mean().item() - loss_ref.item()) < 3e-2 * abs(loss_ref.item())
    got = dict(bb.named_parameters())
    worst = 0.0
    for n, p in ref.named_parameters():
        upd_ref = p.detach() - start[n]
        upd = got[n].detach().cpu() - start[n]
        if upd_ref.norm() < 1e-7:
            continue
        r = _rel(upd, upd_ref)
        worst = max(worst, r)
        # a batch-constant shift in front of a train-mode BatchNorm1d has an analytically zero gradient: those two updates are weight decay
        # plus round-off (fp32 in the oracle, bf16 here)
        tol = 0.3 if n in ("output_layer.0.bias", "output_layer.2.bias") else 0.12
        assert r < tol, (n, r)
    assert _rel(head.weight.detach(), rhead.weight.detach()) < 1e-3
    # EMA of a backbone tensor, a neck tensor and a BatchNorm running statistic
    eng = bb.model.engine
    name, off, numel, shape = eng.entries[5]
    assert _rel(step.ema_flat[off:off + numel].view(shape), ema_ref["model." + name]) < 1e-4
    small_names = [n for n, _ in bb.output_layer.named_parameters()]
    assert _rel(step.ema_small[small_names.index("2.weight")], ema_ref["output_layer.2.weight"]) < 1e-4
    buf_names = [n for n, b in bb.output_layer.named_buffers() if b.dtype.is_floating_point]
    assert _rel(step.ema_buf[buf_names.index("0.running_var")], ema_ref["output_layer.0.running_var"]) < 1e-3


def test_face_train_step_fp32_precision_matches_reference_update(be, dev, monkeypatch):
    """FaceTrainStep(precision="fp32"): the reference's face / CBIR loop has no autocast (engine/procedure/train.py:217-227), so its arithmetic is fp32.  Two clipped
    steps against torch fp32 (CE -> backward -> clip_grad_norm_ -> SGD): loss 1e-5, every non-degenerate parameter UPDATE within 2e-3 of the oracle's (bf16 mode: 0.12)."""
    model, ref, img = _build_cnn(be, dev, monkeypatch)
    head = model.trainingwrapper["head"]
    rhead = _RefArcFace(head.weight.detach().cpu())
    bb = model.trainingwrapper["backbone"]
    lr, mom, wd, max_norm = 0.05, 0.9, 5e-4, 0.5
    step = face.FaceTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, max_norm=max_norm, ema=False, precision="fp32")
    assert step.cos_planes == 3 and bb.model.engine.precision == "fp32"
    params = list(ref.parameters()) + [rhead.weight]
    opt = torch.optim.SGD(params, lr=lr, momentum=mom, weight_decay=wd)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    model.train(); ref.train()
    torch.manual_seed(5)
    for it in range(2):
        x = torch.randn(8, 3, img, img); y = torch.randint(0, 40, (8,))
        opt.zero_grad()
        loss_ref = torch.nn.functional.cross_entropy(rhead(ref(x), y), y)
        loss_ref.backward()
        total = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
        assert total > max_norm
        opt.step()
        rows = step.step(x.to(dev), y.to(dev))
        assert abs(rows.mean().item() - loss_ref.item()) < 2e-5 * abs(loss_ref.item()), (rows.mean().item(), loss_ref.item())
    got = dict(bb.named_parameters())
    for n, p in ref.named_parameters():
        upd_ref = p.detach() - start[n]
        if upd_ref.norm() < 1e-7:
            continue
        r = _rel(got[n].detach().cpu() - start[n], upd_ref)
        # (a batch-constant shift in front of a train-mode BatchNorm has an analytically zero gradient: those updates are weight decay plus round-off)
        tol = 0.3 if n in ("output_layer.0.bias", "output_layer.2.bias") else 2e-3
        assert r < tol, (n, r)
    assert _rel(head.weight.detach(), rhead.weight.detach()) < 1e-5


def test_face_train_step_fp16_operands_under_the_grad_scaler(be, dev, monkeypatch):
    """FaceTrainStep over an fp16 backbone (TimmWrapper(operand="fp16")): backbone, neck and head multiply fp16 operands and the step runs the GradScaler protocol the
    reference's face loop runs too (Trainer.update(model, loss, self.scaler, ...), engine/procedure/train.py:199,203-215).  Two clipped steps against torch fp32: loss within
    2e-3, every non-degenerate parameter UPDATE within 2e-2 of the oracle's (the bf16 step: 0.12), the head within 2e-4 (bf16: 1e-3); then a step whose loss scale
    overflows fp16 is skipped (weights and momentum untouched, scale halved, counter up) exactly as scaler.step / scaler.update do."""
    model, ref, img = _build_cnn(be, dev, monkeypatch, operand="fp16")
    head = model.trainingwrapper["head"]
    rhead = _RefArcFace(head.weight.detach().cpu())
    bb = model.trainingwrapper["backbone"]
    assert bb.model.engine.operand == "fp16" and bb.model.engine.wb16.dtype == torch.float16
    lr, mom, wd, max_norm = 0.05, 0.9, 5e-4, 0.5
    step = face.FaceTrainStep(model, lr=lr, momentum=mom, weight_decay=wd, max_norm=max_norm, ema=False, init_scale=1024.0)
    assert step.amp and step.loss_scale() == 1024.0
    params = list(ref.parameters()) + [rhead.weight]
    opt = torch.optim.SGD(params, lr=lr, momentum=mom, weight_decay=wd)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    model.train(); ref.train()
    torch.manual_seed(5)
    for it in range(2):
        x = torch.randn(8, 3, img, img); y = torch.randint(0, 40, (8,))
        opt.zero_grad()
        loss_ref = torch.nn.functional.cross_entropy(rhead(ref(x), y), y)
        loss_ref.backward()
        assert torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm) > max_norm
        opt.step()
        rows = step.step(x.to(dev), y.to(dev))
        assert abs(rows.mean().item() - loss_ref.item()) < 2e-3 * abs(loss_ref.item()), (rows.mean().item(), loss_ref.item())
    assert step.skipped_steps() == 0 and step.loss_scale() == 1024.0
    got = dict(bb.named_parameters())
    worst = (0.0, None)
    for n, p in ref.named_parameters():
        upd_ref = p.detach() - start[n]
        if upd_ref.norm() < 1e-7:
            continue
        r = _rel(got[n].detach().cpu() - start[n], upd_ref)
        tol = 0.3 if n in ("output_layer.0.bias", "output_layer.2.bias") else 2e-2
        assert r < tol, (n, r)
        if n not in ("output_layer.0.bias", "output_layer.2.bias"):
            worst = max(worst, (r, n))
    print("fp16 face step: worst update error", worst)
    assert _rel(head.weight.detach(), rhead.weight.detach()) < 2e-4
    # overflow: a loss scale beyond fp16's range makes the scaled gradients inf -> the step is skipped, the scale backs off
    sd = step.scaler_state_dict()
    assert sd["scale"] == 1024.0 and sd["_growth_tracker"] == 2 and sd["growth_interval"] == 2000
    step.load_scaler_state_dict(dict(sd, scale=2.0 ** 40))
    before = {n: p.detach().clone() for n, p in bb.named_parameters()}
    hw = head.weight.detach().clone(); mom0 = step.mom_flat.clone()
    step.step(x.to(dev), y.to(dev))
    assert step.skipped_steps() == 1 and step.loss_scale() == 2.0 ** 39
    for n, p in bb.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    assert torch.equal(head.weight.detach(), hw) and torch.equal(step.mom_flat, mom0)


def test_face_train_step_bf16_ignores_a_scaler_state_on_resume(be, dev, monkeypatch):
    """A bf16 backbone has no GradScaler: a scaler state restored from a checkpoint of an fp16 run (or of the reference: scale 65536, engine/vision_engine.py:296,397) must
    not reach loss_state -- vdk_sgd_step_amp divides by it while the bf16 passes never scale.  The step after such a resume equals the step without it, bit for bit."""
    runs = []
    for resume in (False, True):
        model, ref, img = _build_cnn(be, dev, monkeypatch, operand="bf16")
        step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=0.5, ema=False)
        assert not step.amp and step.scaler_state_dict() == {}
        if resume:
            step.load_scaler_state_dict({"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 7})
        assert step.loss_scale() == 1.0
        torch.manual_seed(5)
        x = torch.randn(8, 3, img, img); y = torch.randint(0, 40, (8,))
        step.step(x.to(dev), y.to(dev))
        bb = model.trainingwrapper["backbone"]
        runs.append({n: p.detach().clone() for n, p in bb.named_parameters()})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), n
