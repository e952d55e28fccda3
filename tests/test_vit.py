"""Hot path A end to end: the native ViT engine (forward, backward, fused step) vs the fp32 oracle restatement of
timm's VisionTransformer, same weights, same inputs.  bf16 MFMA operands vs an fp32 oracle: tolerances are stated
per check; a wiring mistake (missing residual/bias, transposed weight, wrong gradient) shows up as an O(1) error."""
import copy

import pytest
import torch

from oracle.vit_ref import VisionTransformerRef, train_step_reference
from visiondk_amd import vit


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


SPEC = vit.VitSpec(img_size=32, patch_size=8, in_chans=3, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, ln_eps=1e-6)


def _pair(be, dev, seed=0):
    torch.manual_seed(seed)
    ref = VisionTransformerRef(SPEC.img_size, SPEC.patch_size, 3, SPEC.num_classes, SPEC.dim, SPEC.depth, SPEC.heads, SPEC.mlp_dim)
    with torch.no_grad():  # non-trivial biases / norms / cls so every gradient path is exercised
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.02)
        for blk in ref.blocks:   # larger weights -> activations of O(1), a more demanding test than N(0, .02)
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                lin.weight.mul_(4.0)
    model = vit.VisionTransformer(SPEC, device=dev, backend=be, seed=1)
    model.load_state_dict(ref.state_dict())
    return ref, model


def test_param_names_and_shapes(be, dev):
    ref, model = _pair(be, dev)
    sd_ref, sd = ref.state_dict(), model.state_dict()
    assert list(sd.keys()) == list(sd_ref.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(sd_ref[k].shape), k
        assert torch.equal(sd[k].cpu(), sd_ref[k])
    assert [n for n, _ in model.named_parameters()] == list(sd_ref.keys())


def test_forward_backward_vs_oracle(be, dev):
    ref, model = _pair(be, dev)
    torch.manual_seed(5)
    x = torch.randn(3, 3, 32, 32)
    y = torch.randint(0, 10, (3,))
    logits_ref = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, y, label_smoothing=0.05)
    loss_ref.backward()
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.05)
    loss.backward()
    assert _rel(logits, logits_ref) < 2e-2, _rel(logits, logits_ref)          # bf16 operands, fp32 accumulate
    assert abs(loss.item() - loss_ref.item()) < 5e-3 * abs(loss_ref.item())
    worst = 0.0
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        assert p.grad is not None, n
        r = _rel(p.grad, pr.grad)
        worst = max(worst, r)
        assert r < 6e-2, (n, r)
    print("worst grad rel err", worst)


def test_fused_step_vs_reference_step(be, dev):
    ref, model = _pair(be, dev, seed=3)
    hyp = dict(lr=0.01, momentum=0.937, weight_decay=5e-4)
    step = vit.FusedTrainStep(model, label_smoothing=0.05, max_norm=10.0, ema=True, **hyp)
    ema_ref = {n: p.detach().clone() for n, p in ref.named_parameters()}
    init_sd = {n: p.detach().clone() for n, p in ref.named_parameters()}
    bufs = None
    torch.manual_seed(11)
    for it in range(3):
        x = torch.randn(4, 3, 32, 32)
        y = torch.randint(0, 10, (4,))
        _, loss_ref, _, _, bufs = train_step_reference(ref, x, y, label_smoothing=0.05, max_norm=10.0, momentum_bufs=bufs, ema=ema_ref,
                                                       updates=it, **hyp)
        step.step(x.to(dev), y.to(dev))
        assert abs(step.loss_value() - loss_ref.item()) < 1e-2 * abs(loss_ref.item()), (it, step.loss_value(), loss_ref.item())
    # compare the UPDATE each tensor received over the 3 steps (weights barely move, so comparing weights proves little)
    sd = model.state_dict()
    for n, p in ref.named_parameters():
        d_ref = p.detach() - init_sd[n]
        d_got = sd[n].cpu() - init_sd[n]
        assert _rel(d_got, d_ref) < 8e-2, (n, _rel(d_got, d_ref))
    for n in ema_ref:
        got = model.engine.view(step.ema, n).cpu()
        assert _rel(got - init_sd[n], ema_ref[n] - init_sd[n]) < 8e-2, n


def test_mixup_pair_loss(be, dev):
    ref, model = _pair(be, dev, seed=4)
    step = vit.FusedTrainStep(model, lr=0.0, momentum=0.0, weight_decay=0.0, label_smoothing=0.1, ema=False)
    x = torch.randn(4, 3, 32, 32); ya = torch.randint(0, 10, (4,)); yb = torch.randint(0, 10, (4,)); lam = 0.35
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    out = ref(x)
    loss_ref = lam * crit(out, ya) + (1 - lam) * crit(out, yb)      # mixup_criterion, train.py:34-35
    step.step(x.to(dev), ya.to(dev), yb.to(dev), lam)
    assert abs(step.loss_value() - loss_ref.item()) < 1e-2 * abs(loss_ref.item())


def test_deepcopy_for_ema_and_eval(be, dev):
    ref, model = _pair(be, dev)
    m2 = copy.deepcopy(model)              # ModelEMA does this (models/ema.py:22)
    x = torch.randn(2, 3, 32, 32).to(dev)
    with torch.no_grad():
        a, b = model(x), m2(x)
    assert torch.equal(a.cpu(), b.cpu())


def test_forward_backward_tn_wgrad_path(be, dev):
    """B=64, N=17: B*N and B*np are multiples of 64, so every wgrad (incl. the patch embedding with its cls-row remap)
    takes the TN LDS-DMA kernel instead of the transpose fallback."""
    ref, model = _pair(be, dev, seed=8)
    torch.manual_seed(9)
    x = torch.randn(64, 3, 32, 32)
    y = torch.randint(0, 10, (64,))
    loss_ref = torch.nn.functional.cross_entropy(ref(x), y, label_smoothing=0.05)
    loss_ref.backward()
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.05)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 5e-3 * abs(loss_ref.item())
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        r = _rel(p.grad, pr.grad)
        assert r < 6e-2, (n, r)


def test_fused_sam_step_vs_reference_update_sam(be, dev):
    from oracle.vit_ref import train_step_reference_sam
    ref, model = _pair(be, dev, seed=6)
    hyp = dict(lr=0.01, momentum=0.937, weight_decay=5e-4)
    step = vit.FusedTrainStep(model, label_smoothing=0.05, ema=False, sam=True, sam_rho=0.05, **hyp)
    init_sd = {n: p.detach().clone() for n, p in ref.named_parameters()}
    bufs = None
    torch.manual_seed(12)
    for it in range(2):
        x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
        loss_ref, bufs = train_step_reference_sam(ref, x, y, label_smoothing=0.05, rho=0.05, momentum_bufs=bufs, **hyp)
        step.step(x.to(dev), y.to(dev))
        assert abs(step.loss_value() - loss_ref.item()) < 1e-2 * abs(loss_ref.item())
    sd = model.state_dict()
    for n, p in ref.named_parameters():
        assert _rel(sd[n].cpu() - init_sd[n], p.detach() - init_sd[n]) < 8e-2, n


def test_backward_bias_gradients_from_dgrad_gemm_byproduct(be, dev):
    """With T % 256 == 0 and the 256x256 kernel in use, the Linear bias gradients come out of the dgrad GEMMs (VdkGemmDesc.a_colsum) instead of a separate
    pass over dY: same gradients as the stand-alone column-sum path."""
    spec = vit.VitSpec(img_size=32, patch_size=16, in_chans=3, num_classes=8, dim=64, depth=2, heads=1, mlp_dim=256, ln_eps=1e-6)
    model = vit.VisionTransformer(spec, device=dev, backend=be, seed=3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
    eng = model.engine
    torch.manual_seed(0)
    x = torch.randn(256, 3, 32, 32, device=dev)                       # T = 256 * 5 tokens: a whole number of 256-row tiles
    dl = (torch.randn(256, eng.cp, device=dev) * 0.1).bfloat16()
    grads = {}
    for mode in (1, 2):                                               # 1: 128x128 kernel -> stand-alone colsum; 2: 256x256 kernel -> by-product
        be.lib.vdk_gemm_force_kernel(mode)
        try:
            eng.forward(x)
            grads[mode] = eng.backward(dl).clone()
        finally:
            be.lib.vdk_gemm_force_kernel(0)
    for name, off, numel, shape in eng.entries:
        a, b = grads[2][off:off + numel], grads[1][off:off + numel]
        rel = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert rel < (2e-3 if name.endswith("bias") else 2e-2), (name, rel)


def test_patch14_padded_operand_copies_vs_oracle(be, dev):
    """timm's patch-14 models (vit_large_patch14_224, pet.yaml's *_patch14_* ids): in_chans * 14 * 14 = 588 is not a multiple of 8, so the patch-embedding GEMM
    reads zero-padded copies ([D, 592] weight, [rows, 592] patches) while the parameter and its gradient keep their [D, 3, 14, 14] shape."""
    spec = vit.VitSpec(img_size=42, patch_size=14, in_chans=3, num_classes=7, dim=64, depth=1, heads=1, mlp_dim=128, ln_eps=1e-6)
    torch.manual_seed(2)
    ref = VisionTransformerRef(spec.img_size, spec.patch_size, 3, spec.num_classes, spec.dim, spec.depth, spec.heads, spec.mlp_dim)
    with torch.no_grad():
        ref.patch_embed.proj.weight.mul_(3.0)
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    model = vit.VisionTransformer(spec, device=dev, backend=be, seed=1)
    model.load_state_dict(ref.state_dict())
    assert tuple(model.state_dict()["patch_embed.proj.weight"].shape) == (64, 3, 14, 14)
    x = torch.randn(3, 3, 42, 42); y = torch.randint(0, 7, (3,))
    lr_ = torch.nn.functional.cross_entropy(ref(x), y); lr_.backward()
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev)); loss.backward()
    assert abs(loss.item() - lr_.item()) < 1e-2 * abs(lr_.item())
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr and p.grad.shape == pr.grad.shape
        assert _rel(p.grad, pr.grad) < 6e-2, (n, _rel(p.grad, pr.grad))
    with pytest.raises(Exception):
        model.forward_precise(x.to(dev))        # the fp32-MFMA path keeps the K % 8 requirement


def test_ohem_prepass_then_step_on_the_kept_samples(be, dev):
    """train.py:113-117: valid = sampler.sample(model(images), labels); images, labels = images[valid], labels[valid]; then the ordinary step on the subset"""
    ref, model = _pair(be, dev)
    step = vit.FusedTrainStep(model, lr=0.01, momentum=0.9, weight_decay=5e-4, label_smoothing=0.0, ema=False)
    torch.manual_seed(11)
    x = torch.randn(12, 3, 32, 32); y = torch.randint(0, 10, (12,))
    with torch.no_grad():
        prob = torch.softmax(ref(x), 1)
    tp = prob.gather(1, y.unsqueeze(1)).squeeze(1)
    sp, si = tp.sort()
    min_kept, thresh = 5, 0.05
    thr = max(sp[min(min_kept, sp.numel() - 1)].item(), thresh)
    keep_ref = torch.zeros(12, dtype=torch.bool); keep_ref[si[sp < thr]] = True
    xs, ys = step.ohem_select(x.to(dev), y.to(dev), min_kept, thresh)
    margin = (tp - thr).abs().min().item()
    if margin > 2e-3:                                     # bf16 logits: only compare when no sample sits on the threshold
        assert torch.equal(ys.cpu(), y[keep_ref]) and xs.shape[0] == int(keep_ref.sum())
    assert 0 < xs.shape[0] < 12
    rows = step.step(xs, ys)                              # a batch size the engine has not seen: the workspace is reused (grow-only)
    assert rows.shape[0] == xs.shape[0] and torch.isfinite(rows).all()
    rows2 = step.step(x.to(dev), y.to(dev))
    assert rows2.shape[0] == 12


def test_step_reads_momentum_and_weight_decay_from_param_groups(be, dev):
    """The reference's Trainer rewrites optimizer.param_groups[i]['momentum'] after the warm-up (warmup_momentum 0.8 -> 0.937, engine/vision_engine.py:169-171,350-352);
    a fused step that captured the constructor's value would silently ignore the switch."""
    ref, m1 = _pair(be, dev, seed=6)
    _, m2 = _pair(be, dev, seed=6)
    s1 = vit.FusedTrainStep(m1, lr=0.01, momentum=0.8, weight_decay=5e-4, ema=False)
    s2 = vit.FusedTrainStep(m2, lr=0.01, momentum=0.937, weight_decay=1e-3, ema=False)
    torch.manual_seed(12)
    x = torch.randn(4, 3, 32, 32).to(dev); y = torch.randint(0, 10, (4,)).to(dev)
    s1.step(x, y); s2.step(x, y)                       # first step: the momentum buffer is the gradient itself; weight decay already differs
    s1.param_groups[0]["momentum"] = 0.937             # what Trainer does once the warm-up is over
    s1.param_groups[0]["weight_decay"] = 1e-3
    s2.param_groups[0]["weight_decay"] = 1e-3
    m1.engine.params.copy_(m2.engine.params); s1.momentum_buf.copy_(s2.momentum_buf); m1.engine.refresh_weights()
    s1.step(x, y); s2.step(x, y)
    assert torch.equal(m1.engine.params, m2.engine.params)
    s1.param_groups[0]["momentum"] = 0.5
    s1.step(x, y); s2.step(x, y)
    assert not torch.equal(m1.engine.params, m2.engine.params)


def test_default_initialisation_follows_torch_manual_seed(be, dev):
    torch.manual_seed(123); a = vit.VisionTransformer(SPEC, device=dev, backend=be)
    torch.manual_seed(123); b = vit.VisionTransformer(SPEC, device=dev, backend=be)
    torch.manual_seed(124); c = vit.VisionTransformer(SPEC, device=dev, backend=be)
    assert torch.equal(a.engine.params, b.engine.params) and not torch.equal(a.engine.params, c.engine.params)


def test_qkv_bias_gradient_from_the_attention_backward(be, dev, monkeypatch):
    """qkv.bias = column sums of dqkv.  The one-pass attention backward delivers per-image partials of them with its stores (vdk_attention_bwd_cs; round 5) and the engine
    reduces those over the batch.  VDK_ATTN_GRID=3 makes every workgroup walk several (image, head) items, so the hand-over of the partials behind the NEXT item's first
    barrier -- and behind the loop for the last one -- is what runs; the gradient must equal the oracle's like every other one, for 2 heads x 7 images = 14 items."""
    monkeypatch.setenv("VDK_ATTN_GRID", "3")
    ref, model = _pair(be, dev)
    torch.manual_seed(11)
    x = torch.randn(7, 3, 32, 32); y = torch.randint(0, 10, (7,))
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    torch.nn.functional.cross_entropy(model(x.to(dev)), y.to(dev)).backward()
    seen = 0
    for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
        if n.endswith("attn.qkv.bias"):
            assert _rel(p.grad, pr.grad) < 3e-2, (n, _rel(p.grad, pr.grad))
            seen += 1
    assert seen == SPEC.depth


# ---- timm pre_norm=True: the CLIP ViTs (`vit_*_clip_*`; cbir.yaml lists vit_base_patch16_clip_224.laion2b_ft_in1k) ----------------------------------------------------------
PRE_SPEC = vit.VitSpec(img_size=32, patch_size=8, in_chans=3, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, ln_eps=1e-5, pre_norm=True)


def _pre_pair(be, dev, seed=0, operand="bf16"):
    torch.manual_seed(seed)
    ref = VisionTransformerRef(PRE_SPEC.img_size, PRE_SPEC.patch_size, 3, PRE_SPEC.num_classes, PRE_SPEC.dim, PRE_SPEC.depth, PRE_SPEC.heads, PRE_SPEC.mlp_dim, eps=1e-5, pre_norm=True)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.02)
        ref.patch_embed.proj.weight.mul_(8.0); ref.pos_embed.mul_(8.0)      # an embedding of O(1): norm_pre then has something to normalise
    model = vit.VisionTransformer(PRE_SPEC, device=dev, backend=be, seed=1, operand=operand)
    model.load_state_dict(ref.state_dict())
    return ref, model


def test_pre_norm_state_dict_names_follow_timm(be, dev):
    ref, model = _pre_pair(be, dev)
    names = [n for n, _ in model.named_parameters()]
    assert names == [n for n, _ in ref.named_parameters()]                 # cls_token, pos_embed, patch_embed.proj.weight, norm_pre.{weight, bias}, blocks..., norm, head
    assert "norm_pre.weight" in names and "patch_embed.proj.bias" not in names
    fresh = vit.VisionTransformer(PRE_SPEC, device=dev, backend=be, seed=4)
    sd = fresh.state_dict()
    assert torch.equal(sd["norm_pre.weight"].cpu(), torch.ones(PRE_SPEC.dim)) and torch.equal(sd["norm_pre.bias"].cpu(), torch.zeros(PRE_SPEC.dim))


@pytest.mark.parametrize("operand", ["bf16", "fp16"])
def test_pre_norm_forward_backward_vs_oracle(be, dev, operand):
    ref, model = _pre_pair(be, dev, operand=operand)
    torch.manual_seed(5)
    x = torch.randn(3, 3, 32, 32); y = torch.randint(0, 10, (3,))
    S = 256.0 if operand == "fp16" else 1.0
    logits_ref = ref(x)
    torch.nn.functional.cross_entropy(logits_ref, y, label_smoothing=0.05).backward()
    logits = model(x.to(dev))
    (torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=0.05) * S).backward()
    tol_l, tol_g = (2e-2, 6e-2) if operand == "bf16" else (3e-3, 1e-2)
    assert _rel(logits, logits_ref) < tol_l, _rel(logits, logits_ref)
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr and p.grad is not None, n
        assert _rel(p.grad / S, pr.grad) < tol_g, (n, _rel(p.grad / S, pr.grad))
    # the unreported bias slot of the patch embedding holds zeros and its gradient is zero: the optimizer leaves it alone
    eng = model.engine
    assert float(eng.params.abs().sum()) > 0
    listed = torch.zeros(eng.n_floats, dtype=torch.bool)
    for _, off, numel, _ in eng.entries:
        listed[off:off + numel] = True
    assert float(eng.params.cpu()[~listed].abs().max()) == 0.0 and float(eng.grads.cpu()[~listed].abs().max()) == 0.0


def test_pre_norm_precise_forward_vs_oracle(be, dev):
    ref, model = _pre_pair(be, dev)
    torch.manual_seed(6)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        want = ref(x)
        got = model.forward_precise(x.to(dev))
    assert _rel(got, want) < 1e-4, _rel(got, want)


def test_pre_norm_fused_step_vs_reference_step(be, dev):
    ref, model = _pre_pair(be, dev, seed=3)
    hyp = dict(lr=0.01, momentum=0.937, weight_decay=5e-4)
    step = vit.FusedTrainStep(model, label_smoothing=0.05, max_norm=10.0, ema=False, **hyp)
    init_sd = {n: p.detach().clone() for n, p in ref.named_parameters()}
    bufs = None
    torch.manual_seed(11)
    for it in range(2):
        x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
        _, loss_ref, _, _, bufs = train_step_reference(ref, x, y, label_smoothing=0.05, max_norm=10.0, momentum_bufs=bufs, ema=None, updates=it, **hyp)
        step.step(x.to(dev), y.to(dev))
        assert abs(step.loss_value() - loss_ref.item()) < 1e-2 * abs(loss_ref.item())
    sd = model.state_dict()
    for n, p in ref.named_parameters():
        assert _rel(sd[n].cpu() - init_sd[n], p.detach() - init_sd[n]) < 8e-2, n


def test_clip_ids_build(be, dev):
    for name in ("vit_base_patch32_clip_224", "vit_base_patch16_clip_224", "vit_large_patch14_clip_224", "vit_large_patch14_clip_336"):
        spec = vit.spec_from_timm_name(name, 7)
        assert spec.pre_norm and spec.ln_eps == 1e-5
