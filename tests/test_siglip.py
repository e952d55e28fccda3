"""timm class_token=False + global_pool='map' (the vit_*_siglip_* family, BASELINE.json configs[4]): the engine without a class token + AttentionPoolLatent + head
against oracle/vit_ref.SiglipVisionTransformerRef (itself pinned against transformers.SiglipVisionModel, tests/test_oracle_vit.py): logits and EVERY gradient."""
import pytest
import torch

from oracle.vit_ref import SiglipVisionTransformerRef
from visiondk_amd import vit


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _pair(be, dev, img=32, patch=8, dim=128, depth=2, heads=2, mlp=256, classes=10, operand="bf16"):
    torch.manual_seed(0)
    ref = SiglipVisionTransformerRef(img, patch, 3, classes, dim, depth, heads, mlp)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        for blk in ref.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                lin.weight.mul_(3.0)
        for lin in (ref.attn_pool.q, ref.attn_pool.kv, ref.attn_pool.proj, ref.attn_pool.mlp.fc1, ref.attn_pool.mlp.fc2):
            lin.weight.mul_(4.0)
    model = vit.VisionTransformerMap(vit.VitSpec(img_size=img, patch_size=patch, num_classes=classes, dim=dim, depth=depth, heads=heads, mlp_dim=mlp, class_token=False),
                                     device=dev, backend=be, seed=1, operand=operand)
    model.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()}, strict=True)
    return ref, model


def test_state_dict_names_are_timms(be, dev):
    ref, model = _pair(be, dev)
    keys = list(model.state_dict().keys())
    assert keys == list(ref.state_dict().keys())
    assert keys[0] == "pos_embed" and "cls_token" not in keys and "attn_pool.latent" in keys and "attn_pool.kv.weight" in keys and keys[-1] == "head.bias"
    assert model.state_dict()["pos_embed"].shape == (1, 16, 128)


@pytest.mark.parametrize("img,B", [(32, 4), (64, 2)])
def test_forward_backward_vs_oracle(be, dev, img, B):
    ref, model = _pair(be, dev, img=img)
    torch.manual_seed(3)
    x = torch.randn(B, 3, img, img); y = torch.randint(0, 10, (B,))
    lr = ref(x); loss_r = torch.nn.functional.cross_entropy(lr, y); loss_r.backward()
    lo = model(x.to(dev)); loss = torch.nn.functional.cross_entropy(lo, y.to(dev)); loss.backward()
    assert _rel(lo.detach(), lr.detach()) < 2e-2 and abs(loss.item() - loss_r.item()) < 1e-2 * abs(loss_r.item())
    got = dict(model.named_parameters())
    worst = max(((_rel(got[n].grad, p.grad), n) for n, p in ref.named_parameters()))
    print(worst)
    for n, p in ref.named_parameters():
        assert got[n].grad is not None, n
        assert _rel(got[n].grad, p.grad) < 6e-2, (n, _rel(got[n].grad, p.grad))


def test_attention_pool_alone_vs_oracle(be, dev):
    """the pooling head on given tokens: fp32 tail, bf16 only in the kv Linear -> 1e-2 on dtokens, 3e-3 on the output"""
    from oracle.vit_ref import AttentionPoolLatentRef
    torch.manual_seed(5)
    D, H, N, B = 128, 2, 37, 3
    ref = AttentionPoolLatentRef(D, H, 256)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    pool = vit.AttentionPoolLatent(D, H, 256, backend=be, device=dev)
    pool.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()}, strict=True)
    t = torch.randn(B, N, D)
    tr = t.clone().requires_grad_(True); tg = t.clone().to(dev).requires_grad_(True)
    o_r = ref(tr); o_g = pool(tg)
    w = torch.randn(B, D)
    (o_r * w).sum().backward(); (o_g * w.to(dev)).sum().backward()
    assert _rel(o_g.detach(), o_r.detach()) < 3e-3
    assert _rel(tg.grad, tr.grad) < 1e-2
    got = dict(pool.named_parameters())
    for n, p in ref.named_parameters():
        assert _rel(got[n].grad, p.grad) < 1e-2, (n, _rel(got[n].grad, p.grad))


def test_create_model_routes_siglip_ids(be, dev):
    m = vit.create_model("vit_base_patch16_siglip_224", num_classes=7, device=dev, backend=be, img_size=32)
    assert isinstance(m, vit.VisionTransformerMap) and m.engine.tokens == 4 and not m.spec.class_token


@pytest.mark.parametrize("sam", [False, True])
def test_train_step_matches_reference_update(be, dev, sam):
    """MapTrainStep (clip + SGD + EMA, or the SAM sequence) against oracle/vit_ref.train_step_reference[_sam] on the same weights: parameter UPDATES of every tensor"""
    from oracle.vit_ref import train_step_reference, train_step_reference_sam
    ref, model = _pair(be, dev)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    step = vit.MapTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1, max_norm=10.0, ema=True, sam=sam)
    if sam:
        loss_r, _ = train_step_reference_sam(ref, x, y, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1)
    else:
        _, loss_r, _, _, _ = train_step_reference(ref, x, y, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1, max_norm=10.0)
    step.step(x.to(dev), y.to(dev))
    assert abs(step.loss_value() - loss_r.item()) < 1e-2 * abs(loss_r.item())
    got = dict(model.named_parameters())
    for n, p in ref.named_parameters():
        upd_r = p.detach() - before[n]; upd = got[n].detach().cpu() - before[n]
        assert _rel(upd, upd_r) < 8e-2, (n, _rel(upd, upd_r))
    # the model still evaluates with the updated weights (bf16 operand copies refreshed)
    with torch.no_grad():
        assert _rel(model(x.to(dev)), ref(x)) < 2e-2


@pytest.mark.parametrize("img,B", [(32, 4), (64, 2)])
def test_forward_backward_fp16_operands_vs_oracle(be, dev, img, B):
    """the same model on fp16 operands (the reference's autocast dtype, train.py:118) with a scaled backward (GradScaler, train.py:205-208): trunk, kv Linear of the
    pooling head, kv / dkv of the one-query attention all in IEEE half -- 8x closer to the fp32 oracle than the bf16 arm above (whose bounds are 2e-2 / 6e-2)"""
    ref, model = _pair(be, dev, img=img, operand="fp16")
    torch.manual_seed(3)
    x = torch.randn(B, 3, img, img); y = torch.randint(0, 10, (B,))
    S = 1024.0
    lr = ref(x); torch.nn.functional.cross_entropy(lr, y).backward()
    lo = model(x.to(dev)); (torch.nn.functional.cross_entropy(lo, y.to(dev)) * S).backward()
    assert _rel(lo.detach(), lr.detach()) < 2.5e-3
    got = dict(model.named_parameters())
    worst = max(((_rel(got[n].grad / S, p.grad), n) for n, p in ref.named_parameters()))
    print(worst)
    assert worst[0] < 8e-3, worst


@pytest.mark.parametrize("sam", [False, True])
def test_train_step_fp16_operands_scaler_protocol(be, dev, sam):
    """MapTrainStep on fp16 operands: the GradScaler protocol around the clipped SGD step AND around the SAM sequence (both passes at the current scale, e(w) invariant
    under it, the base step un-scales) -- updates equal the fp32 reference's to fp16 rounding; an overflowing scale skips the step, restores w and halves the scale."""
    from oracle.vit_ref import train_step_reference, train_step_reference_sam
    ref, model = _pair(be, dev, operand="fp16")
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    step = vit.MapTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1, max_norm=10.0, ema=True, sam=sam, init_scale=1024.0)
    if sam:
        loss_r, _ = train_step_reference_sam(ref, x, y, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1)
    else:
        _, loss_r, _, _, _ = train_step_reference(ref, x, y, lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.1, max_norm=10.0)
    step.step(x.to(dev), y.to(dev))
    assert step.skipped_steps() == 0 and step.loss_scale() == 1024.0
    assert abs(step.loss_value() - loss_r.item()) < 2e-3 * abs(loss_r.item())
    got = dict(model.named_parameters())
    for n, p in ref.named_parameters():
        upd_r = p.detach() - before[n]; upd = got[n].detach().cpu() - before[n]
        assert _rel(upd, upd_r) < 2e-2, (n, _rel(upd, upd_r))
    # an absurd scale overflows the fp16 gradients: the step is skipped, the weights stay, the scale backs off
    kept = {n: p.detach().clone() for n, p in model.named_parameters()}
    step.loss_state[0] = 2.0 ** 40
    step.step(x.to(dev), y.to(dev))
    assert step.skipped_steps() == 1 and step.loss_scale() == 2.0 ** 39
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), kept[n]), n
