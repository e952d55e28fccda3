"""Forward + backward parity at FULL size for the two CNN configurations of BASELINE.json, every parameter gradient on its own:
  * configs[2]: ConvNeXt-Base (224 px, depths 3-3-27-3) + the BatchNorm2d/Linear/BatchNorm1d neck (512) + ArcFace over 100 000 identities, the FUSED head form the
    training step runs (margin + CE + gradient in one pass over cos), against oracle/convnext_ref.TimmWrapperCNNRef + arcface.py:20-36 restated, fp32 on the CPU;
  * configs[0]: ResNet-18 at 224 px, train-mode BatchNorm, BCE, batch 16, against oracle/resnet_ref.ResNetRef.
The engines multiply in bf16 (operands rounded, fp32 accumulation, fp32 master weights): against an fp32 oracle every tensor carries the operand-rounding noise of
its own GEMM chain (2^-9 per operand element, averaged over the contraction, compounding with depth), the same noise the reference's own autocast path has
against ITS fp32 path; tests/test_parity_bf16.py measures that floor for the ViT and shows the engine sits on it.  Bounds below are Frobenius-relative and were
set at 1.5x the values measured on the MI355X (printed by the test; fixed seeds, and the GEMM kernels of all three structures are bit-equal on the same
operands, so the values do not move with the kernel choice): a dropped term, a wrong scale or a mis-indexed tile is O(1).
Measured (round 2): ConvNeXt-B + ArcFace 100k, batch 8: embeddings 5.8e-3, loss 9.2e-5, dfeats 1.9e-3, dW 6.1e-3 (worst sampled class column 3.5e-2), worst backbone
gradient 4.2e-2 (stages.3.blocks.2.mlp.fc2.bias).  ResNet-18, batch 16: loss 4.7e-5; gradients: floor (oracle fp32- vs fp64-accumulate, same bf16 storage points)
median 0.184 / worst 0.247, engine 0.184 / 0.227 from the fp32 one and 0.185 / 0.242 from the fp64 one -- a train-mode-BatchNorm + ReLU network at random init is
that ill-conditioned in bf16 storage (fc.weight, in front of the chain: 6.7e-3)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _arcface_ref(f, w, y, m=0.35, s=32.0):          # models/faceX/head/arcface.py:20-36 (margin_am = 0)
    kn = torch.nn.functional.normalize(w, dim=0); f = torch.nn.functional.normalize(f)
    c = (f @ kn).clamp(-1, 1)
    cm = c * math.cos(m) - torch.sqrt(1 - c ** 2) * math.sin(m)
    cm = torch.where(c > math.cos(math.pi - m), cm, c)
    idx = torch.zeros_like(c).scatter_(1, y.view(-1, 1), 1).bool()
    return torch.where(idx, cm, c) * s


def test_convnext_base_neck_arcface_100k_forward_backward_vs_oracle(hip):
    from oracle.convnext_ref import TimmWrapperCNNRef
    from visiondk_amd import face
    C, B = 100_000, 8
    cfg = {"task": "cbir", "image_size": 224, "backbone": {"timm-convnext_base": {"pretrained": False, "image_size": 224, "feat_dim": 512, "operand": "bf16"}},
           "head": {"arcface": {"feat_dim": 512, "num_class": C, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    torch.manual_seed(0)
    model = face.get_model(cfg, None, 0, backend=hip, device="cuda:0").model.train()
    ref = TimmWrapperCNNRef(512, 224).train()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.1)                       # timm's 1e-6 layer-scale init would switch every block branch off
    bb, head = model.trainingwrapper["backbone"], model.trainingwrapper["head"]
    bb.load_state_dict({k: v.cuda() for k, v in ref.state_dict().items()}, strict=True)
    W = head.weight.detach().cpu().clone().requires_grad_(True)
    x = torch.randn(B, 3, 224, 224); y = torch.randint(0, C, (B,))
    emb_ref = ref(x); emb_ref.retain_grad()
    loss_ref = torch.nn.functional.cross_entropy(_arcface_ref(emb_ref, W, y), y)
    loss_ref.backward()
    emb = bb(x.cuda())
    loss_rows, df, dW = head.margin_ce(emb.detach(), y.cuda())
    emb.backward(df)
    res = {"emb": _rel(emb.detach(), emb_ref.detach()), "loss": abs(loss_rows.mean().item() - loss_ref.item()) / abs(loss_ref.item()),
           "dfeats": _rel(df, emb_ref.grad), "dW": _rel(dW, W.grad)}
    cols = torch.cat([y, torch.randint(0, C, (1000,))])
    res["dW_sampled_cols"] = max(_rel(dW[:, c], W.grad[:, c]) for c in cols[:64].tolist())
    got = dict(bb.named_parameters()); exp = dict(ref.named_parameters())
    gmax = max(p.grad.norm().item() for p in exp.values())
    worst = (0.0, None)
    for n, p in exp.items():
        if p.grad.norm().item() < 1e-5 * gmax:
            assert got[n].grad.norm().item() < 1e-3 * gmax, n       # analytically zero (bias in front of a train-mode BatchNorm)
            continue
        r = _rel(got[n].grad, p.grad)
        worst = max(worst, (r, n))
    res["worst_backbone_grad"] = worst
    print(res)
    # 1.5x the measured 5.8e-3 / 9.2e-5 / 1.9e-3 / 6.1e-3 / 3.5e-2 and 4.2e-2
    assert res["emb"] < 8.7e-3 and res["loss"] < 1.4e-4 and res["dfeats"] < 2.9e-3 and res["dW"] < 9.2e-3 and res["dW_sampled_cols"] < 5.3e-2, res
    assert worst[0] < 6.3e-2, worst


def test_convnext_base_neck_arcface_1m_fp16_operands_meet_the_stated_tolerance(hip):
    """BASELINE.json configs[2] at its configured size -- ConvNeXt-B + neck(512) + ArcFace over C = 1 000 000 identities (configs/faceX/cbir.yaml:30-35) -- in the mode
    tools/bench_cfg3.py times: fp16 operands in backbone, neck and head (cosines from single fp16 operands), gradients under a loss scale as FaceTrainStep's GradScaler
    protocol carries them.  The reference runs this loop in fp32 (engine/procedure/train.py:217-227); north_star's tolerance against it is asserted LITERALLY:
    embeddings <= 1e-3, every gradient (head weight, d(loss)/d(embedding), every backbone / neck parameter) <= 5e-3, loss <= 1e-3."""
    from oracle.convnext_ref import TimmWrapperCNNRef
    from visiondk_amd import face
    C, B, S = 1_000_000, 8, 1024.0          # S: a loss scale (65 536 at the bench's batch 512 is the same per-sample magnitude as 1024 at batch 8)
    cfg = {"task": "cbir", "image_size": 224, "backbone": {"timm-convnext_base": {"pretrained": False, "image_size": 224, "feat_dim": 512, "operand": "fp16"}},
           "head": {"arcface": {"feat_dim": 512, "num_class": C, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    torch.manual_seed(0)
    model = face.get_model(cfg, None, 0, backend=hip, device="cuda:0").model.train()
    ref = TimmWrapperCNNRef(512, 224).train()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.1)                       # timm's 1e-6 layer-scale init would switch every block branch off (tests/test_convnext.py covers gamma = 1e-6)
    bb, head = model.trainingwrapper["backbone"], model.trainingwrapper["head"]
    assert bb.model.engine.operand == "fp16"
    bb.load_state_dict({k: v.cuda() for k, v in ref.state_dict().items()}, strict=True)
    W = head.weight.detach().cpu().clone().requires_grad_(True)
    x = torch.randn(B, 3, 224, 224); y = torch.randint(0, C, (B,))
    emb_ref = ref(x); emb_ref.retain_grad()
    loss_ref = torch.nn.functional.cross_entropy(_arcface_ref(emb_ref, W, y), y)
    loss_ref.backward()
    emb = bb(x.cuda())
    ls = torch.tensor([S, 0.0, 0.0], device="cuda:0")
    loss_rows, df, dW = head.margin_ce(emb.detach(), y.cuda(), cos_planes=1, operand="fp16", loss_scale=ls)
    emb.backward(df)
    res = {"emb": _rel(emb.detach(), emb_ref.detach()), "loss": abs(loss_rows.mean().item() - loss_ref.item()) / abs(loss_ref.item()),
           "dfeats": _rel(df / S, emb_ref.grad), "dW": _rel(dW / S, W.grad)}
    cols = torch.cat([y, torch.randint(0, C, (1000,))])
    res["dW_sampled_cols"] = max(_rel(dW[:, c] / S, W.grad[:, c]) for c in cols[:64].tolist())
    got = dict(bb.named_parameters()); exp = dict(ref.named_parameters())
    gmax = max(p.grad.norm().item() for p in exp.values())
    rr = []
    for n, p in exp.items():
        assert torch.isfinite(got[n].grad).all(), n
        if p.grad.norm().item() < 1e-5 * gmax:
            assert got[n].grad.norm().item() / S < 1e-3 * gmax, n       # analytically zero (bias in front of a train-mode BatchNorm)
            continue
        rr.append((_rel(got[n].grad / S, p.grad), n))
    rr.sort(reverse=True)
    worst = rr[0]
    res["worst_backbone_grad"] = worst
    res["next_worst"] = rr[1:6]
    res["median_backbone_grad"] = rr[len(rr) // 2][0]
    print(res)
    assert res["emb"] <= 1e-3 and res["loss"] <= 1e-3, res
    assert res["dfeats"] <= 5e-3 and res["dW"] <= 5e-3 and worst[0] <= 5e-3, res


def test_convnext_base_classifier_fp16_operands_meet_the_stated_tolerance(hip):
    """pet.yaml:21-22 `timm-convnext_base` as a classifier (37 classes; head = global average pool -> head.norm -> head.fc) with fp16 operands -- the reference's autocast
    dtype on a GPU (engine/procedure/train.py:118) -- forward + CE backward under a loss scale, every parameter gradient against the fp32 oracle: logits <= 1e-3,
    gradients <= 5e-3, asserted literally (bf16 operands on the same model: 4e-3 / 2.7e-2)."""
    from oracle.convnext_ref import ConvNeXtRef
    from visiondk_amd import convnext
    B, S, ncls = 8, 1024.0, 37
    torch.manual_seed(0)
    model = convnext.create_model("convnext_base", num_classes=ncls, device="cuda:0", backend=hip, operand="fp16").train()
    ref = ConvNeXtRef(num_classes=ncls).train()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.1)
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(B, 3, 224, 224); y = torch.randint(0, ncls, (B,))
    lr = ref(x); loss_ref = torch.nn.functional.cross_entropy(lr, y); loss_ref.backward()
    lo = model(x.cuda()); loss = torch.nn.functional.cross_entropy(lo, y.cuda())
    (loss * S).backward()                                   # scaler.scale(loss).backward() (train.py:205)
    got, exp = dict(model.named_parameters()), dict(ref.named_parameters())
    rr = sorted(((_rel(got[n].grad / S, p.grad), n) for n, p in exp.items()), reverse=True)
    res = {"logits": _rel(lo.detach(), lr.detach()), "loss": abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), "worst_grad": rr[0], "next_worst": rr[1:4],
           "median_grad": rr[len(rr) // 2][0]}
    print(res)
    assert res["logits"] <= 1e-3 and res["loss"] <= 1e-3, res
    assert rr[0][0] <= 5e-3, res


def test_convnext_base_neck_arcface_100k_fp32_precision_vs_oracle(hip):
    """cfg3's model in the fp32-class arithmetic mode (engine.precision = "fp32", head precise=True): the reference's face / CBIR loop has no autocast
    (engine/procedure/train.py:217-227).  north_star's bar for this path -- embeddings <= 1e-3, gradients <= 5e-3 of the fp32 oracle -- with two orders of margin."""
    from oracle.convnext_ref import TimmWrapperCNNRef
    from visiondk_amd import face
    C, B = 100_000, 8
    cfg = {"task": "cbir", "image_size": 224, "backbone": {"timm-convnext_base": {"pretrained": False, "image_size": 224, "feat_dim": 512, "operand": "bf16"}},
           "head": {"arcface": {"feat_dim": 512, "num_class": C, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    torch.manual_seed(0)
    model = face.get_model(cfg, None, 0, backend=hip, device="cuda:0").model.train()
    ref = TimmWrapperCNNRef(512, 224).train()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.1)
    bb, head = model.trainingwrapper["backbone"], model.trainingwrapper["head"]
    bb.load_state_dict({k: v.cuda() for k, v in ref.state_dict().items()}, strict=True)
    bb.precision = "fp32"; bb.model.engine.precision = "fp32"
    W = head.weight.detach().cpu().clone().requires_grad_(True)
    x = torch.randn(B, 3, 224, 224); y = torch.randint(0, C, (B,))
    emb_ref = ref(x); emb_ref.retain_grad()
    loss_ref = torch.nn.functional.cross_entropy(_arcface_ref(emb_ref, W, y), y)
    loss_ref.backward()
    emb = bb(x.cuda())
    loss_rows, df, dW = head.margin_ce(emb.detach(), y.cuda(), precise=True)
    emb.backward(df)
    res = {"emb": _rel(emb.detach(), emb_ref.detach()), "loss": abs(loss_rows.mean().item() - loss_ref.item()) / abs(loss_ref.item()),
           "dfeats": _rel(df, emb_ref.grad), "dW": _rel(dW, W.grad)}
    got = dict(bb.named_parameters()); exp = dict(ref.named_parameters())
    gmax = max(p.grad.norm().item() for p in exp.values())
    worst = (0.0, None)
    for n, p in exp.items():
        if p.grad.norm().item() < 1e-5 * gmax:
            assert got[n].grad.norm().item() < 1e-3 * gmax, n
            continue
        worst = max(worst, (_rel(got[n].grad, p.grad), n))
    res["worst_backbone_grad"] = worst
    print(res)
    # measured on the MI355X: embeddings 4.1e-6, loss equal to the last digit, dfeats 1.0e-6, dW 4.1e-6, worst backbone gradient 1.2e-5 (stages.3.downsample.0.bias)
    assert res["emb"] < 2e-5 and res["loss"] < 2e-6 and res["dfeats"] < 1e-5 and res["dW"] < 2e-5, res
    assert worst[0] < 6e-5, worst


def test_resnet18_every_gradient_vs_oracle(hip):
    """ReLU + train-mode BatchNorm make a plain fp32 run differ from ANY run that stores activations in bf16 by tens of percent in the early-layer gradients (a
    pre-activation that rounds across zero flips its mask, BatchNorm's backward subtracts two nearly equal sums; torch's own CPU autocast shows 25-40 %, the
    engine 31 % median here), so the oracle puts the bf16 storage points where the engine has them (oracle/resnet_ref.forward_bf16_storage) and the floor is
    measured like the ViT's: that oracle accumulating in fp32 against itself accumulating in fp64."""
    import copy
    from oracle.resnet_ref import ResNetRef, forward_bf16_storage
    from visiondk_amd import resnet
    torch.manual_seed(1)
    model = resnet.create_model("resnet18", num_classes=5, device="cuda:0", backend=hip).train()
    ref = ResNetRef(5).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, (2.0 / m.weight[0].numel()) ** 0.5)
                m.weight.copy_(m.weight.bfloat16().float())           # bf16-representable weights: the engine's operand copies are bf16
    model.load_state_dict(ref.state_dict(), strict=True)
    ref64 = copy.deepcopy(ref).double()
    x = torch.randn(16, 3, 224, 224); t = (torch.rand(16, 5) > 0.5).float()
    l32 = torch.nn.functional.binary_cross_entropy_with_logits(forward_bf16_storage(ref, x), t); l32.backward()
    l64 = torch.nn.functional.binary_cross_entropy_with_logits(forward_bf16_storage(ref64, x.double()), t.double()); l64.backward()
    lo = torch.nn.functional.binary_cross_entropy_with_logits(model(x.cuda()), t.cuda()); lo.backward()
    got = dict(model.named_parameters()); g64 = dict(ref64.named_parameters())
    e32 = {n: _rel(got[n].grad, p.grad) for n, p in ref.named_parameters()}
    e64 = {n: _rel(got[n].grad, g64[n].grad) for n in e32}
    fl = {n: _rel(p.grad, g64[n].grad) for n, p in ref.named_parameters()}
    w32, w64, wfl = (max(d.items(), key=lambda kv: kv[1]) for d in (e32, e64, fl))
    med = lambda d: sorted(d.values())[len(d) // 2]
    print({"loss_vs_o32": abs(lo.item() - l32.item()) / abs(l32.item()), "worst_vs_o32": w32, "worst_vs_o64": w64, "floor_worst": wfl,
           "median_vs_o32": med(e32), "median_vs_o64": med(e64), "floor_median": med(fl)})
    assert abs(lo.item() - l32.item()) < 2e-3 * abs(l32.item())
    assert w32[1] <= 1.5 * wfl[1] + 1e-3 and w64[1] <= 1.5 * wfl[1] + 1e-3
    assert med(e32) <= 1.5 * med(fl) + 1e-3


def test_siglip_vit_large_336_forward_backward_vs_oracle(hip):
    """BASELINE.json configs[4]'s model at full size -- vit_large_patch14_siglip_336: 576 patch tokens (the streaming attention kernels), 24 blocks of width 1024, the
    patch-14 stem through zero-padded operand copies, AttentionPoolLatent head -- forward + backward for 2 images against oracle/vit_ref.SiglipVisionTransformerRef in
    fp32: logits, loss and EVERY parameter gradient.  Bounds = the measured values with ~1.5x margin (bf16 operands over 24 blocks: the same level as ViT-B/16 in
    test_parity_bf16_fullsize, printed for the record); the fp8 mode of the same model is held to the looser fp8 bounds of tests/test_vit_fp8.py."""
    from oracle.vit_ref import SiglipVisionTransformerRef
    from visiondk_amd import vit
    torch.manual_seed(0)
    ref = SiglipVisionTransformerRef(336, 14, 3, 1000, 1024, 24, 16, 4096)
    model = vit.create_model("vit_large_patch14_siglip_336", num_classes=1000, device="cuda:0", backend=hip)
    model.load_state_dict({k: v.cuda() for k, v in ref.state_dict().items()}, strict=True)
    x = torch.randn(2, 3, 336, 336); y = torch.randint(0, 1000, (2,))
    lr = ref(x); loss_r = torch.nn.functional.cross_entropy(lr, y); loss_r.backward()
    lo = model(x.cuda()); loss = torch.nn.functional.cross_entropy(lo, y.cuda()); loss.backward()
    got = dict(model.named_parameters())
    errs = sorted(((_rel(got[n].grad, p.grad), n) for n, p in ref.named_parameters()), reverse=True)
    print({"logits_rel": _rel(lo.detach(), lr.detach()), "loss": (loss.item(), loss_r.item()), "worst_grads": errs[:4], "median_grad": errs[len(errs) // 2]})
    # measured on the MI355X: logits 4.1e-3, loss 6.89800 vs 6.89972 (2.5e-4; seeded pooling head since round 3), worst gradient 8.0e-3 (blocks.0.norm2.weight), median 5.0e-3
    assert _rel(lo.detach(), lr.detach()) < 6.3e-3 and abs(loss.item() - loss_r.item()) < 3.8e-4 * abs(loss_r.item())      # 1.5x measured
    assert errs[0][0] < 1.2e-2, errs[:4]
    assert errs[len(errs) // 2][0] < 7.5e-3
    # the same model with fp8 operands in the block Linears (current scaling = the calibration step, then delayed scaling from the recorded maxima)
    def run():
        for q in model.parameters():
            q.grad = None
        l8 = model(x.cuda()); torch.nn.functional.cross_entropy(l8, y.cuda()).backward()
        e8 = sorted(((_rel(got[n].grad, p.grad), n) for n, p in ref.named_parameters()), reverse=True)
        return _rel(l8.detach(), lr.detach()), e8
    model.engine.enable_fp8(2); lc, ec = run(); model.engine.fp8_update()
    model.engine.enable_fp8(1); ld, ed = run()
    model.engine.enable_fp8(0)
    print({"fp8_current": (lc, ec[0], ec[len(ec) // 2]), "fp8_delayed": (ld, ed[0], ed[len(ed) // 2])})
    # measured: logits 6.8e-2, worst gradient 1.33e-1 (pos_embed), median 7.6e-2 -- e4m3 / e5m2 operands (2^-4 / 2^-3 relative rounding) through 24 blocks; the delayed-scaling
    # pass (fp8 copies written by the producing kernels) reproduces the calibration pass exactly on the same data
    assert lc < 1e-1 and ld < 1e-1 and ec[0][0] < 2e-1 and ed[0][0] < 2e-1 and ec[len(ec) // 2][0] < 1.14e-1      # 1.5x measured
    assert abs(lc - ld) < 1e-6


@pytest.mark.parametrize("planes", [1, 3])
def test_arcface_head_at_one_million_identities_vs_oracle(hip, planes):
    """cfg3's head at its real width -- ArcFace(feat 512, C = 10^6, m = .35, s = 32) fused with CrossEntropy, 128 rows -- against the reference's arcface.py arithmetic
    restated in fp32 on the CPU (_arcface_ref + torch autograd): loss, d(feats), and dW over all 10^6 columns.  planes = 1: cosines from single bf16 operands (what the
    reference's autocast computes; the default of the cfg3 step), 3: split-bf16 planes (fp32-class cosines)."""
    from visiondk_amd import heads
    C, B, D = 1_000_000, 128, 512
    torch.manual_seed(0)
    head = heads.ArcFace(D, C, margin_arc=0.35, margin_am=0.0, scale=32, backend=hip, device="cuda:0")
    W = head.weight.detach().cpu().clone().requires_grad_(True)
    f = torch.randn(B, D).requires_grad_(True); y = torch.randint(0, C, (B,))
    loss_ref = torch.nn.functional.cross_entropy(_arcface_ref(f, W, y), y)
    loss_ref.backward()
    loss_rows, df, dW = head.margin_ce(f.detach().cuda(), y.cuda(), cos_planes=planes)
    res = {"loss": abs(loss_rows.mean().item() - loss_ref.item()) / abs(loss_ref.item()), "dfeats": _rel(df, f.grad), "dW": _rel(dW, W.grad),
           "dW_target_cols": max(_rel(dW[:, c], W.grad[:, c]) for c in y[:32].tolist())}
    print(planes, res)
    # measured: loss 4.9e-6 / 7.4e-8 (1 / 3 planes), dfeats 2.1e-3, dW 2.0e-3, target columns 2.8e-3 (the backward GEMMs take bf16 operands in both modes)
    tol = {1: (7.4e-6, 3.2e-3, 3.0e-3, 4.2e-3), 3: (2e-7, 3.2e-3, 3.0e-3, 4.2e-3)}[planes]      # 1.5x measured (3 planes: the loss is at fp32 summation noise, 2e-7)
    assert res["loss"] < tol[0] and res["dfeats"] < tol[1] and res["dW"] < tol[2] and res["dW_target_cols"] < tol[3], res
