"""K11 margin heads vs the golden outputs of the REFERENCE's own ArcFace / CircleLoss / MV_Softmax modules
(tests/golden/heads.npz, made by tests/golden/make_golden.py).  The GEMMs use bf16 MFMA operands: cos errors of a few 1e-4
are amplified by the scale (32 / 256) in the logits, so logits are compared in units of the scale; gradients flow through
single-plane bf16 GEMMs (dcos, f^, W^ rounded to bf16): 2e-3 ... 3e-3 relative against the reference's modules (bound 8e-3)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from visiondk_amd import heads

G = Path(__file__).resolve().parent / "golden" / "heads.npz"


def _mk(tag, be, dev):
    if tag == "arcface":
        return heads.ArcFace(64, 257, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev), 32.0
    if tag == "circle":
        return heads.CircleLoss(64, 257, margin=0.25, gamma=256, backend=be, device=dev), 256.0
    if tag == "mv_am":
        return heads.MV_Softmax(64, 257, is_am=True, margin=0.35, mv_weight=1.12, scale=32, backend=be, device=dev), 32.0
    return heads.MV_Softmax(64, 257, is_am=False, margin=0.35, mv_weight=1.12, scale=32, backend=be, device=dev), 32.0


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tag", ["arcface", "circle", "mv_am", "mv_arc"])
def test_head_logits_loss_and_grads_vs_reference(be, dev, tag):
    z = np.load(G)
    h, scale = _mk(tag, be, dev)
    with torch.no_grad():
        h.weight.copy_(torch.from_numpy(z[f"{tag}_weight"]).to(dev))
    feats = torch.from_numpy(z["feats"]).to(dev).requires_grad_(True)
    labels = torch.from_numpy(z["labels"]).to(dev)
    logits = h(feats, labels)
    ref_logits = torch.from_numpy(z[f"{tag}_logits"])
    # cos is computed with split-bf16 planes (hi*hi + lo*hi + hi*lo): fp32-class accuracy even where the margin is steep
    assert (logits.detach().cpu() - ref_logits).abs().max().item() / scale < 2e-5
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    assert abs(loss.item() - float(z[f"{tag}_loss"])) < 1e-4 * abs(float(z[f"{tag}_loss"])) + 1e-4
    assert _rel(feats.grad, z[f"{tag}_dfeats"]) < 8e-3        # measured 2.0e-3 ... 2.8e-3 (dcos, f^ and W^ enter the backward GEMMs rounded to bf16 once)
    assert _rel(h.weight.grad, z[f"{tag}_dweight"]) < 8e-3
    # fused form == the autograd form
    loss_rows, df, dW = h.margin_ce(feats.detach(), labels)
    assert abs(loss_rows.mean().item() - loss.item()) < 1e-5 * abs(loss.item()) + 1e-6
    assert _rel(df, feats.grad) < 2e-2 and _rel(dW, h.weight.grad) < 2e-2


@pytest.mark.parametrize("tag", ["arcface", "circle", "mv_am", "mv_arc"])
def test_wide_head_fused_form_equals_autograd_form(be, dev, tag):
    """C >= 4096 takes the vectorised online-softmax kernel (16-byte loads, cos read twice); it must give what the logits-returning form followed by
    torch's CrossEntropy (label smoothing on) gives.  C is not a multiple of 4: the scalar tail and the zeroed padding columns are covered."""
    torch.manual_seed(3)
    D, Cn, B = 64, 5003, 6
    if tag == "arcface":
        h = heads.ArcFace(D, Cn, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev)
    elif tag == "circle":
        h = heads.CircleLoss(D, Cn, margin=0.25, gamma=64, backend=be, device=dev)
    else:
        h = heads.MV_Softmax(D, Cn, is_am=(tag == "mv_am"), margin=0.35, mv_weight=1.12, scale=32, backend=be, device=dev)
    feats = torch.randn(B, D, device=dev, requires_grad=True)
    labels = torch.tensor([0, 5002, 1234, 4096, 17, 4999], device=dev)
    loss = torch.nn.functional.cross_entropy(h(feats, labels), labels, label_smoothing=0.1)
    loss.backward()
    loss_rows, df, dW = h.margin_ce(feats.detach(), labels, label_smoothing=0.1)
    assert abs(loss_rows.mean().item() - loss.item()) < 2e-6 * abs(loss.item())
    assert _rel(df, feats.grad) < 2e-2 and _rel(dW, h.weight.grad) < 2e-2


def test_arcface_fast_path_equals_the_generic_evaluation(be, dev, monkeypatch):
    """wide ArcFace heads (cfg3: 10^6 identities) evaluate the margin once per row -- the target's logit and jacobian -- and every other entry as s * clamp(cos) with a
    select; VDK_MARGIN_GENERIC=1 runs the generic per-entry evaluation: the same expressions, so loss rows and both gradients are bit-identical.  Includes cosines pushed
    outside [-1, 1] (the clamp's zero jacobian) and a target beyond cos(pi - m) (the fallback branch)."""
    torch.manual_seed(7)
    D, Cn, B = 64, 5003, 6
    h = heads.ArcFace(D, Cn, margin_arc=0.35, margin_am=0.1, scale=32, backend=be, device=dev)
    feats = torch.randn(B, D, device=dev)
    with torch.no_grad():
        h.weight[:, 5] = -feats[1] * 3.0            # target of row 1 nearly opposite: cos < cos(pi - m)
    labels = torch.tensor([0, 5, 1234, 4096, 17, 4999], device=dev)
    outs = []
    for generic in ("0", "1"):
        monkeypatch.setenv("VDK_MARGIN_GENERIC", generic)
        outs.append(h.margin_ce(feats, labels, label_smoothing=0.1))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["arcface", "circle", "mv_am"])
def test_sharded_form_with_one_shard_equals_fused_form(be, dev, tag):
    """heads.sharded_margin_ce without a process group = one shard holding every class: its three local passes (target cosine, statistics, gradient) must
    reproduce the fused kernel; C = 5003 exercises the 16-byte path, its scalar tail and the zeroed padding columns."""
    torch.manual_seed(5)
    D, Cn, B = 64, 5003, 5
    if tag == "arcface":
        h = heads.ArcFace(D, Cn, backend=be, device=dev)
    elif tag == "circle":
        h = heads.CircleLoss(D, Cn, margin=0.25, gamma=64, backend=be, device=dev)
    else:
        h = heads.MV_Softmax(D, Cn, is_am=True, backend=be, device=dev)
    feats = torch.randn(B, D, device=dev)
    labels = torch.tensor([0, 5002, 77, 4096, 2500], device=dev)
    l1, df1, dW1 = h.margin_ce(feats, labels, label_smoothing=0.1)
    l2, df2, dW2 = heads.sharded_margin_ce(h, feats, labels, h.weight.detach(), 0, Cn, label_smoothing=0.1)
    assert _rel(l2, l1) < 1e-6 and _rel(df2, df1) < 1e-3 and _rel(dW2, dW1) < 1e-3


def test_magface_vs_reference_module(be, dev):
    """MagFace (models/faceX/head/magface.py:26-47): logits, the magnitude regulariser and the gradients of CE(logits) + mean(regulariser), including the path through
    the magnitude-dependent margin, against the reference module's own outputs (tests/golden/magface.npz)"""
    z = np.load(G.parent / "magface.npz")
    h = heads.HeadFactory("magface", {"feat_dim": 64, "num_class": 257}, backend=be, device=dev).get_head()
    with torch.no_grad():
        h.weight.copy_(torch.from_numpy(z["weight"]).to(dev))
    feats = torch.from_numpy(z["feats"]).to(dev).requires_grad_(True)
    labels = torch.from_numpy(z["labels"]).to(dev)
    logits, reg = h(feats, labels)
    assert (logits.detach().cpu() - torch.from_numpy(z["logits"])).abs().max().item() / 32.0 < 2e-5
    assert torch.allclose(reg.detach().cpu(), torch.from_numpy(z["reg"]), rtol=1e-6, atol=1e-7)
    loss = torch.nn.functional.cross_entropy(logits, labels) + reg.mean()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * abs(float(z["loss"]))
    assert _rel(feats.grad, z["dfeats"]) < 5e-2 and _rel(h.weight.grad, z["dweight"]) < 5e-2


def test_head_factory_names():
    f = heads.HeadFactory("arcface", {"feat_dim": 8, "num_class": 16}, backend=None, device="cpu")
    assert f.head_type == "arcface"


@pytest.mark.parametrize("D,C", [(64, 260), (512, 520), (192, 264)])
def test_tiled_column_normalisation_and_single_plane_cos(be, dev, D, C):
    """the register-tiled colnorm kernels (16-byte aligned rows, D <= 512) against torch, and the single-plane cos (cos_planes = 1: both operands rounded to bf16 once,
    what torch.mm computes under the reference's autocast, train.py:118) against exactly that arithmetic"""
    torch.manual_seed(3)
    h = heads.ArcFace(D, C, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev)
    B = 24
    feats = torch.randn(B, D); labels = torch.randint(0, C, (B,))
    w = h.weight.detach().cpu()
    st3 = heads._forward_cos(be, feats.to(dev), h.weight.detach(), 3)
    st1 = heads._forward_cos(be, feats.to(dev), h.weight.detach(), 1)
    wn = torch.nn.functional.normalize(w, dim=0); fn = torch.nn.functional.normalize(feats)
    assert _rel(st3.winv[:C], 1.0 / w.norm(dim=0)) < 1e-6 and _rel(st1.winv[:C], 1.0 / w.norm(dim=0)) < 1e-6
    assert torch.equal(st1.wb[:D, :C].cpu(), wn.bfloat16()) or _rel(st1.wb[:D, :C].float(), wn.bfloat16().float()) < 1e-3      # (1-ulp differences where x * (1/n) != x / n)
    assert float(st1.wb[:, C:].float().abs().sum()) == 0.0
    assert (st3.cos[:B, :C].cpu() - fn @ wn).abs().max().item() < 5e-6
    ref1 = st1.fb[:B].float().cpu() @ st1.wb[:D, :C].float().cpu()
    assert (st1.cos[:B, :C].cpu() - ref1).abs().max().item() < 5e-6
    assert (st1.cos[:B, :C].cpu() - fn @ wn).abs().max().item() < 4e-3
    # the fused step in both modes against fp32 autograd of the reference arithmetic; column-normalisation backward through the tiled kernel
    wr = w.clone().requires_grad_(True); fr = feats.clone().requires_grad_(True)
    kn = torch.nn.functional.normalize(wr, dim=0); f2 = torch.nn.functional.normalize(fr)
    c = (f2 @ kn).clamp(-1, 1)
    import math
    cm = torch.where(c > math.cos(math.pi - 0.35), c * math.cos(0.35) - torch.sqrt(1 - c ** 2) * math.sin(0.35), c)
    idx = torch.zeros_like(c).scatter_(1, labels.view(-1, 1), 1).bool()
    loss = torch.nn.functional.cross_entropy(torch.where(idx, cm, c) * 32, labels)
    loss.backward()
    for planes, tol in ((3, 1e-2), (1, 2e-2)):
        lr, df, dW = h.margin_ce(feats.to(dev), labels.to(dev), cos_planes=planes)
        assert abs(lr.mean().item() - loss.item()) < (1e-4 if planes == 3 else 5e-3) * abs(loss.item())
        assert _rel(df, fr.grad) < tol and _rel(dW, wr.grad) < tol


@pytest.mark.parametrize("tag,planes", [("arcface", 3), ("arcface", 1), ("circle", 3), ("mv_am", 3), ("mv_arc", 1)])
def test_epilogue_fused_head_equals_materialised_form(be, dev, tag, planes):
    """margin_ce(fused=True): the cos GEMM runs twice with the head in its epilogue (pass 1: per-slice softmax partials, pass 2: d cos from the row statistics), cos never
    written as fp32 -- against the form that materialises cos (same GEMM operands, same margin code): loss rows 1e-5, gradients 2e-3 (dcos is rounded to bf16 in both;
    ragged B and C exercise the row / column masks).  The 256x256 TN kernel is forced (the toy width would not select it)."""
    torch.manual_seed(9)
    D, Cn, B = 64, 1003, 70
    h = {"arcface": lambda: heads.ArcFace(D, Cn, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev),
         "circle": lambda: heads.CircleLoss(D, Cn, margin=0.25, gamma=64, backend=be, device=dev),
         "mv_am": lambda: heads.MV_Softmax(D, Cn, is_am=True, margin=0.35, mv_weight=1.12, scale=32, backend=be, device=dev),
         "mv_arc": lambda: heads.MV_Softmax(D, Cn, is_am=False, margin=0.35, mv_weight=1.12, scale=32, backend=be, device=dev)}[tag]()
    feats = torch.randn(B, D).to(dev); labels = torch.randint(0, Cn, (B,)).to(dev)
    labels[0] = Cn - 1; labels[1] = 0
    l0, df0, dW0 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=planes, fused=False)
    be.lib.vdk_gemm_force_kernel(2)
    try:
        l1, df1, dW1 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=planes, fused=True)
    finally:
        be.lib.vdk_gemm_force_kernel(0)
    assert (l1.cpu() - l0.cpu()).abs().max().item() < 1e-5 * l0.abs().max().item() + 1e-5
    assert _rel(df1, df0) < 2e-3 and _rel(dW1, dW0) < 2e-3


@pytest.mark.parametrize("kern", [2, 5])
def test_epilogue_fused_head_on_both_256_kernels(be, dev, kern):
    """the margin-head epilogues (per-slice softmax partials, d cos from the row statistics) on the eight-wave and on the four-wave 256x256 TN kernel (K = 3 planes x 128)"""
    torch.manual_seed(10)
    D, Cn, B = 128, 777, 40
    h = heads.ArcFace(D, Cn, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev)
    feats = torch.randn(B, D).to(dev); labels = torch.randint(0, Cn, (B,)).to(dev)
    l0, df0, dW0 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=3, fused=False)
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        l1, df1, dW1 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=3, fused=True)
        assert be.lib.vdk_gemm_last_kernel() in (kern, 1, 2, 5)     # (the backward GEMMs run after the cos passes)
    finally:
        be.lib.vdk_gemm_force_kernel(0)
    assert (l1.cpu() - l0.cpu()).abs().max().item() < 1e-5 * l0.abs().max().item() + 1e-5
    assert _rel(df1, df0) < 2e-3 and _rel(dW1, dW0) < 2e-3


def test_fused_head_falls_back_on_unserved_shapes(be, dev):
    """fused=True on a class count the 256x256 TN kernel does not serve (padded C < 256 -> VDK_EUNSUPPORTED): the materialised form takes over, no error"""
    torch.manual_seed(12)
    D, Cn, B = 64, 100, 24
    h = heads.ArcFace(D, Cn, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device=dev)
    feats = torch.randn(B, D).to(dev); labels = torch.randint(0, Cn, (B,)).to(dev)
    l0, df0, dW0 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=1, fused=False)
    l1, df1, dW1 = h.margin_ce(feats, labels, label_smoothing=0.1, cos_planes=1, fused=True)
    assert torch.equal(l0.cpu(), l1.cpu()) and torch.equal(df0.cpu(), df1.cpu()) and torch.equal(dW0.cpu(), dW1.cpu())
