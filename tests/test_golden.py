"""Golden fixtures produced by the REFERENCE's own modules (tests/golden/make_golden.py, run where /root/reference
exists) vs (1) the oracle restatements and (2) the kernels through the C ABI (emulated on CPU, real on -m gpu)."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import cbir as ocbir
from visiondk_amd import cbir, ops, schedule

G = Path(__file__).resolve().parent / "golden"


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_ce_and_mixup_kernel_vs_reference_loss(be, dev):
    z = np.load(G / "losses.npz")
    logits = torch.from_numpy(z["ce_logits"]).to(dev); y = torch.from_numpy(z["ce_labels"]).to(dev)
    B = logits.shape[0]
    loss, _, dlf = ops.softmax_ce(logits, y, label_smoothing=float(z["ce_eps"]), grad_scale=1.0 / B, backend=be)
    assert abs(loss.mean().item() - float(z["ce_loss"])) < 1e-6 * abs(float(z["ce_loss"])) + 1e-6
    assert _rel(dlf, z["ce_grad"]) < 1e-5
    yb = torch.from_numpy(z["mix_labels_b"]).to(dev)
    loss2, _, dlf2 = ops.softmax_ce(logits, y, yb, float(z["mix_lam"]), float(z["ce_eps"]), 1.0 / B, backend=be)
    assert abs(loss2.mean().item() - float(z["mix_loss"])) < 2e-6 * abs(float(z["mix_loss"]))
    assert _rel(dlf2, z["mix_grad"]) < 1e-5


def test_bce_kernel_vs_reference_loss(be, dev):
    z = np.load(G / "losses.npz")
    x = torch.from_numpy(z["bce_logits"]).to(dev); t = torch.from_numpy(z["bce_targets"]).to(dev)
    loss, _, dlf = ops.bce_logits(x, t, grad_scale=1.0 / x.numel(), backend=be)
    assert abs(loss.sum().item() / x.numel() - float(z["bce_loss"])) < 1e-6
    assert _rel(dlf, z["bce_grad"]) < 1e-5


def test_ema_kernel_vs_reference_modelema(be, dev):
    z = np.load(G / "ema.npz")
    ema = torch.from_numpy(z["p0"]).clone().to(dev)
    n = ema.numel()
    for it in range(3):
        # feed the fused step a gradient that moves p_{it} to p_{it+1} with lr=1, no momentum/wd, then compare its EMA output
        p = torch.from_numpy(z[f"p{it}"]).clone().to(dev)
        g = (torch.from_numpy(z[f"p{it}"]) - torch.from_numpy(z[f"p{it + 1}"])).to(dev)
        m = torch.zeros(n, device=dev)
        d = 0.9999 * (1 - math.exp(-(it + 1) / 2000))          # models/ema.py:24
        ops.sgd_step(p, g, m, lr=1.0, momentum=0.0, weight_decay=0.0, ema=ema, normsq=None, ema_decay=d, first_step=True, backend=be)
        assert _rel(p, z[f"p{it + 1}"]) < 1e-6
        assert _rel(ema, z[f"ema{it + 1}"]) < 1e-6


def test_schedules_vs_reference_scheduler():
    z = np.load(G / "schedulers.npz")
    for key in z.files:
        name, tag = key.rsplit("_", 1)
        lrf = None if tag == "none" else 0.05
        got = [schedule.lr_at(name, t, warm_ep=1 if "warm" in name else 0, epochs=15, lr0=0.006, lrf_ratio=lrf) for t in range(15)]
        np.testing.assert_allclose(got, z[key], rtol=1e-9, atol=1e-12, err_msg=key)


def test_layer_wise_groups_follow_torch_schedulers():
    """Two parameter groups as built/layer_optimizer.py:26-29 makes them (head at 10 x lr0), driven by the torch schedulers the reference's
    engine/scheduler.py:27-57 assembles: every group scales from its own initial lr, the cosine floor is lrf_ratio * lr0 for both."""
    import torch
    from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR
    lr0, epochs, warm = 0.006, 15, 2
    for name in schedule.SCHEDULERS:
        for lrf in (None, 0.05):
            f = 0.1 if lrf is None else lrf
            a, b = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.SGD([{"params": [a], "lr": lr0}, {"params": [b], "lr": lr0 * 10}], lr=lr0, momentum=0.9)
            if name == "linear":
                sch = LinearLR(opt, start_factor=1, end_factor=f, total_iters=epochs)
            elif name == "cosine":
                sch = CosineAnnealingLR(opt, T_max=epochs, eta_min=f * lr0)
            else:
                tail = (LinearLR(opt, start_factor=1, end_factor=f, total_iters=epochs - warm) if name == "linear_with_warm"
                        else CosineAnnealingLR(opt, T_max=epochs - warm, eta_min=f * lr0))
                sch = SequentialLR(opt, schedulers=[LinearLR(opt, start_factor=0.1, end_factor=1, total_iters=warm), tail], milestones=[warm])
            for t in range(epochs):
                kw = dict(warm_ep=warm if "warm" in name else 0, epochs=epochs, lr0=lr0, lrf_ratio=lrf)
                np.testing.assert_allclose(schedule.lr_at(name, t, **kw), opt.param_groups[0]["lr"], rtol=1e-9, err_msg=f"{name} {t}")
                np.testing.assert_allclose(schedule.lr_at(name, t, base_lr=lr0 * 10, **kw), opt.param_groups[1]["lr"], rtol=1e-9, err_msg=f"{name} head {t}")
                opt.step(); sch.step()


def test_cbir_oracle_and_kernel_vs_float64_fixture(be, dev):
    z = np.load(G / "cbir_small.npz")
    q, g = z["queries"], z["gallery"]
    so, io = ocbir.flat_ip_search(q, g, 100)
    # the oracle's DEFINED fp32 summation order may swap neighbours whose float64 scores differ by < 2e-6
    s64 = z["scores_top100"]
    np.testing.assert_allclose(so, s64, atol=2e-6, rtol=0)
    diff = io != z["idx_top100"]
    if diff.any():
        gaps = np.abs(np.diff(s64, axis=1))
        rows, cols = np.nonzero(diff)
        for r, c in zip(rows, cols):
            near = min(gaps[r, max(c - 1, 0)], gaps[r, min(c, gaps.shape[1] - 1)])
            assert near < 2e-6, (r, c, near)
    assert diff.mean() < 0.01
    index = cbir.FlatIPIndex(128, backend=be, device=dev, cap=4096)
    index.add(g)
    s, i = index.search(q, 100)
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(s.view(np.uint32), so.view(np.uint32))
    # k > N pads with (-FLT_MAX, -1) like faiss
    s2, i2 = index.search(q[:3], 1024)
    assert (i2[:, 1000:] == -1).all() and (s2[:, 1000:] == np.float32(-3.4028234663852886e38)).all()
    assert sorted(i2[0, :1000].tolist()) == list(range(1000))


def test_sam_two_step_vs_reference_optimizer(be, dev):
    """engine/optimizer.py SAM(adaptive=True, rho=0.05) first_step / second_step over SGD(lr .01, mom .937, wd 5e-4)"""
    z = np.load(G / "sam.npz")
    p = torch.from_numpy(z["p_init"]).clone().to(dev)
    g1 = torch.from_numpy(z["g1"]).to(dev); g2 = torch.from_numpy(z["g2"]).to(dev)
    old = torch.empty_like(p); m = torch.zeros_like(p)
    ops.sam_first_step(p, g1, old, rho=0.05, adaptive=True, backend=be)
    assert _rel(p, z["p_perturbed"]) < 1e-6
    assert torch.equal(old.cpu(), torch.from_numpy(z["p_init"]))
    p.copy_(old)                                   # second_step: back to w, then the base optimizer steps on the new grads
    ops.sgd_step(p, g2, m, lr=0.01, momentum=0.937, weight_decay=5e-4, normsq=None, first_step=True, backend=be)
    assert _rel(p, z["p_final"]) < 1e-6


def test_cbir_metrics_vs_reference_class():
    from visiondk_amd import metrics
    z = np.load(G / "cbir_metrics.npz")
    labels = [row[row >= 0] for row in z["labels"]]
    got = metrics.compute_metrics(z["preds"], z["scores"], labels, cutoffs=(1, 3, 10))
    exp = dict(zip(z["names"].tolist(), z["values"].tolist()))
    assert set(got) == set(exp)
    for k in exp:
        assert abs(got[k] - exp[k]) < 1e-12, (k, got[k], exp[k])


def test_focal_kernel_vs_reference_loss(be, dev):
    z = np.load(G / "losses.npz")
    x = torch.from_numpy(z["bce_logits"]).to(dev); t = torch.from_numpy(z["bce_targets"]).to(dev)
    loss, _, dlf = ops.bce_logits(x, t, grad_scale=1.0 / x.numel(), focal_gamma=1.5, focal_alpha=0.25, backend=be)
    assert abs(loss.sum().item() / x.numel() - float(z["focal_loss"])) < 1e-6
    assert _rel(dlf, z["focal_grad"]) < 1e-5


def test_face_pair_verification_matches_reference_evaluator():
    """visiondk_amd.evaluate.FaceEvaluator vs the reference's own Evaluator (engine/faceX/evaluation.py:18-118) run on the same embeddings / pair list
    (tests/golden/make_golden.py)."""
    from visiondk_amd.evaluate import FaceEvaluator
    d = np.load(G / "face_pairs.npz")
    feats, a, b, label = d["feats"], d["a"], d["b"], d["label"]
    names = [f"id/img{i:04d}.jpg" for i in range(len(feats))]
    pair_list = [[names[x], names[y], str(int(l))] for x, y, l in zip(a, b, label)]
    ev = FaceEvaluator()
    mean, std = ev.test_one_model(pair_list, {n: f for n, f in zip(names, feats)})
    assert mean == float(d["mean"]) and abs(std - float(d["std"])) < 1e-15
    thr = ev.get_threshold(np.array([feats[x] @ feats[y] for x, y in zip(a[:5400], b[:5400])], dtype=np.float32), label[:5400].astype(np.int8))
    assert thr == float(d["thr_first_5400"])
