"""Generate the committed golden fixtures from the REFERENCE's own code, run in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

/root/reference's torch-only modules are imported BY FILE PATH (whole-package imports fail: timm / torchvision /
faiss / torchmetrics are not installable here).  The fixtures travel to the GPU box; /root/reference does
not.  Everything is seeded; re-running reproduces the files bit-for-bit on the same torch build.
"""
import importlib.util
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True


def load(rel, name):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    torch.manual_seed(0)
    loss_m = load("models/losses/loss.py", "ref_loss")
    ema_m = load("models/ema.py", "ref_ema")
    sch_m = load("engine/scheduler.py", "ref_sched")
    opt_m = load("engine/optimizer.py", "ref_opt")
    arc_m = load("models/faceX/head/arcface.py", "ref_arc")
    cir_m = load("models/faceX/head/circleloss.py", "ref_circle")
    mv_m = load("models/faceX/head/mv_softmax.py", "ref_mv")

    # ---- losses (models/losses/loss.py:68-76) ---------------------------------------------------------
    logits = (torch.randn(8, 257) * 3).requires_grad_(True)
    y = torch.randint(0, 257, (8,))
    ce = loss_m.create_Lossfn("ce")(label_smooth=0.05)
    l = ce(logits, y); l.backward()
    out = {"ce_logits": logits.detach().numpy(), "ce_labels": y.numpy(), "ce_eps": np.float32(0.05), "ce_loss": l.detach().numpy(),
           "ce_grad": logits.grad.numpy().copy()}
    # mixup_criterion (engine/procedure/train.py:34-35)
    yb = torch.randint(0, 257, (8,)); lam = 0.3
    lg2 = logits.detach().clone().requires_grad_(True)
    lm = lam * ce(lg2, y) + (1 - lam) * ce(lg2, yb); lm.backward()
    out.update(mix_labels_b=yb.numpy(), mix_lam=np.float32(lam), mix_loss=lm.detach().numpy(), mix_grad=lg2.grad.numpy().copy())
    bl = (torch.randn(8, 5) * 2).requires_grad_(True); bt = (torch.rand(8, 5) > 0.5).float()
    lb = loss_m.create_Lossfn("bce")()(bl, bt); lb.backward()
    out.update(bce_logits=bl.detach().numpy(), bce_targets=bt.numpy(), bce_loss=lb.detach().numpy(), bce_grad=bl.grad.numpy().copy())
    fl = bl.detach().clone().requires_grad_(True)
    lf = loss_m.create_Lossfn("focal")()(fl, bt); lf.backward()
    out.update(focal_loss=lf.detach().numpy(), focal_grad=fl.grad.numpy().copy())
    np.savez(OUT / "losses.npz", **out)

    # ---- EMA (models/ema.py:28-37) over a model state_dict, 3 updates ---------------------------------------
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5))
    ema = ema_m.ModelEMA(model)
    rec = {"p0": torch.cat([p.detach().flatten() for p in model.parameters()]).numpy()}
    for it in range(3):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        ema.update(model)
        rec[f"p{it + 1}"] = torch.cat([p.detach().flatten() for p in model.parameters()]).numpy()
        rec[f"ema{it + 1}"] = torch.cat([p.detach().flatten() for p in ema.ema.parameters()]).numpy()
    np.savez(OUT / "ema.npz", **rec)

    # ---- LR schedulers (engine/scheduler.py:27-57), stepping once per epoch, 15 epochs ---------------------------
    sch = {}
    for name in ["linear", "cosine", "linear_with_warm", "cosine_with_warm"]:
        for lrf in [None, 0.05]:
            p = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.SGD([p], lr=0.006, momentum=0.937)
            s = sch_m.create_Scheduler(name, opt, warm_ep=1 if "warm" in name else 0, epochs=15, lr0=0.006, lrf_ratio=lrf)
            lrs = []
            for ep in range(15):
                lrs.append(opt.param_groups[0]["lr"])
                opt.step(); s.step()
            sch[f"{name}_{'none' if lrf is None else 'lrf05'}"] = np.array(lrs, np.float64)
    np.savez(OUT / "schedulers.npz", **sch)

    # ---- SAM (engine/optimizer.py:29-87): adaptive two-step on a 3-tensor toy problem ---------------------------
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(2, 2))]
    opt = opt_m.create_Optimizer("sam", lr=0.01, weight_decay=5e-4, momentum=0.937, params=ps)
    g1 = [torch.randn_like(p) for p in ps]; g2 = [torch.randn_like(p) for p in ps]
    rec = {"p_init": torch.cat([p.detach().flatten() for p in ps]).numpy(), "g1": torch.cat([g.flatten() for g in g1]).numpy(),
           "g2": torch.cat([g.flatten() for g in g2]).numpy()}
    for p, g in zip(ps, g1):
        p.grad = g.clone()
    opt.first_step(zero_grad=True)
    rec["p_perturbed"] = torch.cat([p.detach().flatten() for p in ps]).numpy()
    for p, g in zip(ps, g2):
        p.grad = g.clone()
    opt.second_step(zero_grad=True)
    rec["p_final"] = torch.cat([p.detach().flatten() for p in ps]).numpy()
    np.savez(OUT / "sam.npz", **rec)

    # ---- margin heads (models/faceX/head/*.py) + CE, B=8, D=64, C=257 ---------------------------------------------
    torch.manual_seed(2)
    heads = {}
    feats0 = torch.randn(8, 64)
    labels = torch.randint(0, 257, (8,))
    for tag, mk in [("arcface", lambda: arc_m.ArcFace(64, 257, margin_arc=0.35, margin_am=0.0, scale=32)),
                    ("circle", lambda: cir_m.CircleLoss(64, 257, margin=0.25, gamma=256)),
                    ("mv_am", lambda: mv_m.MV_Softmax(64, 257, is_am=True, margin=0.35, mv_weight=1.12, scale=32)),
                    ("mv_arc", lambda: mv_m.MV_Softmax(64, 257, is_am=False, margin=0.35, mv_weight=1.12, scale=32))]:
        torch.manual_seed(3)
        h = mk()
        f = feats0.clone().requires_grad_(True)
        if tag == "arcface":   # plant a target column close to its feature so that cos(theta) > cos(pi - m) branch and a near-1 cosine are hit
            with torch.no_grad():
                h.weight[:, labels[0]] = feats0[0] + 0.05 * torch.randn(64)
        logits = h(f, labels)
        loss = torch.nn.functional.cross_entropy(logits, labels)
        loss.backward()
        heads.update({f"{tag}_weight": h.weight.detach().numpy().copy(), f"{tag}_logits": logits.detach().numpy(),
                      f"{tag}_loss": loss.detach().numpy(), f"{tag}_dfeats": f.grad.numpy().copy(), f"{tag}_dweight": h.weight.grad.numpy().copy()})
    heads["feats"] = feats0.numpy(); heads["labels"] = labels.numpy()
    np.savez(OUT / "heads.npz", **heads)
    make_magface()

    # ---- retrieval: IndexFlatIP semantics stated in float64 (faiss is not installable; engine/cbir/evaluation.py:193) ----
    rng = np.random.default_rng(0)
    g = rng.standard_normal((1000, 128)).astype(np.float32); q = rng.standard_normal((33, 128)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True); q /= np.linalg.norm(q, axis=1, keepdims=True)
    s64 = q.astype(np.float64) @ g.astype(np.float64).T
    order = np.argsort(-s64, axis=1, kind="stable")[:, :100]
    np.savez(OUT / "cbir_small.npz", gallery=g, queries=q, idx_top100=order.astype(np.int64),
             scores_top100=np.take_along_axis(s64, order, 1))
    # ---- retrieval metrics: the reference's CBIRMetrics class, exec'd from its own source (engine/cbir/evaluation.py:14-103;
    #      the module itself cannot be imported: its top-level imports need torchvision / faiss) ------------------------------------
    src = (REF / "engine/cbir/evaluation.py").read_text().splitlines()
    ns = {"np": np}
    from sklearn.metrics import roc_auc_score, ndcg_score
    ns.update(roc_auc_score=roc_auc_score, ndcg_score=ndcg_score)
    exec("\n".join(src[13:103]), ns)
    rng = np.random.default_rng(5)
    Q, k = 12, 20
    preds = np.stack([rng.permutation(200)[:k] for _ in range(Q)])
    labels = [rng.choice(200, size=rng.integers(1, 6), replace=False) for _ in range(Q)]
    for i in range(Q):   # plant some relevant items among the predictions
        if i % 3 != 2:
            preds[i, rng.integers(0, k)] = labels[i][0]
    scores = np.sort(rng.random((Q, k)))[:, ::-1].copy()
    m = ns["CBIRMetrics"](cutoffs=[1, 3, 10])
    sp = [[str(v) for v in row] for row in preds]; sl = [[str(v) for v in row] for row in labels]
    m.compute_mrr(sp, sl); m.compute_precision(sp, sl); m.compute_recall(sp, sl); m.compute_auc(sp, sl, scores); m.compute_ndcg(sp, sl, scores)
    np.savez(OUT / "cbir_metrics.npz", preds=preds, scores=scores, labels=np.array([np.pad(l, (0, 6 - len(l)), constant_values=-1) for l in labels]),
             names=np.array(list(m.metrics.keys())), values=np.array(list(m.metrics.values()), dtype=np.float64))
    # ---- face pair verification: the reference's own Evaluator (engine/faceX/evaluation.py:18-118), exec'd from its source range ----------------------
    src = (REF / "engine/faceX/evaluation.py").read_text().splitlines()
    ns = {"np": np, "os": __import__("os")}
    exec("\n".join(src[17:118]), ns)
    rng = np.random.default_rng(9)
    n_img, dim, n_pairs = 400, 32, 6000                       # the reference hard-codes 10 folds of 600 pairs
    ident = rng.integers(0, 80, n_img)
    centers = rng.standard_normal((80, dim))
    feats = centers[ident] + 0.9 * rng.standard_normal((n_img, dim))
    feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)
    a, b = rng.integers(0, n_img, n_pairs), rng.integers(0, n_img, n_pairs)
    same = rng.random(n_pairs) < 0.5
    for i in np.where(same)[0]:                               # make about half of the pairs genuine
        cand = np.where(ident == ident[a[i]])[0]
        b[i] = rng.choice(cand)
    label = (ident[a] == ident[b]).astype(np.int64)
    names = [f"id{ident[i]:03d}/img{i:04d}.jpg" for i in range(n_img)]
    pair_list = [[names[x], names[y], str(int(l))] for x, y, l in zip(a, b, label)]
    ev = ns["Evaluator"](None)
    mean, std = ev.test_one_model(pair_list, {n: f for n, f in zip(names, feats)})
    thr = ev.getThreshold(np.array([feats[x] @ feats[y] for x, y in zip(a[:5400], b[:5400])], dtype=np.float32), label[:5400].astype(np.int8))
    np.savez(OUT / "face_pairs.npz", feats=feats, a=a, b=b, label=label, mean=np.float64(mean), std=np.float64(std), thr_first_5400=np.float64(thr))
    # ---- validation-time input pipeline: the reference's own ResizeAndPadding2Square (dataset/transforms.py:325-365), exec'd from its source range
    # (the module itself imports cv2 / torchvision, which are not installed); ToTensor + Normalize are torchvision's two float32 tensor expressions.
    from PIL import Image, ImageOps
    src = (REF / "dataset/transforms.py").read_text().splitlines()
    ns = {"Image": Image, "ImageOps": ImageOps, "random": __import__("random")}
    exec("\n".join(src[324:365]), ns)
    rng = np.random.default_rng(11)
    fix = {}
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for i, (w, h, size) in enumerate([(61, 47, 64), (47, 61, 64), (130, 20, 64), (33, 33, 64), (20, 200, 64), (200, 150, 96), (49, 49, 224), (9, 5, 32)]):
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 256, (h, w))], axis=2).astype(np.uint8)
        arr[h // 3: h // 2, w // 4: w // 2] = rng.integers(0, 256, (h // 2 - h // 3, w // 2 - w // 4, 3), dtype=np.uint8)
        out = np.asarray(ns["ResizeAndPadding2Square"](size=size, training=False)(Image.fromarray(arr)))
        t = torch.from_numpy(out.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t.sub_(torch.as_tensor(mean, dtype=torch.float32)[:, None, None]).div_(torch.as_tensor(std, dtype=torch.float32)[:, None, None])
        fix[f"in{i}"], fix[f"u8_{i}"], fix[f"f32_{i}"], fix[f"size{i}"] = arr, out, t.numpy(), np.int64(size)
    # soft BCE targets: the reference's own staticmethod (dataset/basedataset.py:198-231), exec'd from its source range
    import textwrap
    src = (REF / "dataset/basedataset.py").read_text().splitlines()
    ns = {"torch": torch}
    exec(textwrap.dedent("\n".join(src[197:231])), ns)
    slt = ns["set_label_transforms"].__func__ if hasattr(ns["set_label_transforms"], "__func__") else ns["set_label_transforms"]
    fix["lab_int"] = slt(3, 7, 0.1).numpy()
    fix["lab_list"] = slt([0, 1, 0, 0, 1, 0, 1], 7, 0.2).numpy()
    fix["lab_onehot"] = slt(torch.tensor([1., 0., 0., 1., 0., 0., 0.]), 7, 0.1).numpy()
    fix["lab_onehot_nosmooth"] = slt(torch.tensor([1., 0., 0., 1., 0., 0., 0.]), 7, 0.0).numpy()
    np.savez_compressed(OUT / "preprocess.npz", n=np.int64(8), mean=np.array(mean), std=np.array(std), **fix)
    print("golden fixtures written to", OUT)


def make_magface():
    """models/faceX/head/magface.py (dead upstream: its tuple output is never consumed) -> tests/golden/magface.npz; loss = CE(logits) + mean(lamda * loss_g)"""
    mag_m = load("models/faceX/head/magface.py", "ref_mag")
    torch.manual_seed(21)
    feats0 = torch.randn(8, 64) * torch.tensor([0.5, 1.5, 3.0, 6.0, 9.0, 12.0, 14.0, 20.0]).view(8, 1)    # norms ~4 .. 160: both clamp ends and the linear part of the margin
    labels = torch.randint(0, 257, (8,))
    torch.manual_seed(22)
    h = mag_m.MagFace(64, 257, margin_am=0.0, scale=32, l_a=10, u_a=110, l_margin=0.45, u_margin=0.8, lamda=20)
    with torch.no_grad():
        h.weight[:, labels[3]] = feats0[3] + 0.3 * torch.randn(64)        # a target column close to its feature: the active margin branch with a large cosine
    f = feats0.clone().requires_grad_(True)
    logits, reg = h(f, labels)
    loss = torch.nn.functional.cross_entropy(logits, labels) + reg.mean()
    loss.backward()
    np.savez(OUT / "magface.npz", feats=feats0.numpy(), labels=labels.numpy(), weight=h.weight.detach().numpy().copy(), logits=logits.detach().numpy(),
             reg=reg.detach().numpy(), loss=loss.detach().numpy(), dfeats=f.grad.numpy().copy(), dweight=h.weight.grad.numpy().copy())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "magface":
        make_magface()
        raise SystemExit(0)
    main()
