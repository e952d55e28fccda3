"""Evaluation callers of hot path A: the reference's classification `valuate` (engine/procedure/evaluation.py:52-185).

Device work = the model forward (native engine) and the per-row top-k (vdk_topk_rows, replacing `y.argsort(1, descending=True)[:, :top_k]`);
the accuracy / precision / recall / F1 bookkeeping is O(B*k) host arithmetic on the gathered predictions, like the reference's
(torchmetrics `Precision/Recall/F1Score(task='multilabel', average=None)`: per class TP/(TP+FP), TP/(TP+FN), 2TP/(2TP+FP+FN), 0 where the
denominator is 0, then the mean over classes)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops


def valuate(model, dataloader, device, pbar=None, is_training: bool = False, lossfn: Optional[Callable] = None, logger=None,
            thresh: Union[float, Sequence[float]] = 0, top_k: int = 5, conm_path: Optional[str] = None, class_indices: Optional[List[str]] = None):
    """Same arguments and return values as the reference: single-label (thresh == 0): (top1, topk[, loss]); multi-label (BCE, thresh in (0,1),
    scalar or per class): (mean precision, mean recall, mean F1[, loss])."""
    classes = class_indices if class_indices is not None else list(dataloader.dataset.class_indices)
    ncls = len(classes)
    single = isinstance(thresh, (int, float)) and thresh == 0
    if not single:
        t = np.full(ncls, float(thresh), dtype=np.float64) if isinstance(thresh, (int, float)) else np.asarray(thresh, dtype=np.float64)
        assert t.shape == (ncls,), f"Number of thresholds ({t.size}) must match number of classes ({ncls})"
        assert (t > 0).all() and (t < 1).all(), "For multi-label (BCE), all thresholds should be in (0, 1)"
        logit_t = torch.from_numpy(np.log(t / (1 - t)).astype(np.float32))   # sigmoid(y) >= t  <=>  y >= logit(t)
    model.eval()
    be = getattr(getattr(model, "engine", None), "be", None)
    preds, targets, loss, n = [], [], 0.0, 0
    with torch.no_grad():
        for images, labels in dataloader:
            images, labels = images.to(device), labels.to(device)
            y = model(images)
            if single:
                _, idx = ops.topk_rows(y.contiguous(), min(top_k, y.shape[1]), backend=be)
                preds.append(idx.cpu().numpy()); targets.append(labels.cpu().numpy())
            else:
                preds.append((y.cpu() >= logit_t).numpy())
                targets.append((labels.round() == 1).cpu().numpy())
            if lossfn:
                loss += float(lossfn(y, labels))
            n += 1
    loss /= max(n, 1)
    pred, tgt = np.concatenate(preds), np.concatenate(targets)
    say = (logger.console if not is_training else logger.log) if logger is not None else (lambda *_: None)
    if single:
        correct = tgt[:, None] == pred
        top1, topk = float(correct[:, 0].mean()), float(correct.any(1).mean())
        say(f'{"name":<15}{"nums":>8}{"top1":>10}{f"top{top_k}":>10}')
        for i, c in enumerate(classes):
            m = tgt == i
            if m.any():
                say(f"{c:<15}{int(m.sum()):>8}{correct[m, 0].mean():>10.3f}{correct[m].any(1).mean():>10.3f}")
        say(f'{"    ":<15}{len(tgt):>8}{top1:>10.3f}{round(topk, 3):>10.3f}')
        return (top1, topk, loss) if lossfn else (top1, topk)
    tp = (pred & tgt).sum(0).astype(np.float64); fp = (pred & ~tgt).sum(0).astype(np.float64); fn = (~pred & tgt).sum(0).astype(np.float64)
    safe = lambda a, b: np.divide(a, b, out=np.zeros_like(a), where=b > 0)
    precision, recall, f1 = safe(tp, tp + fp), safe(tp, tp + fn), safe(2 * tp, 2 * tp + fp + fn)
    say(f'{"name":<8}{"nums":>8}{"precision":>10}{"recall":>10}{"f1-score":>10}{"thresh":>10}')
    for i, c in enumerate(classes):
        say(f"{c:<8}{int(tgt[:, i].sum()):>8}{precision[i]:>10.3f}{recall[i]:>10.3f}{f1[i]:>10.3f}{t[i]:>10.3f}")
    say(f"mprecision:{precision.mean():.3f}, mrecall:{recall.mean():.3f}, mf1-score:{f1.mean():.3f}")
    out = (float(precision.mean()), float(recall.mean()), float(f1.mean()))
    return out + (loss,) if lossfn else out
