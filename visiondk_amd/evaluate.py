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


class FaceEvaluator:
    """The reference's face pair-verification `Evaluator` (engine/faceX/evaluation.py:18-118): 10 folds of 600 pairs, cosine score of the two unit
    embeddings of each pair, threshold picked on the other 9 folds as the argmax of TPR - FPR over 1000 evenly spaced thresholds, accuracy on the held-out
    fold; returns (mean accuracy, standard error).  Embeddings come from `face.FeatureExtractor.extract_face` (device work); this is host bookkeeping."""

    def __init__(self, feature_extractor=None):
        self.feature_extractor = feature_extractor

    def test(self, pair_list, feature_dataloader, device):
        assert len(pair_list) % 10 == 0, "make sure the number of rows is a multiple of 10 in pair.txt"
        return self.test_one_model(pair_list, self.feature_extractor.extract_face(feature_dataloader, device))

    @staticmethod
    def get_threshold(score_list: np.ndarray, label_list: np.ndarray, num_thresholds: int = 1000) -> float:
        pos, neg = score_list[label_list == 1], score_list[label_list == 0]
        smin, smax = np.min(score_list), np.max(score_list)
        step = (smax - smin) / num_thresholds
        thr = smin + step * np.array(range(1, num_thresholds + 1))
        tpr = (pos[None, :] > thr[:, None]).sum(1) / pos.size
        fpr = (neg[None, :] > thr[:, None]).sum(1) / neg.size
        return thr[int(np.argmax(tpr - fpr))]

    def test_one_model(self, test_pair_list, image_name2feature, is_normalize: bool = True):
        import os
        nps = len(test_pair_list)
        scores = np.zeros((10, nps // 10), dtype=np.float32)
        labels = np.zeros((10, nps // 10), dtype=np.int8)
        for index, pair in enumerate(test_pair_list):
            f1, f2 = image_name2feature[os.path.normpath(pair[0])], image_name2feature[os.path.normpath(pair[1])]
            if not is_normalize:
                f1, f2 = f1 / np.linalg.norm(f1), f2 / np.linalg.norm(f2)
            scores[index // 600][index % 600] = np.dot(f1, f2)          # the reference hard-codes 600 pairs per fold
            labels[index // 600][index % 600] = int(pair[2])
        accs = []
        keep = np.ones(10, dtype=bool)
        for k in range(10):
            keep[k] = False
            thr = self.get_threshold(scores[keep].flatten(), labels[keep].flatten())
            keep[k] = True
            tp = np.sum(scores[k][labels[k] == 1] > thr)
            tn = np.sum(scores[k][labels[k] == 0] < thr)
            accs.append((tp + tn) / 600)
        return float(np.mean(accs)), float(np.std(accs, ddof=1) / np.sqrt(10))
