"""Host logic: retrieval metrics of the reference (engine/cbir/evaluation.py:14-103 `CBIRMetrics`, :202-224 `compute_metrics`),
restated over arrays instead of Python lists of path strings.  Stays on the host like the reference (O(Q*k) work).

preds: [Q, k] retrieved ids per query (what `valuate` builds from the search indices, evaluation.py:269-273),
labels: list of arrays of relevant ids per query, preds_scores: [Q, k]."""
from __future__ import annotations

from typing import Sequence

import numpy as np


def _hits(preds, labels) -> np.ndarray:
    """[Q, k] 0/1: `np.isin(pred, label)` per query (encode_pred2hard, evaluation.py:94-100)"""
    preds = np.asarray(preds)
    return np.stack([np.isin(p, np.asarray(l)) for p, l in zip(preds, labels)]).astype(int)


class CBIRMetrics:
    def __init__(self, cutoffs: Sequence[int] = (1, 10, 100)):
        self.cutoffs = list(cutoffs)
        self.metrics = {}

    def compute_mrr(self, preds, labels):
        h = _hits(preds, labels)
        first = np.where(h.any(1), h.argmax(1) + 1, 0)                  # rank of the first relevant item, 0 = none
        for c in self.cutoffs:
            rr = np.where((first > 0) & (first <= c), 1.0 / np.maximum(first, 1), 0.0)
            self.metrics[f"MRR@{c}"] = rr.sum() / len(h)

    def compute_recall(self, preds, labels):
        preds = np.asarray(preds)
        for c in self.cutoffs:
            v = [len(np.intersect1d(l, p[:c])) / len(l) for p, l in zip(preds, labels)]
            self.metrics[f"Recall@{c}"] = float(np.sum(v)) / len(preds)

    def compute_precision(self, preds, labels):
        preds = np.asarray(preds)
        for c in self.cutoffs:
            v = [len(np.intersect1d(l, p[:c])) / min(c, len(l)) for p, l in zip(preds, labels)]   # engine/ version, not cbir_eval.py
            self.metrics[f"Precision@{c}"] = float(np.sum(v)) / len(preds)

    def compute_auc(self, preds, labels, preds_scores):
        from sklearn.metrics import roc_auc_score
        self.metrics[f"AUC@{self.cutoffs[-1]}"] = roc_auc_score(_hits(preds, labels).flatten(), np.asarray(preds_scores).flatten())

    def compute_ndcg(self, preds, labels, preds_scores):
        from sklearn.metrics import ndcg_score
        h = _hits(preds, labels)
        for c in self.cutoffs:
            self.metrics[f"nDCG@{c}"] = ndcg_score(h, np.asarray(preds_scores), k=c)

    def reset(self):
        self.metrics.clear()


def compute_metrics(preds, preds_scores, labels, metrics=("mrr", "precision", "recall", "auc", "ndcg"), cutoffs=(1, 3, 10)):
    eng = CBIRMetrics(cutoffs=cutoffs)
    for m in metrics:
        if m == "mrr":
            eng.compute_mrr(preds, labels)
        elif m == "precision":
            eng.compute_precision(preds, labels)
        elif m == "recall":
            eng.compute_recall(preds, labels)
        elif m == "auc":
            eng.compute_auc(preds, labels, preds_scores)
        elif m == "ndcg":
            eng.compute_ndcg(preds, labels, preds_scores)
        else:
            raise ValueError(f"{m} is not supported")
    return eng.metrics
