"""Host logic: the reference's LR schedules as closed forms (engine/scheduler.py:27-57 builds them from torch's LinearLR /
CosineAnnealingLR / SequentialLR; `lrf_ratio=None -> 0.1`, :24-25).  `lr_at(name, t, ...)` is the learning rate in effect
for step index t (the value `optimizer.param_groups[0]['lr']` holds before the t-th `scheduler.step()`), so a fused
optimizer can be driven without torch scheduler objects: `step.param_groups[0]['lr'] = lr_at(...)`."""
from __future__ import annotations

import math

SCHEDULERS = ("linear", "cosine", "linear_with_warm", "cosine_with_warm")


def _linear(lr0, start, end, t, total):
    return lr0 * (start + (end - start) * min(t, total) / total)


def _cosine(lr0, eta_min, t, t_max):
    return eta_min + (lr0 - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2


def lr_at(name: str, t: int, *, warm_ep: int, epochs: int, lr0: float, lrf_ratio=None, base_lr=None) -> float:
    """base_lr: the parameter group's own initial lr when it differs from lr0 (built/layer_optimizer.py:26-29 gives the head 10 x lr0);
    torch's schedulers scale every group from its own initial lr, but the cosine floor eta_min = lrf_ratio * lr0 is one absolute
    number for all groups (scheduler.py:33,54)."""
    lrf = 0.1 if lrf_ratio is None else lrf_ratio
    eta_min = lrf * lr0
    if base_lr is not None:
        lr0 = base_lr
    if name == "linear":
        return _linear(lr0, 1.0, lrf, t, epochs)
    if name == "cosine":
        return _cosine(lr0, eta_min, t, epochs)
    if name == "linear_with_warm":
        if t < warm_ep:
            return _linear(lr0, 0.1, 1.0, t, warm_ep)
        return _linear(lr0, 1.0, lrf, t - warm_ep, epochs - warm_ep)
    if name == "cosine_with_warm":
        if t < warm_ep:
            return _linear(lr0, 0.1, 1.0, t, warm_ep)
        return _cosine(lr0, eta_min, t - warm_ep, epochs - warm_ep)
    raise KeyError(f"unknown scheduler '{name}' (have {SCHEDULERS})")
