"""TEST INFRASTRUCTURE: ctypes binding of oracle/cbir_oracle.c (hot path B: IndexFlatIP search as used by
/root/reference engine/cbir/evaluation.py:155-200).  See the C file's header for what is restated/defined."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .build import build

_lib = None


def usable_cores() -> int:
    """Host cores this process may really use: min(affinity, cgroup cpu quota) — a 256-core box may grant 16."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.oracle_l2norm_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]
        _lib.oracle_flat_ip_search.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                               C.c_int64, C.c_void_p, C.c_void_p]
        _lib.oracle_merge_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_void_p]
        _lib.oracle_ip_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        _lib.oracle_ip_pair.restype = C.c_float
        _lib.oracle_set_threads.argtypes = [C.c_int]
        _lib.oracle_set_threads(usable_cores())
    return _lib


def l2norm_rows(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_l2norm_rows(x.ctypes.data, x.shape[0], x.shape[1], eps, out.ctypes.data)
    return out


def flat_ip_search(q: np.ndarray, g: np.ndarray, k: int, idx_base: int = 0):
    """(scores float32 [nq,k] descending, indices int64 [nq,k]); ties -> lower index; pad (-FLT_MAX, -1)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    g = np.ascontiguousarray(g, dtype=np.float32).reshape(-1, q.shape[1])
    scores = np.empty((q.shape[0], k), np.float32)
    idx = np.empty((q.shape[0], k), np.int64)
    lib().oracle_flat_ip_search(q.ctypes.data, q.shape[0], g.ctypes.data, g.shape[0], q.shape[1], k, idx_base,
                                scores.ctypes.data, idx.ctypes.data)
    return scores, idx


def merge_topk(scores: np.ndarray, idx: np.ndarray):
    """scores/idx [S,nq,k] -> merged [nq,k] with the same total order."""
    scores = np.ascontiguousarray(scores, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    S, nq, k = scores.shape
    o_s = np.empty((nq, k), np.float32)
    o_i = np.empty((nq, k), np.int64)
    lib().oracle_merge_topk(scores.ctypes.data, idx.ctypes.data, S, nq, k, o_s.ctypes.data, o_i.ctypes.data)
    return o_s, o_i
