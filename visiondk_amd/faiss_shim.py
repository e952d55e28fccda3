"""A `faiss`-shaped module over the HIP retrieval kernels, so that the reference's OWN `index()` / `search()` run unmodified.

The reference's retrieval path (engine/cbir/evaluation.py:106-200, twin cbir_eval.py:35-122) talks to faiss through exactly these names:

    faiss.index_factory(dim, "Flat", faiss.METRIC_INNER_PRODUCT)      evaluation.py:155
    co = faiss.GpuMultipleClonerOptions(); co.useFloat16 = True       evaluation.py:159-160  (commented twin: faiss.GpuClonerOptions, :158)
    faiss.index_cpu_to_all_gpus(faiss_index, co)                      evaluation.py:162      (commented twin: index_cpu_to_gpu(StandardGpuResources(), 0, index, co), :161)
    faiss_index.train(x); faiss_index.add(x)                          evaluation.py:167-168
    faiss_index.search(q, k=k) -> (scores, indices)                   evaluation.py:193
    faiss.Index                                                       evaluation.py:173 (annotation)

`sys.modules["faiss"] = visiondk_amd.faiss_shim` (or `import visiondk_amd.faiss_shim as faiss` in the reference's two files) is the whole integration; every index
is a `visiondk_amd.cbir.FlatIPIndex` (exact inner product, (score desc, index asc), pads (-FLT_MAX, -1)) resident on the GPU.  There is no CPU index: what faiss calls the
"cpu index" is already a device index with fp32 storage, and the `index_cpu_to_*` calls return its fp16-storage clone when `co.useFloat16` is set -- the arithmetic of
faiss's GPU flat index with useFloat16 (fp16 rows, fp16-rounded queries, fp32 accumulation).  faiss-gpu itself is not a dependency (BASELINE.json: "no faiss-gpu").

tests/test_faiss_shim.py executes the reference's two functions, read from /root/reference at test time, against this module and requires bit-equal results with the oracle.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import cbir

METRIC_INNER_PRODUCT = cbir.METRIC_INNER_PRODUCT
METRIC_L2 = 1          # faiss's constant; only inner-product indexes are built (the reference never asks for another)

Index = cbir.FlatIPIndex            # `faiss.Index` in annotations
IndexFlatIP = cbir.FlatIPIndex      # `faiss.IndexFlatIP(d)`

# where indexes are built: None = the product backend on the current CUDA device.  Tests inject the CPU SIMT emulation here.
_DEFAULT = {"backend": None, "device": None}


def set_default(backend=None, device=None) -> None:
    """backend / device every index of this module is built on (tests: the emulated backend on "cpu"; a multi-GPU host: this rank's device)"""
    _DEFAULT["backend"], _DEFAULT["device"] = backend, device


def _kw(device=None) -> dict:
    return {"backend": _DEFAULT["backend"], "device": device if device is not None else _DEFAULT["device"]}


def index_factory(d: int, description: str = "Flat", metric: int = METRIC_INNER_PRODUCT) -> cbir.FlatIPIndex:
    """faiss.index_factory(dim, "Flat", faiss.METRIC_INNER_PRODUCT) (evaluation.py:155).  Any other description / metric is refused: the reference builds no other index."""
    return cbir.index_factory(int(d), description, metric, **_kw())


class GpuClonerOptions:
    """faiss.GpuClonerOptions: the fields the reference (and faiss's own defaults) name; only useFloat16 changes the arithmetic here"""

    def __init__(self):
        self.useFloat16 = False              # store the vectors as fp16 (what the reference sets, evaluation.py:160)
        self.useFloat16CoarseQuantizer = False
        self.usePrecomputed = False
        self.indicesOptions = 0
        self.reserveVecs = 0
        self.storeTransposed = False
        self.verbose = False


class GpuMultipleClonerOptions(GpuClonerOptions):
    """faiss.GpuMultipleClonerOptions (evaluation.py:159): shard = False replicates the index on every GPU (the reference's setting)"""

    def __init__(self):
        super().__init__()
        self.shard = False
        self.shard_type = 1
        self.common_ivf_quantizer = False


class StandardGpuResources:
    """faiss.StandardGpuResources: faiss's per-GPU scratch allocator.  The HIP index owns its workspace (grow-only, sized per search), so this holds nothing."""

    def setTempMemory(self, nbytes: int) -> None:      # accepted for source compatibility
        self.temp_memory = int(nbytes)

    def noTempMemory(self) -> None:
        self.temp_memory = 0


def _clone(index: cbir.FlatIPIndex, co: Optional[GpuClonerOptions], device=None) -> cbir.FlatIPIndex:
    if not isinstance(index, cbir.FlatIPIndex):
        raise TypeError("expected an index built by this module's index_factory / IndexFlatIP")
    if co is not None and getattr(co, "shard", False):
        raise NotImplementedError("co.shard = True (faiss splits the vectors over the GPUs of ONE process): this library shards one process per GPU -- "
                                  "cbir.search_sharded (all-gather of the queries, per-shard search, exact merge)")
    storage = "float16" if (co is not None and co.useFloat16) else "float32"
    if storage == "float16" and index.method != "prefilter":
        storage = "float32"      # d > 512: the exact fp32-MFMA scan serves it; fp16 storage is built for d <= 512 (faiss would store fp16 here: results then differ by fp16 rounding)
    out = cbir.FlatIPIndex(index.d, backend=index.be, device=device if device is not None else index.device, cap=index.cap, idx_base=index.idx_base,
                           method="auto", storage=storage)
    if index.ntotal:                                     # vectors added before the clone travel with it (faiss copies them)
        g = index._materialize()[:, :index.d]
        out.add(g.float().contiguous())
    return out


def index_cpu_to_all_gpus(index: cbir.FlatIPIndex, co: Optional[GpuMultipleClonerOptions] = None, ngpu: int = -1) -> cbir.FlatIPIndex:
    """faiss.index_cpu_to_all_gpus(index, co) (evaluation.py:162).  faiss replicates the index on every visible GPU of the process and splits the QUERIES; one process per
    GPU is this library's model, so the clone lives on this process's device -- the results are the same list either way."""
    return _clone(index, co)


def index_cpu_to_gpu(res: StandardGpuResources, device: int, index: cbir.FlatIPIndex, co: Optional[GpuClonerOptions] = None) -> cbir.FlatIPIndex:
    """faiss.index_cpu_to_gpu(res, 0, index, co) (the commented alternative at evaluation.py:161)"""
    dev = None
    if index.be.device_only:
        dev = torch.device("cuda", int(device))
    return _clone(index, co, device=dev)


def index_gpu_to_cpu(index: cbir.FlatIPIndex) -> cbir.FlatIPIndex:
    """the way back: an fp32-storage index holding what the GPU index holds (fp16-stored rows come back as their fp16 values, as faiss returns them)"""
    return _clone(index, None)


def get_num_gpus() -> int:
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def normalize_L2(x: np.ndarray) -> None:
    """faiss.normalize_L2: in-place row normalisation of a float32 matrix (host side, like faiss's)"""
    if not isinstance(x, np.ndarray) or x.dtype != np.float32 or x.ndim != 2:
        raise TypeError("normalize_L2 expects a float32 [n, d] array")
    n = np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True))
    np.divide(x, np.maximum(n, 1e-30).astype(np.float32), out=x)
