"""Hot path B host side: a faiss-like inner-product index on MI355X, and the reference's index()/search().

Mirrors the retrieval seam of the reference (SURVEY.md §8(b) "Retrieval index protocol"):

    faiss_index = faiss.index_factory(dim, "Flat", faiss.METRIC_INNER_PRODUCT)   # cbir/evaluation.py:155
    faiss_index.train(x); faiss_index.add(x)                                     # :167-168
    D, I = faiss_index.search(q.astype(np.float32), k=k)                         # :193

`index_factory(dim, "Flat", METRIC_INNER_PRODUCT)` returns a `FlatIPIndex` with the same methods; numpy in /
numpy out like faiss, or device tensors in / device tensors out to skip the host round trip.  All arithmetic
happens in the HIP kernels of csrc/cbir.hip through the C ABI; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT
DEFAULT_CAP = 65536       # candidate-list capacity per query (bounds the stage size, see csrc/cbir.hip)


def l2_normalize(x: torch.Tensor, eps: float = 1e-12, backend: Optional[_lib.Backend] = None) -> torch.Tensor:
    """F.normalize(x, p=2, dim=1, eps) of face_model.py:139 on float32 rows."""
    be = backend or _lib.load()
    if x.dtype != torch.float32 or x.dim() != 2:
        raise ValueError("l2_normalize expects a float32 [n, d] tensor")
    x = x.contiguous()
    out = torch.empty_like(x)
    be.check(be.lib.vdk_l2norm_rows(be.ptr(x), be.ptr(out), x.shape[0], x.shape[1], eps, be.stream()), "vdk_l2norm_rows")
    return out


class FlatIPIndex:
    """Exact inner-product index (faiss IndexFlatIP semantics; ties -> lower index; pads (-FLT_MAX, -1))."""

    def __init__(self, d: int, backend: Optional[_lib.Backend] = None, device=None, cap: int = DEFAULT_CAP,
                 idx_base: int = 0, method: str = "auto"):
        """method: "prefilter" = bf16-MFMA candidate filter with a rigorous error bound + exact fp32 re-scoring (d <= 128),
        "exact_scan" = every pair scored on the fp32 MFMA; "auto" picks prefilter when d <= 128.  Both return bit-identical
        results (tests/test_cbir.py runs every case through both)."""
        if d <= 0:
            raise ValueError("dimension must be positive")
        if method not in ("auto", "prefilter", "exact_scan"):
            raise ValueError("method must be auto | prefilter | exact_scan")
        self.d = int(d)
        self.be = backend or _lib.load()
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda", torch.cuda.current_device()) if self.be.device_only else torch.device("cpu")
        self.cap = int(cap)
        self.idx_base = int(idx_base)
        self._dp = (self.d + 3) // 4 * 4          # kernels need d % 4 == 0: zero-pad (adds exact zeros)
        if method == "prefilter" and self._dp > 128:
            raise ValueError("the prefilter path needs d <= 128")
        self.method = "exact_scan" if (method == "exact_scan" or self._dp > 128) else "prefilter"
        self._gb: Optional[torch.Tensor] = None   # bf16 [N, 128] copy + max row norm, built once per gallery state
        self._gmax: Optional[torch.Tensor] = None
        self._chunks: list[torch.Tensor] = []
        self._gallery: Optional[torch.Tensor] = None
        self.is_trained = True
        self._ws: Optional[torch.Tensor] = None

    # ---- faiss surface ------------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return sum(c.shape[0] for c in self._chunks) + (0 if self._gallery is None else self._gallery.shape[0])

    def train(self, x) -> None:  # IndexFlat needs no training (cbir/evaluation.py:167)
        return None

    def _to_dev(self, x) -> torch.Tensor:
        if isinstance(x, np.ndarray):
            if x.dtype != np.float32:
                raise TypeError("FlatIPIndex only accepts float32 (like faiss)")
            x = torch.from_numpy(np.ascontiguousarray(x))
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected float32 [n, {self.d}]")
        x = x.to(self.device, non_blocking=True)
        if self._dp != self.d:
            x = torch.nn.functional.pad(x, (0, self._dp - self.d))
        return x.contiguous()

    def add(self, x) -> None:
        self._chunks.append(self._to_dev(x))

    def reset(self) -> None:
        self._chunks, self._gallery, self._gb, self._gmax = [], None, None, None

    def _materialize(self) -> torch.Tensor:
        if self._chunks:
            parts = ([self._gallery] if self._gallery is not None else []) + self._chunks
            self._gallery = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
            self._chunks = []
            self._gb = None
        if self._gallery is None:
            self._gallery = torch.empty((0, self._dp), dtype=torch.float32, device=self.device)
        return self._gallery

    def _prepare(self) -> None:
        """add()-time work of the prefilter path: bf16 copy of the gallery and its largest row norm."""
        g = self._materialize()
        if self._gb is not None:
            return
        be, n = self.be, g.shape[0]
        self._gb = torch.empty((max(n, 1), 128), dtype=torch.bfloat16, device=self.device)
        self._gmax = torch.zeros(4, dtype=torch.int32, device=self.device)
        norms = torch.empty(3 * max(n, 1), dtype=torch.float32, device=self.device)
        be.check(be.lib.vdk_cbir_prepare_gallery(be.ptr(g) if n else None, n, self._dp, be.ptr(self._gb), be.ptr(norms),
                                                 be.ptr(self._gmax), be.stream()), "vdk_cbir_prepare_gallery")

    def _workspace(self, nq: int, k: int, cap: int) -> torch.Tensor:
        need = C.c_size_t(0)
        fn = self.be.lib.vdk_cbir_fast_workspace_bytes if self.method == "prefilter" else self.be.lib.vdk_cbir_workspace_bytes
        self.be.check(fn(nq, k, cap, C.byref(need)), "vdk_cbir_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def search(self, q, k: int):
        """D [b,k] float32 descending, I [b,k] int64.  numpy in -> numpy out; tensor in -> tensor out."""
        as_numpy = isinstance(q, np.ndarray)
        qd = self._to_dev(q)
        g = self._materialize()
        nq = qd.shape[0]
        if not 1 <= k <= 1024:
            raise ValueError("1 <= k <= 1024")
        cap = max(self.cap, 2 * k)
        scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if nq:
            ws = self._workspace(nq, k, cap)
            be = self.be
            if self.method == "prefilter":
                self._prepare()
                be.check(be.lib.vdk_cbir_search_fast(be.ptr(qd), nq, be.ptr(g), be.ptr(self._gb), be.ptr(self._gmax), g.shape[0],
                                                     self._dp, k, self.idx_base, be.ptr(scores), be.ptr(idx), cap, be.ptr(ws),
                                                     ws.numel(), be.stream()), "vdk_cbir_search_fast")
            else:
                be.check(be.lib.vdk_cbir_search(be.ptr(qd), nq, be.ptr(g), g.shape[0], self._dp, k, self.idx_base,
                                                be.ptr(scores), be.ptr(idx), cap, be.ptr(ws), ws.numel(), be.stream()),
                         "vdk_cbir_search")
        if as_numpy:
            return scores.cpu().numpy(), idx.cpu().numpy()
        return scores, idx


def index_factory(d: int, description: str = "Flat", metric: int = METRIC_INNER_PRODUCT, **kw) -> FlatIPIndex:
    """faiss.index_factory(dim, "Flat", faiss.METRIC_INNER_PRODUCT) of cbir/evaluation.py:155."""
    if description != "Flat" or metric != METRIC_INNER_PRODUCT:
        raise NotImplementedError("the reference only builds (\"Flat\", METRIC_INNER_PRODUCT) indexes")
    return FlatIPIndex(d, **kw)


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, backend: Optional[_lib.Backend] = None):
    """Merge per-shard results [S, nq, k] (idx < 0 = empty) into the global top-k, bit-identical to one search."""
    be = backend or _lib.load()
    S, nq, k = scores.shape
    scores, idx = scores.contiguous(), idx.contiguous()
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    cap = max(S * k, 2 * k)
    need = C.c_size_t(0)
    be.check(be.lib.vdk_cbir_workspace_bytes(nq, k, cap, C.byref(need)), "vdk_cbir_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=scores.device)
    be.check(be.lib.vdk_cbir_merge_topk(be.ptr(scores), be.ptr(idx), S, nq, k, be.ptr(out_s), be.ptr(out_i),
                                        be.ptr(ws), ws.numel(), be.stream()), "vdk_cbir_merge_topk")
    return out_s, out_i


# ---- the reference's two functions, same names / argument meaning (engine/cbir/evaluation.py:106,171) ----
def index(extractor, gallery_dataloader, device, logger=None, index_factory: str = "Flat", memmap_feat_dim: Optional[int] = None,
          memmap_dtype=np.float16, memmap_save_path: Optional[str] = None, memmap_load_embedding: bool = False, gallery_embeddings=None,
          **kw) -> FlatIPIndex:
    """engine/cbir/evaluation.py:104-169, same arguments: encode the gallery (extractor.extract_cbir) or load it from a memmap file
    (`memmap_load_embedding`, dtype `memmap_dtype` — fp16 by default like the reference — reshaped to [-1, memmap_feat_dim]); optionally save the
    freshly extracted embeddings to `memmap_save_path` in their own dtype, in batches of 10 000 rows; build the inner-product index; add the
    embeddings as float32 ("faiss only accepts float32", :165).  `gallery_embeddings` may also be passed directly (device tensor or ndarray)."""
    if gallery_embeddings is None:
        if memmap_load_embedding:
            if memmap_save_path is None or memmap_feat_dim is None:
                raise ValueError("memmap_load_embedding needs memmap_save_path and memmap_feat_dim")
            dt = np.dtype(str(memmap_dtype).replace("torch.", "")) if not isinstance(memmap_dtype, (type, np.dtype)) else np.dtype(memmap_dtype)
            gallery_embeddings = np.memmap(memmap_save_path, mode="r", dtype=dt).reshape(-1, memmap_feat_dim)
        else:
            gallery_embeddings = extractor.extract_cbir(gallery_dataloader, device)
            if memmap_save_path is not None:
                if logger is not None:
                    logger.console(f"saving embeddings at {memmap_save_path}...")
                mm = np.memmap(memmap_save_path, shape=gallery_embeddings.shape, mode="w+", dtype=gallery_embeddings.dtype)
                for i in range(0, gallery_embeddings.shape[0], 10000):
                    mm[i:i + 10000] = gallery_embeddings[i:i + 10000]
                mm.flush()
    if isinstance(gallery_embeddings, np.ndarray):
        gallery_embeddings = np.ascontiguousarray(gallery_embeddings.astype(np.float32))
    dim = gallery_embeddings.shape[-1]
    idx = globals()["index_factory"](dim, index_factory, METRIC_INNER_PRODUCT, device=device, **kw) if isinstance(index_factory, str) else index_factory
    if logger is not None:
        logger.console("Adding embeddings...")
    idx.train(gallery_embeddings)
    idx.add(gallery_embeddings)
    return idx


def search(extractor, query_dataloader, faiss_index: FlatIPIndex, device, logger=None, k: int = 100,
           batch_size: int = 256, query_embeddings=None):
    """Reference search(): query batches of `batch_size` -> concatenated (scores, indices)."""
    if query_embeddings is None:
        query_embeddings = extractor.extract_cbir(query_dataloader, device)
    n = query_embeddings.shape[0]
    if logger is not None:
        logger.console("Searching ...")
    all_s, all_i = [], []
    # The reference walks over the queries in batches of `batch_size` (256) to bound faiss's temporary memory; every query's result is independent of the
    # batching, and a 256-query launch leaves most of the GPU idle, so the device calls here take up to 16384 queries (one H2D / D2H per chunk).
    batch_size = max(batch_size, 16384)
    for i in range(0, n, batch_size):
        q = query_embeddings[i:min(i + batch_size, n)]
        if isinstance(q, np.ndarray):
            q = np.asarray(q, dtype=np.float32)          # no copy when the embeddings already are float32
        s, ind = faiss_index.search(q, k=k)
        all_s.append(s)
        all_i.append(ind)
    if len(all_s) == 1:
        return all_s[0], all_i[0]
    if isinstance(all_s[0], np.ndarray):
        return np.concatenate(all_s, 0), np.concatenate(all_i, 0)
    return torch.cat(all_s, 0), torch.cat(all_i, 0)


def search_sharded(queries: torch.Tensor, local_gallery: torch.Tensor, k: int, idx_base: int, group=None, backend=None, device=None, cap: int = DEFAULT_CAP):
    """Multi-GPU search (path B): the gallery is row-sharded (`local_gallery` = this rank's rows, global row ids starting at
    `idx_base`), every rank holds its own query block.  Queries are all-gathered, each rank searches its shard, the per-shard top-k lists are
    all-gathered and merged with the same (score desc, index asc) rule -> every rank returns the global (scores, indices) of ITS queries,
    bit-identical to a single-GPU search over the whole gallery.  One process per GPU; two collectives, both outside the scan."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nq_local = torch.tensor([queries.shape[0]], dtype=torch.int64, device=queries.device)
    counts = [torch.zeros_like(nq_local) for _ in range(world)]
    dist.all_gather(counts, nq_local, group=group)
    counts = [int(c.item()) for c in counts]
    qmax = max(counts)
    qpad = torch.zeros((qmax, queries.shape[1]), dtype=torch.float32, device=queries.device)
    qpad[:queries.shape[0]] = queries
    gathered = [torch.empty_like(qpad) for _ in range(world)]
    dist.all_gather(gathered, qpad, group=group)
    allq = torch.cat([g[:c] for g, c in zip(gathered, counts)], 0)
    index = FlatIPIndex(queries.shape[1], backend=backend, device=device if device is not None else queries.device, cap=cap, idx_base=idx_base)
    index.add(local_gallery)
    s, i = index.search(allq, k)                                  # [sum nq, k] for this shard, global row ids
    parts_s = [torch.empty_like(s) for _ in range(world)]; parts_i = [torch.empty_like(i) for _ in range(world)]
    dist.all_gather(parts_s, s.contiguous(), group=group); dist.all_gather(parts_i, i.contiguous(), group=group)
    lo = sum(counts[:rank]); hi = lo + counts[rank]
    return merge_topk(torch.stack([p[lo:hi] for p in parts_s]).contiguous(), torch.stack([p[lo:hi] for p in parts_i]).contiguous(), backend=backend)
