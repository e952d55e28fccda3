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
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT
DEFAULT_CAP = 131072      # candidate-list capacity per query of the GUARANTEED schedule = rows per stage + k (csrc/cbir.hip).  Swept on the MI355X at 10 k x 1 M:
                          # round 6 (approximate ranking): 32 k 3.85 ms, 48 k 3.53, 64 k 3.40, 96 k 3.30, 128 k 3.23, 192 k 3.25 (fewer launches against looser
                          # thresholds per stage; round 3's exact ranking: 96 k 3.54, 128 k 3.53); with small_lists the lists hold SMALL_LIST_CAP entries
SMALL_LIST_CAP = 16384    # staged schedule with small candidate lists (`small_lists=True`): stages of cap - k rows like the guaranteed schedule, lists of this many entries
OPTIMISTIC_CAP = 8192     # ... of the optimistic schedule (bootstrap + two stages; ~10^3 survivors per query in a 10^6-row scan; overflow is detected and repaired)


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
                 idx_base: int = 0, method: str = "auto", storage: str = "float32", optimistic: bool = False, small_lists: bool = True,
                 approx_rank: bool = True):
        """method: "prefilter" = bf16-MFMA candidate filter with a rigorous error bound + exact fp32 re-scoring (d <= 512),
        "exact_scan" = every pair scored on the fp32 MFMA; "auto" picks prefilter when d <= 512.  Both return bit-identical
        results (tests/test_cbir.py runs every case through both).
        storage: "float32", or "float16" = what the reference's GPU index stores (faiss GpuClonerOptions.useFloat16 = True, engine/cbir/evaluation.py:157-162):
        the gallery is kept in fp16 (half the HBM footprint), queries are rounded to fp16 on arrival and every score is the fp32 k-ordered sum of the
        fp16 x fp16 products -- bit-identical to the fp32 index fed the fp16-rounded vectors (the oracle of tests/test_cbir.py); prefilter path only.
        optimistic: prefilter path: bootstrap + two stages with OPTIMISTIC_CAP-entry candidate lists (6 launches per search instead of 34 at 10^6 rows); a
        list overflow is reported by the kernels and the search is repeated with the guaranteed schedule (one device flag read per search).  Measured
        SLOWER than the default at 10 k x 1 M (4.5 vs 3.6 ms: the thresholds of a 16-stage scan rise stage by stage and leave ~10^3 survivors per query, two
        stages leave 2.6 x 10^3 and the ranking kernel's register merge no longer applies), so it is off by default; the default is the guaranteed
        sequential schedule (VDK_CBIR_PIPELINE=1 overlaps the ranking of stage i with the scan of stage i + 1 on a second stream: also measured slower)."""
        if d <= 0:
            raise ValueError("dimension must be positive")
        if method not in ("auto", "prefilter", "exact_scan"):
            raise ValueError("method must be auto | prefilter | exact_scan")
        if storage not in ("float32", "float16"):
            raise ValueError("storage must be float32 | float16")
        self.d = int(d)
        self.be = backend or _lib.load()
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda", torch.cuda.current_device()) if self.be.device_only else torch.device("cpu")
        self.cap = int(cap)
        self.idx_base = int(idx_base)
        self.storage = storage
        align = 8 if storage == "float16" else 4
        self._dp = (self.d + align - 1) // align * align     # kernels need d % 4 == 0 (8 for fp16 rows): zero-pad (adds exact zeros)
        if method == "prefilter" and self._dp > 512:
            raise ValueError("the prefilter path needs d <= 512")
        self.method = "exact_scan" if (method == "exact_scan" or self._dp > 512) else "prefilter"
        if storage == "float16" and self.method != "prefilter":
            raise ValueError("float16 storage is served by the prefilter path (d <= 512)")
        self.optimistic = bool(optimistic)
        # small_lists (default): the guaranteed schedule's stages (cap - k rows each) with candidate lists of SMALL_LIST_CAP entries instead of `cap`: the workspace is
        # nq * 16 384 * 8 B (1.3 GB at 10 k queries) instead of nq * cap * 8 B (7.9 GB).  A list that overflows is reported by the kernels and the search is repeated with
        # the guaranteed schedule, whose workspace is allocated only then (one device flag read per search).  Measured at 10 k x 1 M: 3.576 vs 3.572 ms in a back-to-back loop;
        # results bit-identical.  small_lists=False: the guaranteed schedule alone -- fully asynchronous (no host read, no data-dependent retry), 8 GB per 10 k queries.
        self.small_lists = bool(small_lists)
        # approx_rank (default, with small_lists, k <= 256): between the stages the survivors are ranked on the pre-filter's approximate scores and every row that can
        # still belong to the exact top-k is kept (k + a band of 2 eps); only the rows kept at the end of the scan get the exact fp32 chain -- ~130 gathered gallery rows
        # per query instead of ~650.  Bit-identical results (the kept set contains the exact top-k); a band wider than the kernel's 448 slots (masses of near-duplicate
        # rows) is reported like a list overflow and the search is repeated with the exact schedule.
        self.approx_rank = bool(approx_rank)
        self.fallbacks = 0                        # searches whose optimistic pass overflowed and were repeated with the guaranteed schedule
        self._gb: Optional[torch.Tensor] = None   # bf16 [N, DP] copy (DP = d rounded up to 128) + row-norm maxima, built once per gallery state
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
        x = self._to_dev(x)
        self._chunks.append(x.half() if self.storage == "float16" else x)

    def reset(self) -> None:
        self._chunks, self._gallery, self._gb, self._gmax = [], None, None, None

    def _materialize(self) -> torch.Tensor:
        if self._chunks:
            parts = ([self._gallery] if self._gallery is not None else []) + self._chunks
            self._gallery = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
            self._chunks = []
            self._gb = None
        if self._gallery is None:
            self._gallery = torch.empty((0, self._dp), dtype=torch.float16 if self.storage == "float16" else torch.float32, device=self.device)
        return self._gallery

    def _prepare(self) -> None:
        """add()-time work of the prefilter path: bf16 copy of the gallery and its largest row norm."""
        g = self._materialize()
        if self._gb is not None:
            return
        be, n = self.be, g.shape[0]
        wide = (self._dp + 127) // 128 * 128
        self._gb = torch.empty((max(n, 1), wide), dtype=torch.bfloat16, device=self.device)
        self._gmax = torch.zeros(4, dtype=torch.int32, device=self.device)
        if g.dtype == torch.float32 or n == 0:
            norms = torch.empty(3 * max(n, 1), dtype=torch.float32, device=self.device)
            be.check(be.lib.vdk_cbir_prepare_gallery(be.ptr(g) if n else None, n, self._dp, be.ptr(self._gb), be.ptr(norms),
                                                     be.ptr(self._gmax), be.stream()), "vdk_cbir_prepare_gallery")
            return
        # fp16 storage: the fp32 values the kernel wants exist only for one bounded block of rows at a time (a whole-gallery .float() would be twice the fp16
        # footprint again, exactly where fp16 was chosen to fit); the three norm maxima are positive floats, so their bit patterns combine with an integer max
        step = 1 << 16
        norms = torch.empty(3 * step, dtype=torch.float32, device=self.device)
        gm = torch.zeros(4, dtype=torch.int32, device=self.device)
        for r0 in range(0, n, step):
            blk = g[r0:r0 + step].float()
            be.check(be.lib.vdk_cbir_prepare_gallery(be.ptr(blk), blk.shape[0], self._dp, be.ptr(self._gb[r0:r0 + step]), be.ptr(norms), be.ptr(gm), be.stream()),
                     "vdk_cbir_prepare_gallery")
            self._gmax[:3] = torch.maximum(self._gmax[:3], gm[:3]) if r0 else gm[:3]
            self._gmax[3] = gm[3]

    def _workspace(self, nq: int, k: int, cap: int) -> torch.Tensor:
        need = C.c_size_t(0)
        if self.method == "prefilter":
            self.be.check(self.be.lib.vdk_cbir_fast2_workspace_bytes(nq, self._dp, k, cap, C.byref(need)), "vdk_cbir_fast2_workspace_bytes")
        else:
            self.be.check(self.be.lib.vdk_cbir_workspace_bytes(nq, k, cap, C.byref(need)), "vdk_cbir_workspace_bytes")
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
            be = self.be
            if self.method == "prefilter":
                self._prepare()
                if self.storage == "float16":
                    qd = qd.half().float()            # faiss multiplies fp16 x fp16 (fp32 accumulation): the queries are rounded like the stored rows
                from . import _abi
                gdt = _abi.F16_ if self.storage == "float16" else _abi.F32_

                def run(schedule: int, cap_: int, flag):
                    ws = self._workspace(nq, k, cap_)
                    be.check(be.lib.vdk_cbir_search_fast2(be.ptr(qd), nq, be.ptr(g), gdt, be.ptr(self._gb), be.ptr(self._gmax), g.shape[0], self._dp, k,
                                                          self.idx_base, be.ptr(scores), be.ptr(idx), cap_, schedule, be.ptr(flag), be.ptr(ws), ws.numel(),
                                                          be.stream()), "vdk_cbir_search_fast2")
                done = False
                if self.small_lists and not self.optimistic:
                    if getattr(self, "_flag", None) is None:
                        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                    approx = self.approx_rank and k <= 256
                    # |schedule| >= 1024 = stage length in rows; negative = approximate ranking, which carries up to 448 kept rows (k + the error band) between stages
                    stage_rows = max(1024, cap - (448 if approx else k))
                    run(-stage_rows if approx else stage_rows, max(min(cap, SMALL_LIST_CAP), 2 * k), self._flag)
                    done = int(self._flag.item()) == 0
                    if not done:
                        self.fallbacks += 1
                if self.optimistic:
                    if getattr(self, "_flag", None) is None:
                        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                    run(1, max(min(cap, OPTIMISTIC_CAP), 2 * k), self._flag)
                    done = int(self._flag.item()) == 0        # the one host read of the optimistic schedule
                    if not done:
                        self.fallbacks += 1
                if not done:
                    run(0, cap, None)
            else:
                ws = self._workspace(nq, k, cap)
                be.check(be.lib.vdk_cbir_search(be.ptr(qd), nq, be.ptr(g), g.shape[0], self._dp, k, self.idx_base,
                                                be.ptr(scores), be.ptr(idx), cap, be.ptr(ws), ws.numel(), be.stream()),
                         "vdk_cbir_search")
        if as_numpy:
            return scores.cpu().numpy(), idx.cpu().numpy()
        return scores, idx


def fp16_swap_report(scores32: np.ndarray, idx32: np.ndarray, idx16: np.ndarray) -> dict:
    """How an fp16-storage result differs from the fp32 one ("allow swaps only between neighbours whose fp32 scores differ by < 2^-10 and report
    the count"): positions whose index differs, how many of them are pure re-orderings inside the same top-k set, and the largest fp32 score gap a
    differing position spans (against the fp32 list's own score at that rank)."""
    diff = idx32 != idx16
    rows = np.nonzero(diff.any(1))[0]
    entered = 0
    worst = 0.0
    for r in rows:
        s32 = {int(i): float(s) for i, s in zip(idx32[r], scores32[r])}
        for pos in np.nonzero(diff[r])[0]:
            j = int(idx16[r, pos])
            if j in s32:
                worst = max(worst, abs(s32[j] - float(scores32[r, pos])))
            else:
                entered += 1                              # a row that was not in the fp32 top-k at all: it can only have been within rounding of the k-th score
                worst = max(worst, 0.0)
    return {"positions": int(diff.sum()), "rows_affected": int(len(rows)), "entered_from_outside_topk": int(entered), "max_fp32_gap_swapped": worst,
            "total_positions": int(idx32.size)}


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


def search_sharded(queries: torch.Tensor, local_gallery: torch.Tensor, k: int, idx_base: int, group=None, backend=None, device=None, cap: int = DEFAULT_CAP,
                   query_counts: Optional[Sequence[int]] = None, index: Optional["FlatIPIndex"] = None):
    """Multi-GPU search (path B): the gallery is row-sharded (`local_gallery` = this rank's rows, global row ids starting at
    `idx_base`), every rank holds its own query block.  Queries are all-gathered (one flat-tensor collective), each rank searches its shard, and every rank
    receives from every shard ONLY the top-k lists of its own queries (one all-to-all per tensor: 1/world of an all-gather's traffic over xGMI), merged with the
    same (score desc, index asc) rule -> the global (scores, indices) of ITS queries, bit-identical to a single-GPU search over the whole gallery.
    One process per GPU; the collectives stay outside the scan.  query_counts: per-rank query counts if the caller knows them (equal blocks: the usual case) --
    saves the count exchange and its host synchronisation; index: a prebuilt FlatIPIndex over this rank's shard (else built here)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = queries.device
    if query_counts is None:
        nq_local = torch.tensor([queries.shape[0]], dtype=torch.int64, device=dev)
        call = torch.zeros(world, dtype=torch.int64, device=dev)
        _all_gather_flat(dist, call, nq_local, group)
        counts = [int(c) for c in call.tolist()]
    else:
        counts = [int(c) for c in query_counts]
        assert len(counts) == world and counts[rank] == queries.shape[0]
    qmax, D = max(counts), queries.shape[1]
    if queries.shape[0] == qmax:
        qpad = queries.contiguous().float()
    else:
        qpad = torch.zeros((qmax, D), dtype=torch.float32, device=dev)
        qpad[:queries.shape[0]] = queries
    gathered = torch.empty((world, qmax, D), dtype=torch.float32, device=dev)
    _all_gather_flat(dist, gathered, qpad, group)
    allq = gathered.view(world * qmax, D) if min(counts) == qmax else torch.cat([gathered[r, :c] for r, c in enumerate(counts)], 0)
    if index is None:
        index = FlatIPIndex(D, backend=backend, device=device if device is not None else dev, cap=cap, idx_base=idx_base)
        index.add(local_gallery)
    s, i = index.search(allq, k)                                  # [sum nq, k] for this shard, global row ids
    s = s.contiguous(); i = i.contiguous()
    mine = counts[rank]
    try:        # shard r -> rank q: the rows of q's queries (all-to-all with uneven splits = counts)
        rs = torch.empty((world * mine, k), dtype=s.dtype, device=dev); ri = torch.empty((world * mine, k), dtype=i.dtype, device=dev)
        dist.all_to_all_single(rs, s, output_split_sizes=[mine] * world, input_split_sizes=counts, group=group)
        dist.all_to_all_single(ri, i, output_split_sizes=[mine] * world, input_split_sizes=counts, group=group)
        ps, pi = rs.view(world, mine, k), ri.view(world, mine, k)
    except (RuntimeError, NotImplementedError):   # a backend without all-to-all: gather everything, keep this rank's slice
        parts_s = [torch.empty_like(s) for _ in range(world)]; parts_i = [torch.empty_like(i) for _ in range(world)]
        dist.all_gather(parts_s, s, group=group); dist.all_gather(parts_i, i, group=group)
        lo = sum(counts[:rank]); hi = lo + mine
        ps = torch.stack([p[lo:hi] for p in parts_s]).contiguous(); pi = torch.stack([p[lo:hi] for p in parts_i]).contiguous()
    return merge_topk(ps, pi, backend=backend)


def _all_gather_flat(dist, out: torch.Tensor, inp: torch.Tensor, group) -> None:
    """all-gather into ONE preallocated tensor (no per-rank list, no concatenation copies); list form only where the backend lacks the flat collective"""
    try:
        dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1), group=group)
    except (RuntimeError, NotImplementedError, AttributeError):
        world = dist.get_world_size(group)
        parts = list(out.view(world, -1).unbind(0))
        dist.all_gather(parts, inp.contiguous().view(-1), group=group)
