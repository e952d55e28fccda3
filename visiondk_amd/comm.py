"""Data-parallel exchange for hot path A: bucketed all-reduce of the FLAT gradient buffer, overlapped with backward.

The reference wraps the model in torch DDP (engine/vision_engine.py:313,510: 25 MB buckets, gradient mean over NCCL,
parameters broadcast from rank 0 at construction).  Here the gradients already live in one contiguous fp32 buffer in
reverse-execution order, so a bucket is a slice: `vdk_vit_backward` calls `on_grad_ready(offset, numel)` on the host as
soon as the kernels producing that slice are enqueued, and the slice's all-reduce is issued immediately
(`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  RCCL orders the collective
after the producing kernels (stream dependency on the launch stream) and runs it on its own stream, so it overlaps the
remaining backward kernels.  One process per GPU; sum here, the 1/world factor is folded into the optimizer kernel.

What the optimizer waits for: clip_grad_norm_ needs the norm of the WHOLE reduced gradient (train.py:203-206), so the clipped SGD kernel cannot start on a
reduced bucket while a later one is in flight -- only the LAST bucket's collective is exposed (everything before it runs under the backward).  The bucket rule
keeps that last one small: a bucket closes when it reaches `bucket_bytes` or when no more than its own size is still to come, so the gradient's tail (for the
ViT: block 0, then the embeddings) leaves in pieces of decreasing size instead of one 2-block bucket at the very end.  c10d is the binding to RCCL here on
purpose ("collectives"): the C-ABI exposes the ready-range callback, the exchange itself is the host framework's.

route="abi" sends the same buckets through the library's own collectives instead (csrc/comm.hip: vdk_comm_init / vdk_allreduce_bucket / vdk_comm_finish -- RCCL on the
communicator's own stream, ordered by events): what a C / C++ host without a process group uses, driven here so that both routes run the same tests.  The unique id travels
over the torch process group once, at construction.  `trace=True` arms the timing trace (vdk_comm_trace): where every all-reduce sits in time relative to the backward.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 24 << 20, always_communicate: bool = False, reserve_cus: int = 32,
                 route: str = "c10d", trace: bool = False, standin_us_per_mb: float = 0.0, standin_cus: Optional[int] = None):
        """route: "c10d" (torch.distributed issues the all-reduce) or "abi" (vdk_allreduce_bucket on the library's communicator; GPU only).
        trace (route "abi"): record where each all-reduce runs in time (read with trace_read()).  standin_us_per_mb (route "abi", diagnostics on ONE GPU, where a collective
        over a single rank moves nothing): behind every all-reduce a kernel that holds `reserve_cus` CUs for standin_us_per_mb microseconds per MB of the bucket runs on the
        collectives' stream -- the footprint of an 8-GPU ring all-reduce of that bucket -- so that the exposure of the exchange can be measured without a second GPU.
        reserve_cus: CUs the persistent GEMM grids leave to the collectives' kernels while this object is active on a GPU (vdk_gemm_reserve_cus: an all-reduce in
        flight holds one CU per channel; a persistent GEMM whose static tile walk covers every CU would wait for it to END -- measured with a stand-in kernel,
        tools/w4_contention.py).  always_communicate: issue every collective even in a one-rank group (where they are identities).  The GPU tests use it to drive the real
        stream-ordered RCCL path on a single MI355X (backend "nccl", world_size 1) and require bit-identical results to the communication-free step."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.active = self.world_size > 1 or always_communicate
        # The reserve is scoped to the window in which a collective can actually be in flight: it is set when the step's FIRST bucket is issued (from inside the backward)
        # and the previous value is restored in finish_step().  Forward, loss, optimizer -- and anything else in the process: evaluation, CBIR, another model -- run
        # their persistent GEMM grids on every CU.  (Round 3 set it process-wide in the constructor and never restored it.)
        self._reserve = int(reserve_cus) if (self.active and torch.cuda.is_available()) else 0
        self._be = None
        self._reserve_prev: Optional[int] = None
        if self._reserve:
            from . import _lib
            self._be = _lib.load()
        if route not in ("c10d", "abi"):
            raise ValueError("route must be 'c10d' or 'abi'")
        self.route = route
        self._comm = None
        self._trace = bool(trace)
        self._standin = float(standin_us_per_mb)
        self._standin_cus = standin_cus
        if route == "abi" and self.active:
            import ctypes as C
            from . import _lib
            if not torch.cuda.is_available():
                raise RuntimeError("route='abi' drives RCCL through the C ABI: GPU only (the CPU tests use c10d / gloo)")
            self._be = self._be or _lib.load()
            be = self._be
            uid = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_ubyte * 128)()
                be.check(be.lib.vdk_comm_unique_id(buf), "vdk_comm_unique_id")
                uid = torch.tensor(list(buf), dtype=torch.uint8)
            if self.world_size > 1:      # the 128 bytes travel over the host's process group (any backend)
                t = uid.cuda() if dist.get_backend(group) == "nccl" else uid
                dist.broadcast(t, src=0, group=group)
                uid = t.cpu()
            raw = (C.c_ubyte * 128)(*uid.tolist())
            handle = C.c_void_p()
            be.check(be.lib.vdk_comm_init(raw, self.rank, self.world_size, C.byref(handle)), "vdk_comm_init")
            self._comm = handle
            if self._trace:
                be.check(be.lib.vdk_comm_trace(self._comm, 1), "vdk_comm_trace")
        elif trace or standin_us_per_mb:
            raise ValueError("trace / standin_us_per_mb belong to route='abi'")
        self.collectives = 0               # issued so far (tests / logs)
        self.bucket_bytes = bucket_bytes   # ViT-B: one 28 MB transformer block per collective (ranges arrive per block); far above the size where a ring over xGMI is latency-bound
        self._grads: Optional[torch.Tensor] = None
        self._pending: List = []
        self._lo = self._hi = None

    def broadcast_params(self, flat_params: torch.Tensor, src: int = 0, engine=None) -> None:
        """DDP-constructor semantics (C2): every rank starts from rank `src`'s weights.  c10d collectives do not bump tensor versions, so the engine whose
        bf16 operand copies derive from `flat_params` is told explicitly that they are stale (a forward may have run before the step was constructed)."""
        dist.broadcast(flat_params, src=src, group=self.group)
        self.collectives += 1
        if engine is not None:
            engine._weights_version = None

    # ---- per step -----------------------------------------------------------------------------------
    def begin_step(self, flat_grads: torch.Tensor) -> None:
        self._grads = flat_grads
        self._pending = []
        self._lo = self._hi = None
        self._sent_lo = flat_grads.numel()      # lower end of what has been issued so far: ranges that arrive top-down keep extending directly below it
        if self._comm is not None and self._trace:
            self._be.check(self._be.lib.vdk_comm_trace(self._comm, 1), "vdk_comm_trace")          # a fresh trace per step; mark 0 = the backward is about to be enqueued
            self._be.check(self._be.lib.vdk_comm_mark(self._comm, self._be.stream()), "vdk_comm_mark")

    def _hold_cus(self) -> None:
        if self._reserve and self._reserve_prev is None:
            self._reserve_prev = int(self._be.lib.vdk_gemm_reserved_cus())
            self._be.check(self._be.lib.vdk_gemm_reserve_cus(max(self._reserve, self._reserve_prev)), "vdk_gemm_reserve_cus")

    def _release_cus(self) -> None:
        if self._reserve_prev is not None:
            self._be.check(self._be.lib.vdk_gemm_reserve_cus(self._reserve_prev), "vdk_gemm_reserve_cus")
            self._reserve_prev = None

    def close(self) -> None:
        """give the CUs back if a step was abandoned between begin_step and finish_step; destroy the library's communicator (route "abi")"""
        self._release_cus()
        if self._comm is not None:
            self._be.lib.vdk_comm_destroy(self._comm)
            self._comm = None

    def trace_read(self) -> dict:
        """route "abi" with trace=True, after a step: {"marks_ms": [0, backward enqueued and executed up to here, ...], "allreduce_ms": [(start, end), ...], "numel": [...]}
        in milliseconds after begin_step's mark (synchronises)."""
        import ctypes as C
        be = self._be
        n_ar, n_m = C.c_int32(0), C.c_int32(0)
        ar = (C.c_float * 512)(); numel = (C.c_int64 * 256)(); marks = (C.c_float * 64)()
        be.check(be.lib.vdk_comm_trace_read(self._comm, ar, numel, 256, C.byref(n_ar), marks, 64, C.byref(n_m)), "vdk_comm_trace_read")
        na, nm = min(n_ar.value, 256), min(n_m.value, 64)      # the library reports the RECORDED counts; the buffers above hold 256 collectives / 64 marks
        return {"marks_ms": [marks[i] for i in range(nm)], "allreduce_ms": [(ar[2 * i], ar[2 * i + 1]) for i in range(na)],
                "numel": [numel[i] for i in range(na)], "recorded": (n_ar.value, n_m.value)}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _flush(self) -> None:
        if self._lo is None:
            return
        self._hold_cus()                      # from here until finish_step a collective may be in flight beside the remaining backward GEMMs
        self._sent_lo = min(self._sent_lo, self._lo)
        if self._comm is not None:
            be = self._be
            n = self._hi - self._lo
            be.check(be.lib.vdk_allreduce_bucket(self._comm, be.ptr(self._grads), self._lo, n, be.stream()), "vdk_allreduce_bucket")
            if self._standin > 0.0:      # one GPU: a kernel with the footprint of the multi-GPU collective (CUs held for the time the ring would take) behind the no-op all-reduce
                us = int(self._standin * n * 4 / 1e6)
                if us > 0:
                    be.check(be.lib.vdk_debug_occupy_cus(self._standin_cus if self._standin_cus else max(self._reserve, 1), us, be.lib.vdk_comm_stream(self._comm)), "vdk_debug_occupy_cus")
            if self._trace:
                be.check(be.lib.vdk_comm_trace_close_last(self._comm), "vdk_comm_trace_close_last")
        else:
            sl = self._grads[self._lo:self._hi]
            self._pending.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.collectives += 1
        self._lo = self._hi = None

    def on_grad_ready(self, offset: int, numel: int) -> None:
        """Called from inside vdk_vit_backward, ranges arrive in descending, contiguous order."""
        if not self.active:
            return
        if self._lo is None:
            self._lo, self._hi = offset, offset + numel
        elif offset + numel == self._lo:
            self._lo = offset
        else:  # non-adjacent: close the current bucket first
            self._flush()
            self._lo, self._hi = offset, offset + numel
        size = self._hi - self._lo
        # full -- or (top-down arrival only: this bucket sits directly below everything issued so far, so what is still to come are the offsets below _lo) the rest of
        # the gradient is no bigger than this bucket.  For any other arrival order _lo says nothing about what is left and only the size rule applies.
        top_down = self._hi == self._sent_lo
        if size * 4 >= self.bucket_bytes or (top_down and size >= self._lo):
            self._flush()

    def finish_step(self) -> None:
        if not self.active:
            return
        self._flush()
        if self._comm is not None:
            if self._trace:      # mark 1: everything the backward enqueued has executed up to here
                self._be.check(self._be.lib.vdk_comm_mark(self._comm, self._be.stream()), "vdk_comm_mark")
            self._be.check(self._be.lib.vdk_comm_finish(self._comm, self._be.stream()), "vdk_comm_finish")      # the launch stream waits on the device
            if self._trace:      # mark 2: ... and every collective has landed (what the clip / optimizer kernels wait for)
                self._be.check(self._be.lib.vdk_comm_mark(self._comm, self._be.stream()), "vdk_comm_mark")
        for w in self._pending:
            w.wait()          # makes the launch stream wait for the collective (no host sync on NCCL/RCCL)
        self._pending = []
        self._release_cus()   # everything enqueued from here on runs after the last collective: the persistent GEMMs get the whole chip back
