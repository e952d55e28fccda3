"""Data-parallel exchange for hot path A: bucketed all-reduce of the FLAT gradient buffer, overlapped with backward.

The reference wraps the model in torch DDP (engine/vision_engine.py:313,510: 25 MB buckets, gradient mean over NCCL,
parameters broadcast from rank 0 at construction).  Here the gradients already live in one contiguous fp32 buffer in
reverse-execution order, so a bucket is a slice: `vdk_vit_backward` calls `on_grad_ready(offset, numel)` on the host as
soon as the kernels producing that slice are enqueued, and the slice's all-reduce is issued immediately
(`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  RCCL orders the collective
after the producing kernels (stream dependency on the launch stream) and runs it on its own stream, so it overlaps the
remaining backward kernels.  One process per GPU; sum here, the 1/world factor is folded into the optimizer kernel.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 32 << 20, always_communicate: bool = False):
        """always_communicate: issue every collective even in a one-rank group (where they are identities).  The GPU tests use it to drive the real
        stream-ordered RCCL path on a single MI355X (backend "nccl", world_size 1) and require bit-identical results to the communication-free step."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.active = self.world_size > 1 or always_communicate
        self.collectives = 0               # issued so far (tests / logs)
        self.bucket_bytes = bucket_bytes   # xGMI is per-link bound: fewer, larger collectives than DDP's 25 MB
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

    def _flush(self) -> None:
        if self._lo is None:
            return
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
        if (self._hi - self._lo) * 4 >= self.bucket_bytes:
            self._flush()

    def finish_step(self) -> None:
        if not self.active:
            return
        self._flush()
        for w in self._pending:
            w.wait()          # makes the launch stream wait for the collective (no host sync on NCCL/RCCL)
        self._pending = []
