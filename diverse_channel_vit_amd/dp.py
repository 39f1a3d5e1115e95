"""Data-parallel gradient exchange for the DiChaViT hot path: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" IS RCCL on ROCm).  Replaces torch DDP's reducer
(trainer.py:1185) for this model.

The encoder's gradients are produced by the hand-written backward into ONE flat fp32 arena, last
layer first.  As soon as a layer's slice is complete the model calls ``grad_ready(arena, lo, hi)``
and the slice is all-reduced asynchronously on RCCL's stream while the backward of the earlier
layers keeps the compute stream busy (14 contiguous buckets of <= 7 MB for DiChaViT-S; no gather /
scatter copies, no per-parameter hooks).  The few small parameters outside the encoder node (head,
channel embeddings, positional table) are reduced from post-accumulate hooks.  ``finalize()`` makes
the compute stream wait for every outstanding bucket; HipAdamW calls it before the update.

Parameters that received no gradient this step (``proxies`` in cross-entropy mode, dichavit.py:803-805)
take no part — the reference needs ``find_unused_parameters=True`` for the same reason.
Ranks must draw the same HCS channel subset (same seed on every rank): sequence lengths
then match and the step stays balanced."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class DataParallel:
    #: CUs the backward leaves to RCCL's kernels (dichavit.py, _run_backward); matches the channel cap `limit_rccl_channels` sets
    reserved_cus = 8

    @staticmethod
    def limit_rccl_channels(n: int = 8) -> None:
        """Call BEFORE torch.distributed.init_process_group: caps RCCL at `n` channels (= workgroups = CUs held during an
        all-reduce) unless the user already chose.  86 MB of gradients per ~40 ms step need a small fraction of the xGMI
        bandwidth, while every CU RCCL holds stalls a persistent GEMM workgroup."""
        import os
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(n, 4)))

    def __init__(self, model, process_group=None, min_bucket_bytes: int = 4 << 20, force_collectives: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        self._avg = self.backend == "nccl"  # RCCL has a native AVG; gloo sums and we scale
        self._force = force_collectives     # issue the collectives even at world_size 1 (single-GPU plumbing tests)
        self._works: List = []
        self._scaled: List[torch.Tensor] = []
        self._pending = None  # (arena, lo, hi) waiting to be merged into a bucket of >= min_bucket_bytes
        self.min_bucket = min_bucket_bytes // 4
        self.buckets_launched = 0
        model._dp = self
        self._hooks = []
        self._hooked = set()

    # ---- encoder arena: called by the model's backward in reverse layer order ---------------------
    def grad_ready(self, arena: torch.Tensor, lo: int, hi: int) -> None:
        if self._pending is not None and self._pending[0] is arena and hi == self._pending[1]:
            lo, hi = lo, self._pending[2]  # adjacent (earlier layer sits just below): merge
            self._pending = (arena, lo, hi)
        else:
            self._launch_pending()
            self._pending = (arena, lo, hi)
        if self._pending[2] - self._pending[1] >= self.min_bucket:
            self._launch_pending()

    def _launch_pending(self) -> None:
        if self._pending is None:
            return
        arena, lo, hi = self._pending
        self._pending = None
        self._reduce(arena[lo:hi])

    def _reduce(self, t: torch.Tensor) -> None:
        if self.world == 1 and not self._force:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append(dist.all_reduce(t, op=op, group=self.group, async_op=True))
        if not self._avg:
            self._scaled.append(t)
        self.buckets_launched += 1

    def flush(self) -> None:
        """End of the encoder backward: launch whatever is still being merged."""
        self._launch_pending()

    # ---- small parameters outside the encoder node --------------------------------------------------
    def hook_misc_params(self) -> None:
        """All-reduce the gradient of every parameter that is not part of the encoder arena as soon as
        autograd has accumulated it."""
        enc = {id(p) for p in getattr(self.model, "_enc_params", [])}
        for p in self.model.parameters():
            if id(p) in enc or id(p) in self._hooked or not p.requires_grad:
                continue
            self._hooked.add(id(p))
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda q: self._reduce(q.grad)))

    def finalize(self) -> None:
        """Make the current stream wait for every outstanding all-reduce (call before optimizer.step)."""
        self.flush()
        for w in self._works:
            w.wait()
        self._works.clear()
        if self._scaled:
            inv = 1.0 / self.world
            for t in self._scaled:
                t.mul_(inv)
            self._scaled.clear()

    def broadcast_parameters(self, src: int = 0) -> None:
        """Same initial weights on every rank (DDP does this in its constructor)."""
        arena = getattr(self.model, "_arena", None)
        if arena is not None:
            dist.broadcast(arena, src=src, group=self.group)
        else:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=src, group=self.group)
