"""Data-parallel gradient exchange for the DiChaViT hot path: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" IS RCCL on ROCm).  Replaces torch DDP's reducer
(trainer.py:1185) for this model.

The encoder's gradients are produced by the hand-written backward into ONE flat fp32 arena, last
layer first.  As soon as a layer's slice is complete the model calls ``grad_ready(arena, lo, hi)``
and the slice is all-reduced asynchronously on RCCL's stream while the backward of the earlier
layers keeps the compute stream busy (14 contiguous buckets of <= 7 MB for DiChaViT-S; no gather /
scatter copies, no per-parameter hooks).  The few small parameters outside the encoder node (head,
channel embeddings, positional table) are reduced from post-accumulate hooks.

The backward is SELF-SYNCHRONISING (as torch DDP's is): the model queues ``finalize()`` as an
end-of-backward callback of the autograd engine, so whatever runs after ``loss.backward()`` — HipAdamW,
torch/timm AdamW, ``torch.nn.utils.clip_grad_norm_``, ``scaler.unscale_`` — sees fully reduced gradients.
With gradient accumulation (several backward passes per optimiser step, the CHAMMI step of
trainer.py:846-935) the second and later passes wait for their own buckets before autograd adds them to
``.grad``; the sum of per-pass averages equals the average of the sums.

Parameters that received no gradient this step (``proxies`` in cross-entropy mode, dichavit.py:803-805)
take no part — the reference needs ``find_unused_parameters=True`` for the same reason.
Ranks must draw the same HCS channel subset (same seed on every rank): sequence lengths
then match and the step stays balanced."""
from __future__ import annotations

import os
import re
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
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(n, 4)))

    @staticmethod
    def rccl_channels_from_log(path: str) -> Optional[int]:
        """Number of channels RCCL actually set up, parsed from an ``NCCL_DEBUG=INFO`` log (``NCCL_DEBUG_FILE``): the
        ``Channel 03/08 :`` ring lines or the ``N coll channels`` summary.  None when the log has neither."""
        try:
            with open(path, errors="replace") as f:
                txt = f.read()
        except OSError:
            return None
        m = re.findall(r"(\d+) coll channels", txt)
        if m:
            return int(m[-1])
        m = re.findall(r"Channel \d+/(\d+)\s*:", txt)
        if m:
            return int(m[-1])
        return None

    def __init__(self, model, process_group=None, min_bucket_bytes: int = 4 << 20, force_collectives: bool = False,
                 grad_dtype: torch.dtype = torch.float32, overlap: bool = True, dist_module=None):
        """grad_dtype=torch.bfloat16 exchanges a bf16 copy of every bucket (43 MB instead of 86 MB per step for DiChaViT-S;
        the fp32 arena is overwritten with the averaged bf16 values).  overlap=False issues ONE all-reduce of everything
        that became ready, after the backward (no collective runs beside the backward's kernels).
        dist_module: the object the collectives are called on (default ``torch.distributed``); tests pass one that keeps RCCL's stream
        semantics — asynchronous on a stream of its own, ordered after the issuing stream's position — on a box where RCCL cannot run
        two ranks (tests/test_model_gpu.py, test_dp_async_collectives_emulated_on_one_gpu)."""
        dist = self._dist = dist_module if dist_module is not None else globals()["dist"]
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype: float32 or bfloat16")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        self._avg = self.backend == "nccl"  # RCCL has a native AVG; gloo sums and we scale
        self._force = force_collectives     # issue the collectives even at world_size 1 (single-GPU plumbing tests)
        self._works: List = []              # (work or None, reduced buffer, fp32 destination or None)
        self._pending = None  # (arena, lo, hi) waiting to be merged into a bucket of >= min_bucket_bytes
        self._deferred: List[torch.Tensor] = []  # overlap=False: slices waiting for finalize()
        self.min_bucket = min_bucket_bytes // 4
        self.grad_dtype = grad_dtype
        self.overlap = overlap
        self.buckets_launched = 0
        self.bytes_reduced = 0
        self._callback_queued = False
        self._poisoned = None       # set when a backward pass failed at world > 1: the ranks' collective sequences no longer match
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
        if not self.overlap:
            self._deferred.append(t)
            return
        self._all_reduce(t)

    def _all_reduce(self, t: torch.Tensor) -> None:
        buf, dst = t, None
        if self.grad_dtype != t.dtype:
            buf, dst = t.to(self.grad_dtype), t  # bf16 copy travels; the average is written back in finalize()
        if self.backend == "gloo" and buf.is_cuda:
            # gloo (tests on a one-GPU box) reduces host memory: stage through the CPU, synchronously
            h = buf.float().cpu()
            self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.group)
            t.copy_(h.mul_(1.0 / self.world))
        else:
            op = self._dist.ReduceOp.AVG if self._avg else self._dist.ReduceOp.SUM
            w = self._dist.all_reduce(buf, op=op, group=self.group, async_op=True)
            self._works.append((w, buf, dst))
        self.buckets_launched += 1
        self.bytes_reduced += buf.numel() * buf.element_size()

    def flush(self) -> None:
        """End of the encoder backward: launch whatever is still being merged."""
        self._launch_pending()

    def queue_finalize(self) -> None:
        """Called from inside a backward pass (the model's autograd node, the parameter hooks): finalize() runs when the
        autograd engine has finished this pass, before ``loss.backward()`` returns to the caller."""
        if self._callback_queued:
            return
        self._callback_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(self._engine_callback)

    def _engine_callback(self) -> None:
        self._callback_queued = False
        self.finalize()

    @staticmethod
    def _current_task() -> int:
        """Id of the autograd graph task this thread is executing (-1 outside a backward pass)."""
        f = getattr(torch._C, "_current_graph_task_id", None)
        return int(f()) if f is not None else -1

    def begin_forward(self) -> None:
        """Called by the model at the start of every training forward.  The autograd engine skips its end-of-backward callbacks when a
        backward pass raises (HIP error, out of memory, an exception in user code), which would leave `_callback_queued` set for good —
        no later backward would queue finalize() and torch optimisers would read gradients whose all-reduces are still in flight.

        A set flag is NOT proof of that by itself: a training forward that legitimately runs INSIDE a backward pass (activation
        checkpointing's recompute, a forward called from a hook after the callback was queued) sees it too.  Such a forward runs while an
        autograd graph task is executing on this thread — then nothing is touched.  Outside any backward pass a set flag does mean the pass
        that set it never finished:
          * world size 1: nothing is shared with anybody; drain what the failed pass left behind and start clean;
          * world size > 1: this rank cannot repair it alone — its peers have issued every bucket of that step, so this rank's next
            collectives would pair with the wrong buckets or sizes (an RCCL hang or silent corruption).  The reducer is marked poisoned
            and this and every later forward raise until the job is restarted (what torch DDP's reducer does after a failed backward)."""
        if self._poisoned is not None:
            raise RuntimeError(self._poisoned)
        if not self._callback_queued:
            return
        if self._current_task() != -1:
            return  # a forward inside a running backward pass: the callback of that pass is still going to run
        if self.world > 1:
            self._poisoned = ("diverse_channel_vit_amd.DataParallel: a backward pass did not finish on this rank (it raised before the autograd "
                              "engine ran the end-of-backward callback).  The ranks' all-reduce sequences no longer match; gradient exchange "
                              "cannot be resynchronised from one rank — restart the job.")
            raise RuntimeError(self._poisoned)
        self._callback_queued = False
        self._pending = None
        self._deferred = []
        for w, _, _ in self._works:
            try:
                w.wait()
            except Exception:  # the failed pass's collectives may have failed with it
                pass
        self._works.clear()

    # ---- small parameters outside the encoder node --------------------------------------------------
    def hook_misc_params(self) -> None:
        """All-reduce the gradient of every parameter that is not part of the encoder arena as soon as
        autograd has accumulated it.  Safe to call before the first forward (the arena need not exist yet)."""
        enc = {id(p) for p in self.model._enc_param_list()}
        for p in self.model.parameters():
            if id(p) in enc or id(p) in self._hooked or not p.requires_grad:
                continue
            self._hooked.add(id(p))
            self._hooks.append(p.register_post_accumulate_grad_hook(self._misc_hook))

    def _misc_hook(self, q) -> None:
        # .grad holds the sum over this step's backward passes.  Reducing the whole of it in every pass is right: what
        # earlier passes left there is already the rank average, which AVG (or SUM / world) maps onto itself.
        self._reduce(q.grad)
        self.queue_finalize()

    def pending_state(self) -> dict:
        """What the reducer is waiting for, for a diagnostic line (bench.py's watchdog): the outstanding collectives with their sizes and whether
        each has completed, the bucket being merged, the slices deferred to finalize().  Reads only; safe from another thread."""
        works = []
        for w, buf, _ in list(self._works):
            done = None
            try:
                done = bool(w.is_completed()) if w is not None else None
            except Exception:  # a backend without is_completed()
                pass
            works.append({"elements": int(buf.numel()), "dtype": str(buf.dtype).replace("torch.", ""), "completed": done})
        pend = self._pending
        return {"collectives_outstanding": works, "buckets_launched": self.buckets_launched,
                "bucket_being_merged": None if pend is None else {"lo": int(pend[1]), "hi": int(pend[2])},
                "slices_deferred": len(self._deferred), "callback_queued": bool(self._callback_queued), "poisoned": self._poisoned is not None}

    def _wait_all(self) -> None:
        for w, buf, dst in self._works:
            w.wait()
            if not self._avg:
                buf.mul_(1.0 / self.world)
            if dst is not None:
                dst.copy_(buf)
                if buf.is_cuda:  # buf may have been allocated while another stream was current (the model's weight-gradient stream)
                    buf.record_stream(torch.cuda.current_stream(buf.device))
        self._works.clear()

    def finalize(self) -> None:
        """Make the current stream wait for every outstanding all-reduce.  Runs by itself at the end of every backward
        pass (queue_finalize); HipAdamW.step and clip_grad_norm_ call it too (idempotent)."""
        self.flush()
        if self._deferred:
            ts, self._deferred = self._deferred, []
            # The encoder's slices are views of ONE arena and arrive last layer first: neighbours are merged into a single view and reduced
            # in place (no gather, no copy back: 2 x 86 MB of traffic per step for DiChaViT-S otherwise); what does not merge (the few small
            # parameters outside the arena) travels as one concatenated buffer.
            merged, loose = self._merge_views(ts)
            for m in merged:
                self._all_reduce(m)
            if loose:
                flat = torch.cat([t.reshape(-1) for t in loose]) if len(loose) > 1 else loose[0].reshape(-1)
                self._all_reduce(flat)
                self._wait_all()
                if len(loose) > 1:
                    o = 0
                    for t in loose:
                        t.copy_(flat[o:o + t.numel()].view_as(t))
                        o += t.numel()
        self._wait_all()

    @staticmethod
    def _merge_views(ts):
        """Splits `ts` into (merged, loose): 1-D contiguous views of a common base whose element ranges touch are replaced by one view over
        their union (in address order); everything else is returned unchanged in `loose`."""
        groups, loose = {}, []
        for t in ts:
            base = t._base
            if base is None or t.dim() != 1 or not t.is_contiguous() or not base.is_contiguous() or base.dtype != t.dtype:
                loose.append(t)
            else:
                groups.setdefault(id(base), (base, []))[1].append((t.storage_offset() - base.storage_offset(), t.numel()))
        merged = []
        for base, spans in groups.values():
            spans.sort()
            flat = base.view(-1)
            lo, hi = spans[0][0], spans[0][0] + spans[0][1]
            for o, n in spans[1:]:
                if o <= hi:
                    hi = max(hi, o + n)
                else:
                    merged.append(flat[lo:hi])
                    lo, hi = o, o + n
            merged.append(flat[lo:hi])
        return merged, loose

    def broadcast_parameters(self, src: int = 0) -> None:
        """Same initial weights on every rank (DDP does this in its constructor)."""
        arena = getattr(self.model, "_arena", None)
        if arena is not None:
            if self.backend == "gloo" and arena.is_cuda:
                h = arena.cpu()
                self._dist.broadcast(h, src=src, group=self.group)
                arena.copy_(h)
            else:
                self._dist.broadcast(arena, src=src, group=self.group)
        else:
            for p in self.model.parameters():
                if self.backend == "gloo" and p.is_cuda:
                    h = p.data.cpu()
                    self._dist.broadcast(h, src=src, group=self.group)
                    p.data.copy_(h)
                else:
                    self._dist.broadcast(p.data, src=src, group=self.group)
