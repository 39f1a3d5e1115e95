"""HIP-graph capture of one whole training step (zero_grad -> forward -> loss -> backward -> HipAdamW.step).

The eager step issues ~390 kernel launches from Python; on MI355X that leaves the GPU idle ~10 % of the time
(profiles/: 5 ms of 52 ms).  Capturing the step once and replaying it removes the launch gaps; data and
optimiser scalars stay live: inputs are copied into static buffers before each replay and HipAdamW(capturable=True)
reads lr / weight decay / bias corrections from a device buffer that ``advance()`` rewrites every step, so LR
schedulers and weight-decay schedules that edit ``param_groups[0]`` (trainer.py:1009-1019) keep working.

Constraints (checked or documented): fixed input shape and channel subset (enable_sample=False, or a pinned
``hcs_sampler``), DropPath masks from the device generator (torch.cuda.graph registers it: every replay draws new masks), single process (RCCL collectives are not captured: the multi-GPU path stays eager)."""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, chunk_name: str = "train", training_chunks: Optional[str] = None,
                 loss_fn: Optional[Callable] = None, extra_loss_lambda: float = 1.0, warmup: int = 2):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs HipAdamW(capturable=True)")
        if model.training and model.feature_extractor.patch_embed.enable_sample and model.hcs_sampler is None:
            raise ValueError("HCS sampling changes the sequence length every step: pin it (model.hcs_sampler) or run eager")
        if model.training and (model.cfg.get("dropout_tokens_hcs", "none") if hasattr(model.cfg, "get") else "none") not in (None, "none"):
            raise ValueError("dropout_tokens_hcs draws a new token subset every step from the python RNG: a captured step would "
                             "replay the subset drawn at capture time — run eager")
        if model.training and getattr(model, "drop_path_sampler", None) is not None and any(r > 0 for r in model.feature_extractor.drop_path_rates):
            raise ValueError("drop_path_sampler hands back host-made masks: a captured step would replay the masks of the capture — "
                             "unset it (the default draw uses the device generator, which torch.cuda.graph advances per replay) or run eager")
        self.model, self.opt = model, optimizer
        self.chunk_name, self.training_chunks = chunk_name, training_chunks
        self.loss_fn = loss_fn or torch.nn.CrossEntropyLoss()
        self.lam = extra_loss_lambda
        self.warmup = warmup
        self.graph = None
        self.capture_two_streams = False  # see _capture
        self.static_x = self.static_y = self.static_loss = self.static_out = self.static_extra = None

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        out, extra = self.model(self.static_x, self.chunk_name, self.training_chunks, init_first_layer=None,
                                new_channel_init=None, cur_epoch=0)
        loss = self.loss_fn(out, self.static_y) + extra * self.lam
        loss.backward()
        self.opt.step()
        self.static_out, self.static_extra = out.detach(), extra.detach()  # valid after every replay, like the loss
        return loss

    def _capture(self, x, y):
        self.static_x, self.static_y = x.clone(), y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):  # allocator warm-up + lazy state (arena, AdamW moments) outside the graph
                self.opt.advance()
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        # One stream inside the capture.  With the weight-gradient GEMMs on a second stream every hand-over between the streams becomes an
        # edge between two branches of the graph, which the replay realises as dependency packets in BOTH queues — the waits the eager
        # step keeps out of its compute queue (per-layer scratch).  Measured at the headline shape (round 4, one box, gpurun_out/r4f):
        # eager 35.38 (one stream) / 35.33 ms (two); captured 35.21 ms (one) / 35.94 ms (two).  The captured step therefore runs on one
        # stream (0.4 % faster than the best eager form instead of 1.7 % slower); `capture_two_streams = True` keeps the model's setting.
        two = self.model.wgrad_stream
        if not self.capture_two_streams:
            self.model.wgrad_stream = False
        try:
            with torch.cuda.graph(g):
                self.static_loss = self._eager()
        finally:
            self.model.wgrad_stream = two
        self.graph = g

    def __call__(self, x, y):
        """Runs one optimiser step on (x, y); returns the loss tensor (device, overwritten by the next call).
        The first call also performs `warmup` eager steps on the same batch before capturing."""
        if self.graph is None:
            self._capture(x, y)
        if x.shape != self.static_x.shape or y.shape != self.static_y.shape:
            raise ValueError("GraphedTrainStep was captured for a different batch shape")
        self.static_x.copy_(x, non_blocking=True)
        self.static_y.copy_(y, non_blocking=True)
        self.opt.advance()
        self.graph.replay()
        return self.static_loss
