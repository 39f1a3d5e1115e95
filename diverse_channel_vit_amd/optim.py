"""Fused AdamW over the model's flat parameter/gradient arenas.

Restates what the reference builds with ``make_my_optimizer('adamw', ...)`` -> timm AdamW
(optimizers.py:20-21, trainer.py:1224-1230): ONE parameter group, decoupled weight decay on every
parameter that has a gradient (LayerNorm/bias/pos/cls included), bias-corrected Adam.  lr and
weight_decay are read from ``param_groups[0]`` every step, so the reference's per-epoch LR scheduler
and per-step weight-decay schedule (trainer.py:1009-1019) drive it unchanged.

When every encoder parameter's ``.grad`` aliases the model's gradient arena (the normal
zero_grad(set_to_none=True) -> backward flow) the whole encoder (≈99.7 % of the parameters) is
updated by ONE kernel launch over one contiguous range; anything else takes one launch per tensor.
Parameters whose grad is None are skipped, exactly like torch/timm AdamW."""
from __future__ import annotations

import torch

from . import hip


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("HipAdamW mirrors the reference's single parameter group")
        self.model = model
        self._step = 0
        self._m = self._v = None
        # capturable: the kernels read {lr, betas, eps, wd, bias corrections} from a device buffer that advance()
        # rewrites before every step, so step() can live inside a captured HIP graph (graph.GraphedTrainStep)
        self.capturable = capturable
        self._hyper_dev = None

    def advance(self):
        """capturable mode: bump the step count and hand this step's scalars to the device.  They travel BY VALUE as the
        arguments of a one-thread kernel on the current stream (dcv_adamw_set_hyper): stream-ordered against the replays, and
        there is no host staging buffer that a host running several steps ahead of the device could overwrite."""
        grp = self.param_groups[0]
        self._step += 1
        b1, b2 = grp["betas"]
        if self._hyper_dev is None:
            self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=grp["params"][0].device)
        hip.adamw_set_hyper(self._hyper_dev, float(grp["lr"]), b1, b2, float(grp["eps"]), float(grp["weight_decay"]), self._step, 1.0)

    def _ensure_state(self):
        model = self.model
        arena = model._arena
        if self._m is None or self._m.shape != arena.shape or self._m.device != arena.device:
            self._m = torch.zeros_like(arena)
            self._v = torch.zeros_like(arena)
            self._bound = set()
            self._off = {id(p): o for p, o in zip(model._all_params, model._all_off)}
            for p in model._all_params:
                if p in self.state and "exp_avg" in self.state[p]:  # moments that load_state_dict() put there (a resumed run)
                    self._bind(p, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"])

    def _bind(self, p, m_src=None, v_src=None):
        """state[p] = views into the flat moment arenas.  Done lazily, for parameters that actually get updated — like
        torch/timm AdamW, parameters that never receive a gradient (``proxies`` in CE mode) have no optimizer state, so
        state_dict() has the reference's layout (trainer.py:1296)."""
        o = self._off[id(p)]
        mv, vv = self._m[o:o + p.numel()].view(p.shape), self._v[o:o + p.numel()].view(p.shape)
        if m_src is not None:
            mv.copy_(m_src)
            vv.copy_(v_src)
        st = self.state[p]
        st["exp_avg"], st["exp_avg_sq"] = mv, vv
        self._bound.add(id(p))

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.AdamW / timm AdamW layouts ("optimizer_params" of the reference's checkpoints, trainer.py:1321):
        per-parameter exp_avg / exp_avg_sq / step, indexed in model.parameters() order.  The moments are copied into the
        flat arenas at the next step; the step count (bias corrections) continues from the loaded value."""
        super().load_state_dict(state_dict)
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        self._step = max(steps) if steps else 0
        self._m = self._v = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self.model
        if model is None or model._arena is None:
            raise RuntimeError("HipAdamW needs the arena-backed DiChaViT it optimises (pass model=...) after its first forward")
        if model._dp is not None:
            model._dp.finalize()
        self._ensure_state()
        grp = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(grp["lr"]), grp["betas"], float(grp["eps"]), float(grp["weight_decay"])
        if self.capturable:
            if self._hyper_dev is None:
                raise RuntimeError("capturable HipAdamW: call advance() before step()")
        else:
            self._step += 1
        step = self._step
        mine = {id(p) for p in grp["params"]}
        ga = model._grad_arena
        enc = model._enc_params
        fused = ga is not None and all(
            id(p) in mine and p.grad is not None and p.grad.data_ptr() == ga.data_ptr() + o * 4
            for p, o in zip(enc, model._enc_off))
        done = set()
        if fused:
            n = model._enc_size
            if self.capturable:
                hip.adamw_dyn(model._arena, ga, self._m, self._v, n, self._hyper_dev)
            else:
                hip.adamw(model._arena, ga, self._m, self._v, n, lr, b1, b2, eps, wd, step, 1.0)
            done = {id(p) for p in enc}
        off = {id(p): o for p, o in zip(model._all_params, model._all_off)}
        for p in grp["params"]:
            if id(p) in done or p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            o = off.get(id(p))
            if o is None:
                raise RuntimeError("parameter is not part of the model's arena")
            n = p.numel()
            if (o * 4) % 16 or g.data_ptr() % 16:
                raise RuntimeError("unaligned parameter slot")
            if self.capturable:
                hip.adamw_dyn(model._arena[o:o + n], g, self._m[o:o + n], self._v[o:o + n], n, self._hyper_dev)
            else:
                hip.adamw(model._arena[o:o + n], g, self._m[o:o + n], self._v[o:o + n], n, lr, b1, b2, eps, wd, step, 1.0)
        for p in grp["params"]:
            if p.grad is None:
                continue
            if id(p) not in self._bound:
                self._bind(p)
            self.state[p]["step"] = step
        return loss


@torch.no_grad()
def clip_grad_norm_(model, max_norm: float) -> torch.Tensor:
    """trainer.py:1003-1004 -> torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) (L2) for the arena-backed model:
    one sum-of-squares launch over the encoder's gradient arena plus one per small parameter outside it, then every
    gradient is multiplied by min(1, max_norm / (total_norm + 1e-6)).  The coefficient never leaves the device (no host
    sync; capturable).  Returns the total norm as a 0-d device tensor, like torch.  Under DataParallel the pending
    all-reduces are awaited first, so the norm is that of the averaged gradient on every rank."""
    if model._dp is not None:
        model._dp.finalize()
    ga = model._grad_arena
    enc = model._enc_params
    in_arena = ga is not None and all(p.grad is not None and p.grad.data_ptr() == ga.data_ptr() + o * 4 for p, o in zip(enc, model._enc_off))
    bufs = []
    seen = set()
    if in_arena:
        bufs.append((ga, model._enc_size))
        seen = {id(p) for p in enc}
    for p in model.parameters():
        if id(p) in seen or p.grad is None:
            continue
        seen.add(id(p))
        g = p.grad
        if not g.is_contiguous() or g.dtype != torch.float32 or g.data_ptr() % 16:
            p.grad = g = g.contiguous().float().clone()
        bufs.append((g, g.numel()))
    if not bufs:
        return torch.zeros((), device=model._arena.device if model._arena is not None else "cpu")
    acc = getattr(model, "_clip_acc", None)
    if acc is None or acc.device != bufs[0][0].device:
        acc = model._clip_acc = torch.zeros(1, dtype=torch.float32, device=bufs[0][0].device)
    acc.mul_(0.0)  # a kernel, not a memset node (memset nodes did not replay correctly inside captured graphs)
    for g, n in bufs:
        hip.sumsq_acc(g, n, acc)
    for g, n in bufs:
        hip.clip_scale(g, n, acc, max_norm)
    return acc.sqrt().reshape(())
