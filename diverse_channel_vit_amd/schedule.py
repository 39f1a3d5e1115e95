"""Host-side restatement of the schedules the reference's trainer drives the optimiser with.
Pure arithmetic on ``optimizer.param_groups`` — works with HipAdamW (eager or graph-captured: the scalars are
re-uploaded by ``advance()`` every step) and with any torch optimizer.

* CosineLRSchedule — timm ``CosineLRScheduler`` as configured by configs/scheduler/cosine.yaml and called by the
  trainer: ``step(epoch)`` once per epoch (trainer.py:345-348); ``t_in_epochs=True`` so ``step_update`` is a no-op
  (trainer.py:1009-1010).  lr(e) = warmup_lr_init + e*(base-warmup_lr_init)/warmup_t for e < warmup_t, else
  lr_min + 0.5*(base-lr_min)*(1+cos(pi*(e-warmup_t)/(t_initial-warmup_t))) with ``warmup_prefix`` semantics.
  timm is not importable here and the reference holds no fixture for it: parity is unpinned at this boundary
  (documented algorithm restated; first-epoch value 4.9e-5 for the JUMP-CP script matches SURVEY §8c).
* cosine_wd_schedule — utils.cosine_scheduler (utils.py:563-574) used for weight decay when ``weight_decay_end`` is
  set (trainer.py:217-228, 1011-1019): one value per optimiser step."""
from __future__ import annotations

import math
from typing import List


class CosineLRSchedule:
    def __init__(self, optimizer, t_initial: int, lr_min: float = 1e-6, warmup_t: int = 0, warmup_lr_init: float = 1e-5,
                 warmup_prefix: bool = False, cycle_limit: int = 1):  # defaults of configs/scheduler/cosine.yaml
        self.opt = optimizer
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.t_initial, self.lr_min, self.warmup_t = t_initial, lr_min, warmup_t
        self.warmup_lr_init, self.warmup_prefix, self.cycle_limit = warmup_lr_init, warmup_prefix, cycle_limit
        if warmup_t:
            for g in optimizer.param_groups:
                g["lr"] = warmup_lr_init

    def lr_at(self, epoch: int, base: float) -> float:
        if epoch < self.warmup_t:
            return self.warmup_lr_init + epoch * (base - self.warmup_lr_init) / self.warmup_t
        t = epoch - self.warmup_t if self.warmup_prefix else epoch
        if t // self.t_initial >= self.cycle_limit:
            return self.lr_min
        t_cur = t % self.t_initial
        return self.lr_min + 0.5 * (base - self.lr_min) * (1 + math.cos(math.pi * t_cur / self.t_initial))

    def step(self, epoch: int) -> None:
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self.lr_at(epoch, b)

    def step_update(self, num_updates: int) -> None:  # t_in_epochs=True: per-batch updates do nothing
        return None


def cosine_wd_schedule(base_value: float, final_value: float, epochs: int, niter_per_ep: int, warmup_epochs: int = 0,
                       start_warmup_value: float = 0.0) -> List[float]:
    """utils.cosine_scheduler: linear warm-up then half-cosine from base_value to final_value, one entry per iteration."""
    warm = []
    wi = warmup_epochs * niter_per_ep
    if warmup_epochs > 0:
        warm = [start_warmup_value + (base_value - start_warmup_value) * i / max(wi - 1, 1) for i in range(wi)]
    n = epochs * niter_per_ep - wi
    rest = [final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n)) for i in range(n)]
    out = warm + rest
    assert len(out) == epochs * niter_per_ep
    return out
