"""Host-side schedules.  The weight-decay schedule is pinned to the reference's
utils.cosine_scheduler through tests/golden/schedules.npz; the LR schedule restates timm's CosineLRScheduler,
which is neither importable here nor covered by any fixture of the reference: PARITY UNPINNED — only the
documented values are checked (first-epoch LR of the JUMP-CP script)."""
import math

import numpy as np
import torch

from conftest import load_golden
from diverse_channel_vit_amd.schedule import CosineLRSchedule, cosine_wd_schedule


def test_wd_schedule_matches_reference():
    meta, a = load_golden("schedules")
    for k, kw in enumerate(meta["cases"]):
        mine = np.asarray(cosine_wd_schedule(**kw))
        assert mine.shape == a[f"wd_{k}"].shape
        assert np.abs(mine - a[f"wd_{k}"]).max() < 1e-12


def test_cosine_lr_documented_values():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=4e-4)
    sch = CosineLRSchedule(opt, t_initial=100, lr_min=1e-6, warmup_t=10, warmup_lr_init=1e-5)
    assert opt.param_groups[0]["lr"] == 1e-5                       # initialize=True: starts at warmup_lr_init
    sch.step(1)
    assert abs(opt.param_groups[0]["lr"] - 4.9e-5) < 1e-12        # 1e-5 + (4e-4 - 1e-5)/10  
    sch.step(10)                                                    # warmup_prefix=False: cosine measured from epoch 0
    assert abs(opt.param_groups[0]["lr"] - (1e-6 + 0.5 * (4e-4 - 1e-6) * (1 + math.cos(math.pi * 10 / 100)))) < 1e-12
    sch.step(100)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12           # past the single cycle: lr_min
    lrs = [sch.lr_at(e, 4e-4) for e in range(10, 100)]
    assert all(x >= y for x, y in zip(lrs, lrs[1:]))                # monotone decay after warm-up
    sch.step_update(123)                                            # per-batch call is a no-op (t_in_epochs=True)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12
