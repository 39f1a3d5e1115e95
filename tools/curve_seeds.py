"""Spread of the distinct-batch loss-curve errors over weight-rounding seeds, with and without the pre-scaled-q attention path: is a change in the
curve statistics between two builds a change of the arithmetic or one more draw of the stochastic rounding?  python tools/curve_seeds.py [seeds...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
import diverse_channel_vit_amd as dcv

dev = torch.device("cuda", 0)
seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]


def run(name, ps, seed, stochastic=True):
    meta, a = T.load_golden(name)
    model, _ = T.build(meta, dev)
    model.stochastic_weight_rounding = stochastic
    model.attn_prescaled = ps
    model._sr_seed = torch.full((1,), seed, dtype=torch.int32, device=dev)
    opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=meta["lr"], weight_decay=meta["wd"], betas=tuple(meta["betas"]),
                       eps=meta["eps"], model=model)
    batches = [T.orc.make_batch(meta["seed"] + 100 + i, meta["B"], meta["n_channels"], meta["img"], meta["num_classes"]) for i in range(meta["n_batches"])]
    batches = [(x.to(dev), y.to(dev)) for x, y in batches]
    ref = a["losses"][:, 0]
    errs = []
    for s in range(meta["steps"]):
        x, y = batches[s % meta["n_batches"]]
        opt.zero_grad()
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        opt.step()
        errs.append(abs(loss.item() - ref[s]))
    return np.array(errs)


for name in ("curve100_jumpcp_s_b8", "curve100_so2sat_s_distinct"):
    for ps in (False, True):
        for seed in seeds:
            e = run(name, ps, seed)
            print(f"{name:28s} ps={int(ps)} seed={seed}: step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 {e[-20:].max():.3e} above1e-3 {int((e > 1e-3).sum())}", flush=True)
