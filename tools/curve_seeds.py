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


if os.environ.get("CS_BATCH_SCALING"):  # round 5: the headline distinct-batch curves at bs 8 / 16 / 32 (default build), spread over rounding seeds
    rows = []
    for name in ("curve100_jumpcp_s_b8", "curve100_jumpcp_s_b16", "curve100_jumpcp_s_b32"):
        if not os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".npz")):
            continue
        es = []
        for seed in seeds:
            e = run(name, True, seed)
            es.append(e)
            print(f"{name:24s} seed={seed}: step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 max {e[-20:].max():.3e} tail20 mean {e[-20:].mean():.3e} above1e-3 {int((e > 1e-3).sum())}", flush=True)
        es = np.array(es)
        rows.append((name, es))
        print(f"  -> over {len(seeds)} draws: mean {es.mean(1).min():.2e} .. {es.mean(1).max():.2e}; last-20 max {es[:, -20:].max(1).min():.2e} .. {es[:, -20:].max(1).max():.2e};"
              f" worst step {es.max(1).min():.2e} .. {es.max(1).max():.2e}; rms over steps 20..99 and draws {np.sqrt((es[:, 20:] ** 2).mean()):.3e}", flush=True)
    sys.exit(0)
for name in ("curve100_jumpcp_s_b8", "curve100_so2sat_s_distinct"):
    for ps in (False, True):
        for seed in seeds:
            e = run(name, ps, seed)
            print(f"{name:28s} ps={int(ps)} seed={seed}: step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 {e[-20:].max():.3e} above1e-3 {int((e > 1e-3).sum())}", flush=True)
