"""|loss - golden| of the six recorded HCS draws (tests/golden/hcs.npz) with the forward LayerNorm fused or not, stochastic or round-to-nearest
weight copies (round 4: the single-channel draw sits at 4.9e-3 / 5.1e-3 with stochastic copies, 1e-4 .. 2e-3 with round-to-nearest — the spread is
the weight-rounding noise of ONE forward, not the kernels).  python tools/hcs_loss_probe.py   (GPU)"""
import os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden
import test_model_gpu as T
from oracle import dichavit_oracle as orc
dev = torch.device("cuda")
meta, a = load_golden("hcs")
for fuse in (0, 1):
    for sr in (True, False):
        model, _ = T.build(meta, dev)
        model.fuse_ln_fwd = bool(fuse); model.fuse_ln_min_tiles = 0
        model.stochastic_weight_rounding = sr
        x, y = orc.make_batch(42, 3, 6, 32, 7)
        errs = []
        for k, d in enumerate(meta["draws"]):
            picked = a[f"d{k}_picked"].tolist()
            model.hcs_sampler = lambda m, chunk, cur, picked=picked: (picked, [cur.index(c) for c in picked])
            model.zero_grad(set_to_none=True)
            out, extra = model(x.to(dev), "train", None)
            loss = torch.nn.CrossEntropyLoss()(out, y.to(dev)) + extra
            errs.append(abs(loss.item() - float(a[f"d{k}_loss"])))
        print(f"fuse_ln={fuse} stochastic={sr} |dloss| per draw:", " ".join(f"{e:.2e}" for e in errs))
