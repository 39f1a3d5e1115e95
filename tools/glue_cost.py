"""What do the two tiny-tensor regularisers cost on the step's critical path?  Headline step with the proxy term / both terms switched off
(proxy_loss_lambda = 0: ~40 launches fewer; ortho_loss_v1_lambda = 0 as well: also the tokeniser's statistics kernels), alternating, ms per step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((64, 8, 224, 224)).astype(np.float32)).to(dev); y = torch.from_numpy(rs.randint(0, 161, 64)).to(dev)
ce = torch.nn.CrossEntropyLoss()


def make(**over):
    cfg = bench.model_cfg("small", 8, 224, 16, 161)
    cfg.update(over)
    torch.manual_seed(0)
    m = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
    o = dcv.HipAdamW([p for p in m.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=m)
    return m, o


def run(m, o, n=30):
    def step():
        o.zero_grad()
        out, extra = m(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        (ce(out, y) + extra).backward()
        o.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


variants = {"both regularisers (headline)": {}, "no proxy term": dict(proxy_loss_lambda=0.0), "neither": dict(proxy_loss_lambda=0.0, ortho_loss_v1_lambda=0.0)}
models = {k: make(**v) for k, v in variants.items()}
res = {k: [] for k in variants}
for rnd in range(3):
    for k, (m, o) in models.items():
        res[k].append(run(m, o))
for k, v in res.items():
    print(f"{k:32s} {np.median(v):7.3f} ms per step  ({' '.join(f'{t:.3f}' for t in v)})")
