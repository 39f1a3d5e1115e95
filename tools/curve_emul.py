"""CPU experiment (oracle only, not product): how much of the 100-step loss-curve gap comes from bf16 WEIGHT operands?
Runs the So2Sat-S curve with (a) fp32, (b) matrices rounded to bf16 for forward/backward, (c) hi+lo two-term bf16."""
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
from oracle import dichavit_oracle as orc
from tests.conftest import load_golden

meta, a = load_golden("curve100_so2sat_s")
ref = a["losses"][:, 0]
cfg = meta["cfg"]
torch.set_num_threads(8)

def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def hilo(x):
    h = bf(x); return h + bf(x - h)

def run(q):
    shapes = orc.state_shapes(cfg, meta["n_channels"], meta["img"], meta["num_classes"])
    sd = orc.make_state(shapes, meta["seed"])
    ms = {k: torch.zeros_like(v) for k, v in sd.items()}; vs = {k: torch.zeros_like(v) for k, v in sd.items()}
    batches = [orc.make_batch(meta["seed"] + 100 + i, meta["B"], meta["n_channels"], meta["img"], meta["num_classes"]) for i in range(meta["n_batches"])]
    ch = meta["mapper"]["train"]; idx = list(range(len(ch))); errs = []
    for s in range(meta["steps"]):
        x, y = batches[s % meta["n_batches"]]
        leaf = {}
        for k, v in sd.items():
            w = v.clone()
            if q is not None and v.dim() >= 2 and ("attn" in k or "mlp" in k or "proj.weight" in k):
                w = q(w)
            leaf[k] = w.requires_grad_(True)
        loss, *_ = orc.train_loss(leaf, x, y, cfg, ch, idx)
        loss.backward()
        errs.append(abs(loss.item() - ref[s]))
        for k in sd:
            if k == 'proxies': continue
            g = leaf[k].grad
            if g is None: continue
            with torch.no_grad(): orc.adamw_step(sd[k], g, ms[k], vs[k], s + 1, meta["lr"], *meta["betas"], meta["eps"], meta["wd"])
    e = np.array(errs)
    return e
_g = torch.Generator().manual_seed(5)
def sr(x):
    i = x.contiguous().view(torch.int32)
    r = torch.randint(0, 1 << 16, i.shape, generator=_g, dtype=torch.int32)
    return ((i + r) & ~0xFFFF).view(torch.float32)
for name, q in (("stochastic-rounded bf16 weights", sr),):
    e = run(q)
    print(f"{name:14s} step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 {e[-20:].max():.3e}", flush=True)


# (d) forward with hi+lo weights, input-gradient (dgrad) with single-term bf16 weights, exact weight gradient
ACT = True
class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, w):
        ctx.save_for_backward(u, w)
        return (bf(u) if ACT else u) @ hilo(w).t()
    @staticmethod
    def backward(ctx, dy):
        u, w = ctx.saved_tensors
        if ACT: dy = bf(dy); u = bf(u)
        du = dy @ (hilo(w) if ACT else bf(w))
        dw = dy.reshape(-1, dy.shape[-1]).t() @ u.reshape(-1, u.shape[-1])
        return du, dw

class QW:
    def __init__(self, w): self.w = w
    def reshape(self, *s): return QW(self.w.reshape(*s))
    def t(self): return QWT(self.w)
    @property
    def grad(self): return self.w.grad
    def dim(self): return self.w.dim()
    @property
    def shape(self): return self.w.shape
class QWT:
    def __init__(self, w): self.w = w
    def __rmatmul__(self, u): return _MM.apply(u, self.w)

def run_d():
    shapes = orc.state_shapes(cfg, meta["n_channels"], meta["img"], meta["num_classes"])
    sd = orc.make_state(shapes, meta["seed"])
    ms = {k: torch.zeros_like(v) for k, v in sd.items()}; vs = {k: torch.zeros_like(v) for k, v in sd.items()}
    batches = [orc.make_batch(meta["seed"] + 100 + i, meta["B"], meta["n_channels"], meta["img"], meta["num_classes"]) for i in range(meta["n_batches"])]
    ch = meta["mapper"]["train"]; idx = list(range(len(ch))); errs = []
    for s in range(meta["steps"]):
        x, y = batches[s % meta["n_batches"]]
        leaf = {}; raw = {}
        for k, v in sd.items():
            w = v.clone().requires_grad_(True); raw[k] = w
            leaf[k] = QW(w) if (v.dim() >= 2 and (("attn" in k or "mlp" in k) and k.endswith("weight") or "proj.weight" in k)) else w
        loss, *_ = orc.train_loss(leaf, x, y, cfg, ch, idx)
        loss.backward()
        errs.append(abs(loss.item() - ref[s]))
        for k in sd:
            if k == 'proxies': continue
            g = raw[k].grad
            if g is None: continue
            with torch.no_grad(): orc.adamw_step(sd[k], g, ms[k], vs[k], s + 1, meta["lr"], *meta["betas"], meta["eps"], meta["wd"])
    return np.array(errs)
e = run_d() if 0 else np.zeros(1)
print(f"{'hi+lo fwd+dgrad, bf16 activations/grad operands':14s} step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 {e[-20:].max():.3e}", flush=True)
