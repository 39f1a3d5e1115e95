"""CPU experiment (oracle only, never product): where does the 100-step loss-curve gap of the HIP path come from, and what would
a hi+lo (two-term bf16) split of selected GEMM operands buy?  Emulates the MI355X path on the So2Sat-S curve with the oracle:
weights stochastically rounded to bf16 each step (as dcv_cast_bf16_sr), GEMM activation operands rounded to bf16, fp32 accumulate.
    mode A: every activation operand bf16 (the shipped path)
    mode B: LayerNorm outputs u1 / u2 (A operands of attn.qkv and mlp.fc1) as hi+lo, the rest bf16      
    mode C: every FORWARD activation operand hi+lo (u1, u2, attention output, GELU output), gradients operands bf16
    mode D: every activation and gradient operand hi+lo (weights still stochastically rounded bf16)
usage: python tools/curve_emul2.py A B C D"""
import sys, numpy as np, torch
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


_g = torch.Generator().manual_seed(5)
def sr(x):
    i = x.contiguous().view(torch.int32)
    r = torch.randint(0, 1 << 16, i.shape, generator=_g, dtype=torch.int32)
    return ((i + r) & ~0xFFFF).view(torch.float32)


MODE = "A"
class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, w, split_u):
        ctx.save_for_backward(u, w)
        ctx.split_u = split_u
        qu = hilo(u) if (split_u or MODE in "CD") else bf(u)
        return qu @ w.t()  # w arrives already stochastically rounded to bf16
    @staticmethod
    def backward(ctx, dy):
        u, w = ctx.saved_tensors
        qd = hilo(dy) if MODE == "D" else bf(dy)
        qu = hilo(u) if (ctx.split_u or MODE in "CD") else bf(u)
        du = qd @ w
        dw = qd.reshape(-1, dy.shape[-1]).t() @ qu.reshape(-1, u.shape[-1])
        return du, dw, None


class QW:
    def __init__(self, w, split): self.w, self.split = w, split
    def reshape(self, *s): return QW(self.w.reshape(*s), self.split)
    def t(self): return QWT(self.w, self.split)
    def dim(self): return self.w.dim()
    @property
    def shape(self): return self.w.shape
class QWT:
    def __init__(self, w, split): self.w, self.split = w, split
    def __rmatmul__(self, u): return _MM.apply(u, self.w, self.split)


def run():
    shapes = orc.state_shapes(cfg, meta["n_channels"], meta["img"], meta["num_classes"])
    sd = orc.make_state(shapes, meta["seed"])
    ms = {k: torch.zeros_like(v) for k, v in sd.items()}; vs = {k: torch.zeros_like(v) for k, v in sd.items()}
    batches = [orc.make_batch(meta["seed"] + 100 + i, meta["B"], meta["n_channels"], meta["img"], meta["num_classes"]) for i in range(meta["n_batches"])]
    ch = meta["mapper"]["train"]; idx = list(range(len(ch))); errs = []
    for s in range(meta["steps"]):
        x, y = batches[s % meta["n_batches"]]
        leaf, raw = {}, {}
        for k, v in sd.items():
            is_mat = v.dim() >= 2 and ((("attn" in k or "mlp" in k) and k.endswith("weight")) or "proj.weight" in k)
            if is_mat:
                w = sr(v.clone()).requires_grad_(True)
                leaf[k] = QW(w, MODE == "B" and ("attn.qkv" in k or "mlp.fc1" in k))
            else:
                w = v.clone().requires_grad_(True)
                leaf[k] = w
            raw[k] = w
        loss, *_ = orc.train_loss(leaf, x, y, cfg, ch, idx)
        loss.backward()
        errs.append(abs(loss.item() - ref[s]))
        for k in sd:
            if k == 'proxies': continue
            g = raw[k].grad
            if g is None: continue
            with torch.no_grad(): orc.adamw_step(sd[k], g, ms[k], vs[k], s + 1, meta["lr"], *meta["betas"], meta["eps"], meta["wd"])
    return np.array(errs)


for MODE in (sys.argv[1:] or ["A", "B"]):
    e = run()
    print(f"mode {MODE}: step0 {e[0]:.2e} max {e.max():.3e} mean {e.mean():.3e} tail20 {e[-20:].max():.3e} steps>1e-3: {(e > 1e-3).sum()}", flush=True)
