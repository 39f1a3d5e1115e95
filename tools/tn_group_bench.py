"""One block's four weight-gradient products (fc2, fc1, proj, qkv at the headline shape) as four dcv_gemm_tn_acc launches against ONE
dcv_gemm_tn_group launch: microseconds per block (both in the current reduction mode), and the results against each other and fp32 torch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
M, D = int(os.environ.get("TB_M", 64 * 1569)), 384
bf = torch.bfloat16
torch.manual_seed(0)
g = lambda m, n, s=1.0: (torch.randn(m, n, device="cuda") * s).to(bf)
dxb, h, dz, u2, o, dqkv, u1 = g(M, D, 0.1), g(M, 4 * D), g(M, 4 * D, 0.1), g(M, D), g(M, D), g(M, 3 * D, 0.1), g(M, D)
prods = [(dxb, h), (dz, u2), (dxb, o), (dqkv, u1)]  # (Y, X): fc2, fc1, proj, qkv
mk = lambda: [(torch.zeros(Y.shape[1], X.shape[1], device="cuda"), torch.zeros(Y.shape[1], device="cuda")) for Y, X in prods]


def separate(outs):
    for (Y, X), (dW, db) in zip(prods, outs):
        hip.gemm_tn_acc(Y, X, dW, db)


def grouped(outs):
    hip.gemm_tn_acc_group([(Y, X, dW, db) for (Y, X), (dW, db) in zip(prods, outs)])


for det in (True, False):
    hip.set_deterministic(det)
    a, b = mk(), mk()
    separate(a); grouped(b)
    torch.cuda.synchronize()
    worst = 0.0
    for (Y, X), (dWa, dba), (dWb, dbb) in zip(prods, a, b):
        ref = Y.float().T @ X.float()
        rb = Y.float().sum(0)
        e = lambda t, r: ((t - r).norm() / r.norm()).item()
        worst = max(worst, e(dWb, ref), e(dbb, rb), e(dWa, ref), e(dba, rb))
        assert e(dWb, dWa) < 1e-5 and e(dbb, dba) < 1e-5, (e(dWb, dWa), e(dbb, dba))
    if det:
        c = mk(); grouped(c); torch.cuda.synchronize()
        assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(b, c)), "grouped deterministic form is not bit-reproducible"
    res = {"separate": [], "grouped": []}
    outs = mk()
    for rnd in range(12):
        for name, fn in (("separate", separate), ("grouped", grouped)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                fn(outs)
            e.record(); torch.cuda.synchronize()
            if rnd >= 2:
                res[name].append(s.elapsed_time(e) * 1e3 / 3)
    print(f"M{M} {'deterministic' if det else 'atomic':13s}: four launches {np.median(res['separate']):7.1f} us (min {min(res['separate']):7.1f})   one grouped launch "
          f"{np.median(res['grouped']):7.1f} us (min {min(res['grouped']):7.1f})   worst relative error vs fp32 {worst:.2e}", flush=True)
