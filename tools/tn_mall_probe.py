"""Does the weight-gradient launch run faster when its Y operand was written JUST before it (still in the 256 MB memory-side cache) than when the cache was flushed in between?
One product at a time (dcv_gemm_tn_acc, atomic mode), Y regenerated in place before every launch (write-only), optionally followed by a 2 GB fill that displaces it.
python tools/tn_mall_probe.py  ->  median us per product, hot and cold."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load(); hip.set_deterministic(False)
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
flush = torch.empty(1 << 30, dtype=torch.int16, device="cuda")  # 2 GB
for name, P, Q in (("qkv   P1152 Q384 ", 3 * D, D), ("fc1   P1536 Q384 ", 4 * D, D), ("fc2   P384  Q1536", D, 4 * D), ("proj  P384  Q384 ", D, D)):
    Y = torch.empty(M, P, dtype=bf, device="cuda"); X = torch.randn(M, Q, device="cuda").to(bf)
    dW = torch.zeros(P, Q, device="cuda"); db = torch.zeros(P, device="cuda")
    res = {}
    for mode in ("hot Y", "cold", "hot Y", "cold", "hot X+Y"):
        ts = []
        for it in range(12):
            if mode == "hot X+Y":
                X.normal_()
            Y.normal_(0, 0.1)
            if mode == "cold":
                flush.fill_(it)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); hip.gemm_tn_acc(Y, X, dW, db); e.record(); torch.cuda.synchronize()
            if it >= 2:
                ts.append(s.elapsed_time(e) * 1e3)
        res.setdefault(mode, []).append(np.median(ts))
    print(f"{name}  Y {M * P * 2 / 1e6:5.0f} MB  X {M * Q * 2 / 1e6:5.0f} MB   " + "   ".join(f"{k}: " + " / ".join(f"{v:6.1f}" for v in vs) for k, vs in res.items()), flush=True)
