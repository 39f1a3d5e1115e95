"""Kernel-time bound for fusing the LayerNorm backward into the input-gradient GEMM, from kernels that exist, in ONE process, interleaved:
  pair   = the two launches the step runs today: dcv_gemm_nt (plain bf16 output, N = 384) + dcv_ln_bwd (reads du, x, dx_in; writes dx fp32 + bf16; column sums);
  standin = dcv_gemm_nt_resid_ln at the SAME K: one launch with a full-row epilogue of the same structure as the fused backward would have (fp32 tile read, row statistics
            exchanged across 8 lanes and two waves, fp32 + bf16 tile written) but WITHOUT its second fp32 tile read (dx_in: 154 MB), its second pass over x-hat and its
            2 x 384 column sums per tile — i.e. a lower bound on the fused kernel's time.
python tools/ln_bwd_bound.py  ->  median us per shape; gain bound = pair - standin - (154 MB at the copy rate)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
x = torch.randn(M, D, device="cuda"); dx = torch.randn(M, D, device="cuda"); dxo = torch.empty(M, D, device="cuda"); dxb = torch.empty(M, D, dtype=bf, device="cuda")
mean = x.mean(1); rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda"); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
du = torch.empty(M, D, dtype=bf, device="cuda")
xo = torch.empty(M, D, device="cuda"); u = torch.empty(M, D, dtype=bf, device="cuda"); mo = torch.empty(M, device="cuda"); ro = torch.empty(M, device="cuda")
bias = torch.zeros(D, device="cuda")
def tm(fn, n=12):
    ts = []
    for i in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e) * 1e3)
    return float(np.median(ts))
for K, name in ((1536, "fc1 input gradient -> LN2 backward"), (1152, "qkv input gradient -> LN1 backward")):
    A = (torch.randn(M, K, device="cuda") * 0.1).to(bf); W = (torch.randn(D, K, device="cuda") * 0.05).to(bf)
    res = {"gemm": [], "ln_bwd": [], "standin": []}
    for rnd in range(3):
        res["gemm"].append(tm(lambda: hip.gemm_nt(A, W, hip.EPI_PLAIN_BF16, du)))
        res["ln_bwd"].append(tm(lambda: hip.ln_bwd(du, x, mean, rstd, g, dx, dxo, dxb, dg, db, M, D)))
        res["standin"].append(tm(lambda: hip.gemm_nt_resid_ln(A, W, bias, x, xo, g, b, 1e-6, u, mo, ro)))
    gm, lb, st = (float(np.median(res[k])) for k in ("gemm", "ln_bwd", "standin"))
    extra = 154.2e6 / 4.78e12 * 1e6  # the fused backward's second fp32 tile read at this box's device-copy rate (profiles/r02_x8)
    print(f"K {K:4d} ({name}): input-gradient GEMM {gm:6.1f} us + ln_bwd {lb:6.1f} us = {gm + lb:6.1f};  one launch with the forward's fused full-row epilogue {st:6.1f} us "
          f"(+ {extra:.0f} us for the second tile read) -> the fusion can save at most {gm + lb - st - extra:5.1f} us per LayerNorm", flush=True)
