"""Times the three attention kernels at the headline shape for one or several builds of the library, interleaved in ONE process
(cdna guide rule 24): python tools/attn_bench.py [lib1.so lib2.so ...]   (default: the in-tree library).
Each build is loaded through ctypes directly (same C ABI); prints median / min microseconds per kernel and build."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = sys.argv[1:] or [os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so")]
B, N, H, D = int(os.environ.get("AB_B", 64)), int(os.environ.get("AB_N", 1569)), 6, 384
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, N, device="cuda")
dO = torch.randn(B, N, D, device="cuda").to(torch.bfloat16)
if os.environ.get("AB_ZERO"):  # DVFS probe: all-zero operands draw less power (cdna guide rule 25); compare against random data
    qkv.zero_(); dO.zero_()
dqkv = torch.empty_like(qkv)
ws = torch.empty(2, B, H, N, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
handles = [C.CDLL(l) for l in libs]


def calls(h):
    return {"fwd": lambda: h.dcv_attn_fwd_rows(p(qkv), p(o), p(lse), B, N, N, H, 64, C.c_float(0.125), st),
            "dq": lambda: h.dcv_attn_bwd_dq_rows(p(qkv), p(o), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st),
            "dkdv": lambda: h.dcv_attn_bwd_dkdv_rows(p(qkv), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st)}


fns = [calls(h) for h in handles]
res = {(i, k): [] for i in range(len(libs)) for k in ("fwd", "dq", "dkdv")}
outs = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", 12))):
    for i, f in enumerate(fns):
        for k in ("fwd", "dq", "dkdv"):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = f[k]()
            e.record()
            assert rc == 0, (libs[i], k, rc)
            torch.cuda.synchronize()
            if rnd >= 2:
                res[(i, k)].append(s.elapsed_time(e) * 1e3)
        if rnd == 0:
            outs[i] = (o.float().clone(), dqkv.float().clone())
for i, l in enumerate(libs):
    t = {k: res[(i, k)] for k in ("fwd", "dq", "dkdv")}
    same = "" if i == 0 else f"  max|dO-ref| {float((outs[i][0] - outs[0][0]).abs().max()):.3g} max|dqkv-ref| {float((outs[i][1] - outs[0][1]).abs().max()):.3g}"
    print(f"{os.path.basename(l):40s} " + "  ".join(f"{k} {np.median(v):7.1f} (min {min(v):7.1f})" for k, v in t.items()) +
          f"  sum {sum(np.median(v) for v in t.values()):7.1f} us" + same)
