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
qkv_ps = qkv.clone()
qkv_ps[:, :, :D] = (qkv[:, :, :D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
ws = torch.empty(2, B, H, N, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
handles = [C.CDLL(l) for l in libs]


wsf = {}
errw = {}


def fused(h):
    if not hasattr(h, "dcv_attn_bwd_fused"):
        return None
    h.dcv_attn_bwd_fused_ws_bytes.restype = C.c_size_t
    nb = h.dcv_attn_bwd_fused_ws_bytes(B, N, H)
    buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    off = (-buf.data_ptr()) % 256
    wsf[id(h)] = buf
    wp = C.c_void_p(buf.data_ptr() + off)
    h.dcv_attn_bwd_fused_err_ptr.restype = C.c_void_p
    eoff = h.dcv_attn_bwd_fused_err_ptr(wp, B, N, H) - buf.data_ptr()
    errw[id(h)] = lambda: buf[eoff:eoff + 128].view(torch.int32).tolist()
    return lambda: h.dcv_attn_bwd_fused(p(qkv), p(o), p(dO), p(lse), wp, p(dqkv), B, N, H, 64, C.c_float(0.125), st)


def calls(h):
    f = fused(h)
    d = _calls(h)
    if f is not None:
        d["fused"] = f
    return d


def _calls(h):
    d = {"fwd": lambda: h.dcv_attn_fwd_rows(p(qkv), p(o), p(lse), B, N, N, H, 64, C.c_float(0.125), st),
         "dq": lambda: h.dcv_attn_bwd_dq_rows(p(qkv), p(o), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st),
         "dkdv": lambda: h.dcv_attn_bwd_dkdv_rows(p(qkv), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st)}
    if hasattr(h, "dcv_attn_fwd_rows_ps"):  # the pre-scaled-q entries on the same operand with its q part scaled (same score distribution)
        d.update({"fwd_ps": lambda: h.dcv_attn_fwd_rows_ps(p(qkv_ps), p(o), p(lse), B, N, N, H, 64, st),
                  "dq_ps": lambda: h.dcv_attn_bwd_dq_rows_ps(p(qkv_ps), p(o), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st),
                  "dkdv_ps": lambda: h.dcv_attn_bwd_dkdv_rows_ps(p(qkv_ps), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st)})
    return d


fns = [calls(h) for h in handles]
KEYS = tuple(os.environ["AB_ONLY"].split(",")) if os.environ.get("AB_ONLY") else ("fwd", "fwd_ps", "dq", "dq_ps", "dkdv", "dkdv_ps", "fused")  # AB_ONLY=fwd: ablation builds
res = {(i, k): [] for i in range(len(libs)) for k in KEYS}
outs = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", 12))):
    for i, f in enumerate(fns):
        for k in KEYS:
            if k not in f:
                continue
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
    if id(handles[i]) in errw:
        print("  error word of the one-pass backward:", errw[id(handles[i])]()[0])
    t = {k: res[(i, k)] for k in KEYS if res[(i, k)]}
    same = "" if i == 0 else f"  max|dO-ref| {float((outs[i][0] - outs[0][0]).abs().max()):.3g} max|dqkv-ref| {float((outs[i][1] - outs[0][1]).abs().max()):.3g}"
    print(f"{os.path.basename(l):40s} " + "  ".join(f"{k} {np.median(v):7.1f} (min {min(v):7.1f})" for k, v in t.items()) +
          (f"  sum(fwd,dq,dkdv) {sum(np.median(t[k]) for k in ('fwd', 'dq', 'dkdv')):7.1f} us" if all(k in t for k in ('fwd', 'dq', 'dkdv')) else "") + same)
