"""Per-item and per-tile cost of the persistent dK/dV kernel from launches with Nq = 64 .. N query rows (no in-kernel stamps: they perturb the loop).
python tools/attn3_fit.py [lib.so] [N]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so"))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
B, H = 64, 6
D = 64 * H
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for zero in (True, False):
    qkv = torch.randn(B, N, 3 * D, device="cuda"); qkv[:, :, :D] *= 0.125 * 1.4426950408889634; qkv = qkv.to(torch.bfloat16)
    dO = torch.randn(B, N, D, device="cuda").to(torch.bfloat16)
    if zero:
        qkv.zero_(); dO.zero_()
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda"); ws = torch.empty(2, B, H, N, device="cuda")
    out = torch.empty_like(qkv)
    lib.dcv_attn_fwd_rows_ps(p(qkv), p(o), p(lse), B, N, N, H, 64, st)
    lib.dcv_attn_bwd_dq_rows_ps(p(qkv), p(o), p(dO), p(lse), p(ws), p(out), B, N, N, H, 64, C.c_float(0.125), st)
    xs, ys = [], []
    for fn in ("dcv_attn_bwd_dkdv_rows_ps", "dcv_attn_bwd_dkdv_rows_ps"):
        xs, ys = [], []
        for nq in (64, 128, 256, 512, 1024, N):
            ts = []
            for r in range(8):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    getattr(lib, fn)(p(qkv), p(dO), p(lse), p(ws), p(out), B, N, nq, H, 64, C.c_float(0.125), st)
                e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / 4 * 1e3)
            xs.append((nq + 63) // 64); ys.append(sorted(ts)[len(ts) // 2])
        A = np.vstack([np.ones(len(xs)), xs]).T
        c, *_ = np.linalg.lstsq(A, np.array(ys), rcond=None)
        print(("zeros " if zero else "random"), fn[-8:], "tiles:", xs, "us:", [round(y, 1) for y in ys], f" fit: {c[0]:.1f} us + {c[1]:.2f} us per tile", flush=True)
