"""Reads the cycle stamps of a -DDCV_K3_STAMP build of csrc/attn_bwd3.hip (diagnostic variant, never the product library):
python tools/attn3_stamps.py libdcv_hip_k3stamp.so [N]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
B, H = 64, 6
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
D = 64 * H
for zero in (True, False):
    qkv = torch.randn(B, N, 3 * D, device="cuda")
    qkv[:, :, :D] *= 0.125 * 1.4426950408889634
    qkv = qkv.to(torch.bfloat16)
    dO = torch.randn(B, N, D, device="cuda").to(torch.bfloat16)
    if zero:
        qkv.zero_(); dO.zero_()
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda"); ws = torch.empty(2, B, H, N, device="cuda")
    out = torch.empty_like(qkv)
    lib.dcv_attn_fwd_rows_ps(p(qkv), p(o), p(lse), B, N, N, H, 64, st)
    lib.dcv_attn_bwd_dq_rows_ps(p(qkv), p(o), p(dO), p(lse), p(ws), p(out), B, N, N, H, 64, C.c_float(0.125), st)
    for _ in range(20):
        lib.dcv_attn_bwd_dkdv_rows_ps3(p(qkv), p(dO), p(lse), p(ws), p(out), B, N, N, H, 64, C.c_float(0.125), st)
    torch.cuda.synchronize()
    nwg = B * H * ((N + 255) // 256)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert lib.dcv_k3_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
    s = buf.reshape(-1, 8).astype(np.int64)
    nt = s[:, 7]
    full = s[nt == nt.max()]
    pro, loop, epi = full[:, 1] - full[:, 0], full[:, 2] - full[:, 1], full[:, 3] - full[:, 2]
    print(("zeros " if zero else "random"), f"N{N}: workgroups {len(full)}  prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} = {np.median(loop)/nt.max():.0f} per tile ({np.median(loop)/nt.max()/64:.1f} per MFMA)"
          f"  wait+barrier {np.median(full[:,4])/nt.max():.0f} per tile  DMA issue {np.median(full[:,5])/nt.max():.0f} per tile  epilogue {np.median(epi):.0f}  total {np.median(full[:,3]-full[:,0]):.0f} cycles")
    t0 = s[:, 0].min()
    print("   kernel span (first entry -> last exit)", s[:, 3].max() - t0, "cycles;  starts by round:", np.percentile(s[:, 0] - t0, [0, 10, 25, 50, 75, 90, 100]).astype(int))
