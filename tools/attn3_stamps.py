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
        lib.dcv_attn_bwd_dkdv_rows_ps(p(qkv), p(dO), p(lse), p(ws), p(out), B, N, N, H, 64, C.c_float(0.125), st)
    torch.cuda.synchronize()
    nwg = min(256, B * H * ((N + 255) // 256))
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert lib.dcv_k3_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
    s = buf.reshape(-1, 8).astype(np.int64)
    tot, seam, items, wait, iss = s[:, 3] - s[:, 0], s[:, 1], s[:, 2], s[:, 4], s[:, 5]
    tiles = items * s[:, 7]
    print(("zeros " if zero else "random"), f"N{N}: workgroups {len(s)}  items/WG {items.min()}-{items.max()}  total {np.median(tot):.0f} cycles  per item {np.median(tot/items):.0f}"
          f"  seam+prologue per item {np.median(seam/items):.0f}  wait+barrier per tile {np.median(wait/tiles):.0f}  DMA issue per tile {np.median(iss/tiles):.0f}"
          f"  -> steps per tile {np.median((tot-seam-wait-iss)/tiles):.0f} ({np.median((tot-seam-wait-iss)/tiles)/64:.1f} per MFMA)")
