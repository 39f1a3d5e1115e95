"""Where a wave of the attention forward spends its cycles: python tools/attn_stamp.py libdcv_hip_astamp.so  (a -DDCV_ATTN_STAMP=1 build).
Per wave: cycles in the counted vmcnt wait, in the s_barrier, in the DMA issue, and in total; printed as shares (median over waves)."""
import ctypes as C, sys, numpy as np, torch
lib = C.CDLL(sys.argv[1])
B, N, H, D = 64, 1569, 6, 384
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
grid = B * H * ((N + 127) // 128)
lse = torch.zeros(B * H * N + grid * 4 * 4 * 2 + 16, device="cuda")  # LSE rows + 4 u64 per wave
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    rc = lib.dcv_attn_fwd_rows(p(qkv), p(o), p(lse), B, N, N, H, 64, C.c_float(0.125), st)
torch.cuda.synchronize(); assert rc == 0
raw = lse[B * H * N:B * H * N + grid * 32].cpu().numpy().view(np.uint64).reshape(grid, 4, 4).astype(np.float64)
tot = raw[..., 3]
for name, i in (("vmcnt wait", 0), ("barrier", 1), ("DMA issue", 2)):
    sh = raw[..., i] / tot
    print(f"{name:12s} median {100 * np.median(sh):5.1f} %   p10 {100 * np.percentile(sh, 10):5.1f} %   p90 {100 * np.percentile(sh, 90):5.1f} %")
print(f"total cycles per wave: median {np.median(tot):.0f}  ({np.median(tot) / ((N + 63) // 64):.0f} per key tile)")
