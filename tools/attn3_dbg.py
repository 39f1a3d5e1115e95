import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = C.CDLL(os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so"))
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, N, H, nq = [int(x) for x in sys.argv[1:5]]
D = 64 * H
g = torch.Generator(device="cuda").manual_seed(N)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g); qkv[:, :, :D] *= 0.125 * 1.4426950408889634; qkv = qkv.to(torch.bfloat16)
dO = torch.randn(B, N, D, device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda"); ws = torch.empty(2, B, H, N, device="cuda")
h.dcv_attn_fwd_rows_ps(p(qkv), p(o), p(lse), B, N, nq, H, 64, st)
dq = torch.zeros_like(qkv)
h.dcv_attn_bwd_dq_rows_ps(p(qkv), p(o), p(dO), p(lse), p(ws), p(dq), B, N, nq, H, 64, C.c_float(0.125), st)
outs = []
for fn in ("dcv_attn_bwd_dkdv_rows_ps", "dcv_attn_bwd_dkdv_rows_ps", "dcv_attn_bwd_dkdv_rows_ps"):
    out = torch.full_like(qkv, float("nan"))
    getattr(h, fn)(p(qkv), p(dO), p(lse), p(ws), p(out), B, N, nq, H, 64, C.c_float(0.125), st)
    torch.cuda.synchronize()
    outs.append(out.float().view(B, N, 3, H, 64))
ref, new, new2 = outs
print("run-to-run identical:", bool(((new == new2) | (new.isnan() & new2.isnan())).all()))
for which, name in ((1, "dK"), (2, "dV")):
    bad = (ref[:, :, which] != new[:, :, which]) | new[:, :, which].isnan()   # [B, N, H, 64]
    print(name, "bad", int(bad.sum()), "nan", int(new[:, :, which].isnan().sum()))
    if bad.any():
        print("  by batch", bad.sum((1, 2, 3)).tolist())
        print("  by head", bad.sum((0, 1, 3)).tolist())
        kb = bad.sum((0, 2, 3))
        print("  keys with errors (first 40):", torch.nonzero(kb).flatten().tolist()[:40], " count", int((kb > 0).sum()))
        print("  by d:", bad.sum((0, 1, 2)).tolist())
