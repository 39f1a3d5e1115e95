"""Accuracy of the attention kernels of several library builds against the fp32 materialised softmax (torch), on the parity
test's inputs (incl. its spiked query/key pair): max abs error of O, LSE, dQ, dK, dV per build.  python tools/attn_accuracy.py lib1.so lib2.so"""
import ctypes as C, os, sys
import torch
libs = sys.argv[1:]
hs = [C.CDLL(l) for l in libs]
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ref(qkv, B, N, H, scale):
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * 64)
    return o, torch.logsumexp(s, -1)


for (B, N, H) in [(1, 128, 3), (2, 197, 3), (2, 289, 6), (1, 1569, 6), (3, 130, 2), (1, 192, 2), (2, 257, 1)]:
    D = H * 64
    g = torch.Generator(device="cpu").manual_seed(N)
    qkv = (torch.randn(B, N, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    qkv[0, N // 2, :64] *= 4
    qkv[0, N - 1, D:D + 64] = qkv[0, N // 2, :64]
    g2 = torch.Generator(device="cpu").manual_seed(7)
    dO = torch.randn(B, N, D, generator=g2).to(torch.bfloat16).cuda()
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = ref(qr, B, N, H, 0.125)
    o_ref.backward(dO.float())
    gr = qr.grad.reshape(B, N, 3, D)
    for l, h in zip(libs, hs):
        o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
        dqkv = torch.empty_like(qkv); ws = torch.empty(2, B, H, N, device="cuda")
        assert h.dcv_attn_fwd_rows(p(qkv), p(o), p(lse), B, N, N, H, 64, C.c_float(0.125), st) == 0
        assert h.dcv_attn_bwd_rows(p(qkv), p(o), p(dO), p(lse), p(ws), p(dqkv), B, N, N, H, 64, C.c_float(0.125), st) == 0
        d = dqkv.float().reshape(B, N, 3, D)
        errs = [float((o.float() - o_ref).abs().max()), float((lse - lse_ref).abs().max())] + [float((d[:, :, i] - gr[:, :, i]).abs().max()) for i in range(3)]
        tol = [3e-2 * float(gr[:, :, i].abs().max()) for i in range(3)]
        print(f"B{B} N{N} H{H} {os.path.basename(l):24s} O {errs[0]:.4f} LSE {errs[1]:.5f} dQ {errs[2]:.4f} (atol {tol[0]:.3f}) dK {errs[3]:.4f} ({tol[1]:.3f}) dV {errs[4]:.4f} ({tol[2]:.3f})")
