import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo")
from diverse_channel_vit_amd import hip
hip.load()
bf = torch.bfloat16
D = 384
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for M in (21 * 589, 21 * 785, 22 * 981, 16 * 1569, 32 * 1569):
    for name, N, K, epi in (("qkv", 1152, 384, hip.EPI_BIAS_BF16), ("fc1", 1536, 384, hip.EPI_BIAS_GELU_BF16), ("fc2r", 384, 1536, hip.EPI_BIAS_RESID_F32), ("fc1T", 384, 1536, hip.EPI_PLAIN_BF16)):
        A = torch.randn(M, K, device="cuda").to(bf); W = (torch.randn(N, K, device="cuda") * 0.05).to(bf); bias = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
        out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
        aux = torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None
        r = {}
        for tl, nm in ((hip.TILE_NARROW, "narrow"), (hip.TILE_WIDE, "wide")):
            r[nm] = min(t(lambda: hip.gemm_nt(A, W, epi, out, bias=bias, out2=out2, aux=aux, tile=tl)) for _ in range(3))
        tw = ((M + 255) // 256) * (N // 384); tn = ((M + 255) // 256) * (N // 128)
        print(f"M {M:6d} {name:5s} narrow {r['narrow']:7.1f} us ({tn:4d} tiles)  wide {r['wide']:7.1f} us ({tw:4d} tiles)")
