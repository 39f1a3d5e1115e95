"""Where a persistent workgroup of the 256 x 384 GEMM spends its cycles: k-loop (operand DMA + MFMA) vs epilogue (slab transposition,
fused op, stores).  Needs the diagnostic build: python -c "from diverse_channel_vit_amd import _build; _build.build_variant('stamp', ['DCV_STAMP=1'], instrumented=('gemm.hip',))"
then  DCV_LIB=diverse_channel_vit_amd/libdcv_hip_stamp.so python tools/gemm_stamp.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").to(bf); A4 = torch.randn(M, 4 * D, device="cuda").to(bf)
ZERO = bool(os.environ.get("GB_ZERO"))
if ZERO:
    A.zero_(); A4.zero_()
cases = [("qkv  N1152 K384  bias->bf16", A, 3 * D, hip.EPI_BIAS_BF16), ("fc1  N1536 K384  bias+GELU (2 outputs)", A, 4 * D, hip.EPI_BIAS_GELU_BF16),
         ("fc2  N384  K1536 bias+resid f32", A4, D, hip.EPI_BIAS_RESID_F32), ("dgrad N384 K1536 plain bf16", A4, D, hip.EPI_PLAIN_BF16),
         ("gelu-bwd N1536 K384", A, 4 * D, hip.EPI_GELU_BWD_BF16), ("proj N384 K384 bias+resid", A, D, hip.EPI_BIAS_RESID_F32)]
tile = hip.TILE_NARROW if os.environ.get("STAMP_TILE") == "narrow" else hip.TILE_WIDE
print("tile:", "256x128" if tile == hip.TILE_NARROW else "256x384", "(cycles are shader cycles; the 'us' columns assume 1 cycle = 10 ns: read them as cycles / 100)")
for name, a, N, epi in cases:
    K = a.shape[1]
    W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    if ZERO:
        W.zero_()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
    out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
    aux = torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else (torch.randn(M, N, device="cuda").to(bf) if epi == hip.EPI_GELU_BWD_BF16 else None)
    stamp = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
    ts = []
    for it in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        sv = stamp.view(torch.float32)
        hip.gemm_nt(a, W, epi, out, bias=bias, out2=sv if epi == hip.EPI_BIAS_RESID_F32 else out2, aux=aux,
                    aux2=None if epi == hip.EPI_BIAS_RESID_F32 else sv, tile=tile)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    st = stamp.view(256, 4).cpu().numpy().astype(np.float64)
    loop, epi_c, n = st[:, 0], st[:, 1], st[:, 2]
    tot = loop + epi_c
    # s_memtime ticks at 100 MHz on gfx9 (constant), not the shader clock: report shares and microseconds
    print(f"{name:42s} launch {np.median(ts):7.1f} us | per WG: tiles {n.mean():.2f}  k-loop {loop.mean()/100:7.1f} us ({100*loop.sum()/tot.sum():4.1f} %)  "
          f"epilogue {epi_c.mean()/100:7.1f} us ({100*epi_c.sum()/tot.sum():4.1f} %) | per tile: loop {loop.sum()/n.sum()/100:6.2f} us  epilogue {epi_c.sum()/n.sum()/100:6.2f} us")
