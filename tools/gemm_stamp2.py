"""Fine cycle stamps of the 256 x 384 GEMM's k-loop, per wave: waiting for the stage's DMA (vmcnt), waiting at the barrier, first reads + DMA
issue, the 24 MFMA steps; plus the epilogue.  Needs the diagnostic build:
python -c "from diverse_channel_vit_amd import _build; _build.build_variant('stamp2', ['DCV_STAMP=2'])"
then  DCV_LIB=$PWD/diverse_channel_vit_amd/libdcv_hip_stamp2.so python tools/gemm_stamp2.py"""
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
cases = [("qkv  N1152 K384  bias->bf16", A, 3 * D, hip.EPI_BIAS_BF16), ("fc1  N1536 K384  bias+GELU", A, 4 * D, hip.EPI_BIAS_GELU_BF16),
         ("fc2  N384  K1536 bias+resid f32", A4, D, hip.EPI_BIAS_RESID_F32), ("dgrad N384 K1536 plain bf16", A4, D, hip.EPI_PLAIN_BF16)]
for name, a, N, epi in cases:
    K = a.shape[1]
    W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
    out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
    aux = torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None
    stamp = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    ts = []
    for it in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        sv = stamp.view(torch.float32)
        hip.gemm_nt(a, W, epi, out, bias=bias, out2=sv if epi == hip.EPI_BIAS_RESID_F32 else out2, aux=aux,
                    aux2=None if epi == hip.EPI_BIAS_RESID_F32 else sv, tile=hip.TILE_WIDE)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    st = stamp.view(256, 8, 8).cpu().numpy().astype(np.float64)
    nk = K // 64
    stages = st[:, :, 5] * nk  # stages per wave
    print(f"{name:34s} launch {np.median(ts):7.1f} us   tiles/WG {st[:, 0, 5].mean():.2f}   (cycles per STAGE, mean over workgroups)")
    for w in range(8):
        vm, bar, iss, mf = (st[:, w, i].sum() / stages[:, w].sum() for i in range(4))
        epi_c = st[:, w, 4].sum() / st[:, w, 5].sum()
        print(f"   wave {w}: vmcnt wait {vm:7.0f}  barrier {bar:7.0f}  reads+DMA issue {iss:7.0f}  MFMA steps {mf:7.0f}  | stage total {vm + bar + iss + mf:7.0f} | epilogue per tile {epi_c:8.0f}")
