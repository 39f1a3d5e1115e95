"""Is the NT GEMM's epilogue bound per CU or by what all CUs share?  dcv_gemm_nt_ex at the headline shapes with the persistent grid capped
to 256 / 192 / 128 / 64 workgroups (one per CU): if the time per tile and CU does not move with the number of CUs at work, the phases are bound inside the
CU; if it falls, the CUs — all in the same phase at the same time — share a bound (HBM) that a phase shift ACROSS CUs would relieve.
python tools/gemm_caps.py  ->  us per launch and CU-us per 256-row tile for every cap."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
lib = hip.load()
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").to(bf); A4 = torch.randn(M, 4 * D, device="cuda").to(bf)
cases = [("qkv        N1152 K384  bias", A, 3 * D, hip.EPI_BIAS_BF16), ("fc1        N1536 K384  bias+gelu", A, 4 * D, hip.EPI_BIAS_GELU_BF16),
         ("gelu-bwd   N1536 K384", A, 4 * D, hip.EPI_GELU_BWD_BF16), ("dgrad fc1T N384  K1536 plain", A4, D, hip.EPI_PLAIN_BF16),
         ("fc2        N384  K1536 bias+resid", A4, D, hip.EPI_BIAS_RESID_F32), ("proj       N384  K384  bias+resid", A, D, hip.EPI_BIAS_RESID_F32)]
caps = [int(c) for c in os.environ.get("GC_CAPS", "256,192,128,64").split(",")]
rounds = int(os.environ.get("GB_ROUNDS", 10))
for name, a, N, epi in cases:
    K = a.shape[1]
    W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
    out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
    aux = torch.randn(M, N, device="cuda").to(bf) if epi == hip.EPI_GELU_BWD_BF16 else (torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None)
    for tile in (hip.TILE_WIDE, hip.TILE_NARROW):
        if lib.dcv_gemm_nt_pick(M, N, K, epi, tile) != tile:
            continue
        tiles = ((M + 255) // 256) * (N // (384 if tile == hip.TILE_WIDE else 128))
        res = {c: [] for c in caps}
        for rnd in range(rounds):
            for c in caps:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(3):
                    hip.gemm_nt(a, W, epi, out, bias=bias, out2=out2, aux=aux, tile=tile, grid_cap=c)
                e.record(); torch.cuda.synchronize()
                if rnd >= 2:
                    res[c].append(s.elapsed_time(e) * 1e3 / 3)
        tn = "wide  " if tile == hip.TILE_WIDE else "narrow"
        # rounds of tiles a CU walks = ceil(tiles / cap); CU-us per tile = launch time / that
        print(f"{name:34s} {tn} " + "  ".join(f"cap {c:3d}: {np.median(v):7.1f} us = {np.median(v) / -(-tiles // c):6.2f} per tile" for c, v in res.items()), flush=True)
