"""Times dcv_gemm_nt_ex at the headline step's shapes for every legal tile variant, interleaved in one process.
python tools/gemm_bench.py   ->  median / min microseconds per (product, epilogue, variant)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").to(bf); A3 = torch.randn(M, 3 * D, device="cuda").to(bf); A4 = torch.randn(M, 4 * D, device="cuda").to(bf)
if os.environ.get("GB_ZERO"):  # DVFS probe: all-zero operands (cdna guide rule 25)
    A.zero_(); A3.zero_(); A4.zero_()
cases = [("qkv        N1152 K384  bias", A, 3 * D, hip.EPI_BIAS_BF16), ("fc1        N1536 K384  bias+gelu", A, 4 * D, hip.EPI_BIAS_GELU_BF16),
         ("gelu-bwd   N1536 K384", A, 4 * D, hip.EPI_GELU_BWD_BF16), ("dgrad fc1T N384  K1536 plain", A4, D, hip.EPI_PLAIN_BF16),
         ("dgrad qkvT N384  K1152 plain", A3, D, hip.EPI_PLAIN_BF16), ("dgrad proj N384  K384  plain", A, D, hip.EPI_PLAIN_BF16),
         ("fc2        N384  K1536 bias+resid", A4, D, hip.EPI_BIAS_RESID_F32), ("proj       N384  K384  bias+resid", A, D, hip.EPI_BIAS_RESID_F32)]
names = {hip.TILE_NARROW: "narrow", hip.TILE_WIDE: "wide", hip.TILE_PAIR: "pair", hip.TILE_ALT: "alt"}
rounds = int(os.environ.get("GB_ROUNDS", 12))
for name, a, N, epi in cases:
    K = a.shape[1]
    W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    if os.environ.get("GB_ZERO"):
        W.zero_()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
    out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
    aux = torch.randn(M, N, device="cuda").to(bf) if epi == hip.EPI_GELU_BWD_BF16 else (torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None)
    res = {t: [] for t in names}
    lib = hip.load()
    legal = [t for t in names if lib.dcv_gemm_nt_pick(M, N, K, epi, t) == t]
    res = {t: [] for t in legal}
    for rnd in range(rounds):
        for t in legal:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                hip.gemm_nt(a, W, epi, out, bias=bias, out2=out2, aux=aux, tile=t)
            e.record(); torch.cuda.synchronize()
            if rnd >= 2:
                res[t].append(s.elapsed_time(e) * 1e3 / 3)
    print(f"{name:34s} " + "  ".join(f"{names[t]} {np.median(v):7.1f} (min {min(v):7.1f})" for t, v in res.items()), flush=True)
