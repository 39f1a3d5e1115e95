"""Runs only the GEMM entries at the headline step's shapes, each case REPS times back to back in a fixed order (for rocprofv3 --pmc passes;
tools/pmc_gemm_read.py maps dispatches to cases by that order).  usage: python tools/gemm_only.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
hip.set_deterministic(False)  # the atomic forms: one dispatch per launch (the deterministic forms add a reduction launch per weight gradient)
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").to(bf); A3 = torch.randn(M, 3 * D, device="cuda").to(bf); A4 = torch.randn(M, 4 * D, device="cuda").to(bf)
CASES = [("nt qkv      N1152 K384  bias", "nt", A, 3 * D, hip.EPI_BIAS_BF16), ("nt fc1      N1536 K384  bias+gelu", "nt", A, 4 * D, hip.EPI_BIAS_GELU_BF16),
         ("nt gelu-bwd N1536 K384", "nt", A, 4 * D, hip.EPI_GELU_BWD_BF16), ("nt dgrad    N384  K1536 plain", "nt", A4, D, hip.EPI_PLAIN_BF16),
         ("nt dgrad    N384  K1152 plain", "nt", A3, D, hip.EPI_PLAIN_BF16), ("nt dgrad    N384  K384  plain", "nt", A, D, hip.EPI_PLAIN_BF16),
         ("nt fc2      N384  K1536 bias+resid", "nt", A4, D, hip.EPI_BIAS_RESID_F32), ("nt proj     N384  K384  bias+resid", "nt", A, D, hip.EPI_BIAS_RESID_F32),
         ("nt fc2      N384  K1536 bias+resid+LN", "ntln", A4, D, None), ("nt proj     N384  K384  bias+resid+LN", "ntln", A, D, None),
         ("tn wgrad    P1152 Q384", "tn", A3, A, None), ("tn wgrad    P1536 Q384", "tn", A4, A, None), ("tn wgrad    P384  Q1536", "tn", A, A4, None),
         ("tn wgrad    P384  Q384", "tn", A, A, None), ("tn group    fc2+fc1+proj+qkv", "tng", None, None, None)]
if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for name, kind, a, n_or_x, epi in CASES:
        if kind == "nt":
            N, K = n_or_x, a.shape[1]
            W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
            bias = torch.zeros(N, device="cuda")
            out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
            out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
            aux = torch.randn(M, N, device="cuda").to(bf) if epi == hip.EPI_GELU_BWD_BF16 else (torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None)
            torch.cuda.synchronize()
            for _ in range(reps):
                hip.gemm_nt(a, W, epi, out, bias=bias, out2=out2, aux=aux)
        elif kind == "ntln":  # the residual GEMMs with the following LayerNorm in the epilogue (dcv_gemm_nt_resid_ln, gemm_nt384_kernel<6>)
            N, K = n_or_x, a.shape[1]
            W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
            bias = torch.zeros(N, device="cuda"); resid = torch.randn(M, N, device="cuda"); x_out = torch.empty(M, N, device="cuda")
            gamma = torch.ones(N, device="cuda"); beta = torch.zeros(N, device="cuda"); u_out = torch.empty(M, N, dtype=bf, device="cuda")
            mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
            torch.cuda.synchronize()
            for _ in range(reps):
                hip.gemm_nt_resid_ln(a, W, bias, resid, x_out, gamma, beta, 1e-6, u_out, mean, rstd)
        elif kind == "tng":  # a block's four weight gradients in one launch (dcv_gemm_tn_group)
            prods = [(A, A4), (A4, A), (A, A), (A3, A)]
            outs = [(torch.zeros(Y.shape[1], X.shape[1], device="cuda"), torch.zeros(Y.shape[1], device="cuda")) for Y, X in prods]
            torch.cuda.synchronize()
            for _ in range(reps):
                hip.gemm_tn_acc_group([(Y, X, dW, db) for (Y, X), (dW, db) in zip(prods, outs)])
        else:
            Y, X = a, n_or_x
            dW = torch.zeros(Y.shape[1], X.shape[1], device="cuda"); db = torch.zeros(Y.shape[1], device="cuda")
            torch.cuda.synchronize()
            for _ in range(reps):
                hip.gemm_tn_acc(Y, X, dW, db)
        torch.cuda.synchronize()
