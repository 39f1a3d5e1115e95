"""Can the weight-gradient GEMMs (load-bound) hide under the attention backward (VALU-bound) on a second stream?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diverse_channel_vit_amd import hip
hip.load()
B, N, H = 64, 1569, 6; D = 384; M = B * N
torch.manual_seed(0)
bf = torch.bfloat16
qkv = torch.randn(B, N, 3 * D, device="cuda").to(bf); o = torch.randn(B, N, D, device="cuda").to(bf); lse = torch.zeros(B, H, N, device="cuda")
dO = torch.randn(B, N, D, device="cuda").to(bf); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device="cuda")
hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125)
A = torch.randn(M, D, device="cuda").to(bf); Y3 = torch.randn(M, 3 * D, device="cuda").to(bf); Z = torch.randn(M, 4 * D, device="cuda").to(bf)
dWq = torch.zeros(3 * D, D, device="cuda"); dWp = torch.zeros(D, D, device="cuda"); dW1 = torch.zeros(4 * D, D, device="cuda"); dW2 = torch.zeros(D, 4 * D, device="cuda")
b3 = torch.zeros(3 * D, device="cuda"); b1 = torch.zeros(D, device="cuda"); b4 = torch.zeros(4 * D, device="cuda")
side = torch.cuda.Stream()
def attn(): hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, 0.125)
def tns():
    hip.gemm_tn_acc(Y3, A, dWq, b3); hip.gemm_tn_acc(A, A, dWp, b1); hip.gemm_tn_acc(Z, A, dW1, b4); hip.gemm_tn_acc(A, Z, dW2, b1)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def seq(): attn(); tns()
def par():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): tns()
    attn()
    torch.cuda.current_stream().wait_stream(side)
print("attn_bwd alone us", round(timeit(attn)), " 4xTN alone us", round(timeit(tns)), " sequential", round(timeit(seq)), " two streams", round(timeit(par)), "env", os.environ.get("DCV_TN_SMALL_TILE"))
