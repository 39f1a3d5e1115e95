"""What the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the step's GEMM shapes: a ceiling reference for csrc/gemm.hip."""
import torch, json
M = 64 * 1569
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
res = {}
for name, (m, n, k, tn) in dict(qkv=(M, 1152, 384, 0), proj=(M, 384, 384, 0), fc1=(M, 1536, 384, 0), fc2=(M, 384, 1536, 0),
                                wgrad_qkv=(1152, 384, M, 1), wgrad_fc1=(1536, 384, M, 1), wgrad_fc2=(384, 1536, M, 1), wgrad_proj=(384, 384, M, 1)).items():
    if not tn:
        A = torch.randn(m, k, device="cuda").bfloat16(); W = torch.randn(n, k, device="cuda").bfloat16()
        us = t(lambda: torch.matmul(A, W.t()))
    else:
        Y = torch.randn(k, m, device="cuda").bfloat16(); X = torch.randn(k, n, device="cuda").bfloat16()
        us = t(lambda: torch.matmul(Y.t(), X))
    res[name] = dict(us=round(us, 1), tflops=round(2 * m * n * k / us / 1e6, 1))
print(json.dumps(res))
