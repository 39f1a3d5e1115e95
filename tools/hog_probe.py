"""How do the persistent one-round GEMM grids behave when another kernel holds some CUs (as RCCL's all-reduce does during the
data-parallel backward)?  A hog kernel (tools/probes/hog_probe.hip, built into tools/probes/libhog_probe.so) occupies `wgs` CU slots on a side stream for ~3 ms; the GEMMs are timed on
the main stream meanwhile."""
import ctypes, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
lib = hip.load()
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes', 'libhog_probe.so'))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B, N, H, D = 64, 1569, 6, 384; M = B * N
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").bfloat16(); W = (torch.randn(3 * D, D, device="cuda") * 0.05).bfloat16(); bias = torch.zeros(3 * D, device="cuda"); out = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda")
W1 = (torch.randn(4 * D, D, device="cuda") * 0.05).bfloat16(); z = torch.empty(M, 4 * D, dtype=torch.bfloat16, device="cuda"); hh = torch.empty_like(z)
Wp = (torch.randn(D, D, device="cuda") * 0.05).bfloat16(); bp = torch.zeros(D, device="cuda"); xr = torch.randn(M, D, device="cuda"); xo = torch.empty_like(xr)
dW = torch.zeros(3 * D, D, device="cuda"); db = torch.zeros(3 * D, device="cuda")
qkv = torch.randn(B, N, 3 * D, device="cuda").bfloat16(); o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
side = torch.cuda.Stream()
def timed(fn, wgs, lds):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    if wgs:
        with torch.cuda.stream(side):
            hog.hog_launch(wgs, lds, 8000, side.cuda_stream)  # ~4 ms
        import time; time.sleep(0.0005)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 * 1e3
cases = {"nt384 qkv": lambda: hip.gemm_nt(A, W, hip.EPI_BIAS_BF16, out, bias=bias),
         "nt384 fc1+gelu": lambda: hip.gemm_nt(A, W1, hip.EPI_BIAS_GELU_BF16, z, bias=bias.repeat(2)[:4 * D].contiguous() if False else torch.zeros(4 * D, device="cuda"), out2=hh),
         "nt256 proj+resid": lambda: hip.gemm_nt(A, Wp, hip.EPI_BIAS_RESID_F32, xo, bias=bp, aux=xr),
         "tn384 wgrad qkv": lambda: hip.gemm_tn_acc(out, A, dW, db),
         "attn fwd": lambda: hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125)}
for name, fn in cases.items():
    row = {}
    for wgs, lds in ((0, 0), (8, 1024), (8, 21184), (8, 65536), (32, 1024), (32, 21184), (32, 65536)):  # 21184 B: RCCL's device kernels
        row[f"{wgs}x{lds // 1024}K"] = round(timed(fn, wgs, lds))
    print(name, json.dumps(row), flush=True)
