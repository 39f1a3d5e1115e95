"""Times dcv_ln_fwd / dcv_ln_bwd at the headline shape (M = 100 416, D = 384) for one or several library builds, interleaved in one
process, and checks the builds against each other: python tools/ln_bench.py [lib1.so lib2.so ...]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so")]
hs = [C.CDLL(l) for l in libs]
M, D = 64 * 1569, 384
torch.manual_seed(0)
xs = [torch.randn(M, D, device="cuda") for _ in range(3)]
dus = [torch.randn(M, D, device="cuda").to(torch.bfloat16) for _ in range(3)]
g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
u = [torch.empty(M, D, dtype=torch.bfloat16, device="cuda") for _ in hs]
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dxi = torch.randn(M, D, device="cuda")
dxo = [torch.empty(M, D, device="cuda") for _ in hs]
dxb = [torch.empty(M, D, dtype=torch.bfloat16, device="cuda") for _ in hs]
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L = C.c_long
it = [0]
def fwd(i):
    it[0] += 1
    return hs[i].dcv_ln_fwd(p(xs[it[0] % 3]), L(D), p(g), p(b), p(u[i]), 0, p(mean), p(rstd), M, D, C.c_float(1e-6), st)
def bwd(i):
    it[0] += 1
    k = it[0] % 3
    return hs[i].dcv_ln_bwd(p(dus[k]), 0, p(xs[k]), L(D), p(mean), p(rstd), p(g), p(dxi), p(dxo[i]), L(D), p(dxb[i]), p(dg), p(db), M, D, st)
res = {(i, k): [] for i in range(len(hs)) for k in ("fwd", "bwd")}
for rnd in range(int(os.environ.get("LB_ROUNDS", 12))):
    for i in range(len(hs)):
        for k, fn in (("fwd", fwd), ("bwd", bwd)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6):
                rc = fn(i)
            e.record(); torch.cuda.synchronize()
            assert rc == 0
            if rnd >= 2:
                res[(i, k)].append(s.elapsed_time(e) * 1e3 / 6)
for i, l in enumerate(libs):
    print(f"{os.path.basename(l):28s} " + "  ".join(f"{k} {np.median(res[(i, k)]):6.1f} (min {min(res[(i, k)]):6.1f})" for k in ("fwd", "bwd")), flush=True)
# same inputs through every build: outputs must agree bit for bit (the variants only change how bytes travel)
it[0] = 0
for i in range(len(hs)):
    it[0] = 2; fwd(i); it[0] = 2; bwd(i)
torch.cuda.synchronize()
for i in range(1, len(hs)):
    print(f"{os.path.basename(libs[i])} vs {os.path.basename(libs[0])}: u equal {torch.equal(u[i], u[0])}, dx equal {torch.equal(dxo[i], dxo[0])}, dx_bf16 equal {torch.equal(dxb[i], dxb[0])}")
