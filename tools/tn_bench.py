"""Times dcv_gemm_tn_acc_ex at the headline step's weight-gradient shapes for one or several library builds (interleaved, one process).
python tools/tn_bench.py [lib1.so lib2.so ...]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so")]
hs = [C.CDLL(l) for l in libs]
M, D = int(os.environ.get("TB_M", 64 * 1569)), 384  # TB_M: token rows (small M: the operands stay in the 256 MB memory-side cache between launches)
bf = torch.bfloat16
torch.manual_seed(0)
T = {d: torch.randn(M, d, device="cuda").to(bf) for d in (384, 1152, 1536)}
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (P, Q) in [(1536, 384), (384, 1536), (1152, 384), (384, 384)]:
    Y, X = T[P], (T[Q] if Q != P else torch.randn(M, Q, device="cuda").to(bf))
    dW = torch.zeros(P, Q, device="cuda"); db = torch.zeros(P, device="cuda")
    res = {i: [] for i in range(len(libs))}
    for rnd in range(int(os.environ.get("TB_ROUNDS", 12))):
        for i, h in enumerate(hs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                rc = h.dcv_gemm_tn_acc_ex(p(Y), P, p(X), Q, M, P, Q, p(dW), Q, p(db), int(os.environ.get("TB_TILE", 0)), st)
            e.record(); torch.cuda.synchronize()
            assert rc == 0
            if rnd >= 2:
                res[i].append(s.elapsed_time(e) * 1e3 / 3)
    print(f"M{M} P{P} Q{Q}: " + "  ".join(f"{os.path.basename(libs[i])} {np.median(v):7.1f} (min {min(v):7.1f})" for i, v in res.items()), flush=True)
