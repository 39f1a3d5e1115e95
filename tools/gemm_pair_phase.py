"""the two-workgroups-per-CU NT GEMM with a one-time start offset for the second workgroup of each CU.
Times fc1+GELU, GELU-bwd, fc2+resid, proj (+ qkv) on the pair tile for several -DDCV_PAIR_OFFSET builds against the product tiles, and reads the
-DDCV_PAIR_STAMP epilogue stamps: for every CU, the phase of workgroup B's epilogue starts inside workgroup A's tile period, tile by tile.
python tools/gemm_pair_phase.py  (builds libdcv_hip_pairNNNN.so must exist: tools/gemm_pair_phase.sh)"""
import ctypes as C, glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diverse_channel_vit_amd import hip
M, D = 64 * 1569, 384
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = {"product": os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so")}
for f in sorted(glob.glob(os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip_pair*.so"))):
    libs[os.path.basename(f)[11:-3]] = f
H = {k: C.CDLL(v) for k, v in libs.items()}
for zero in (False, True):
    torch.manual_seed(0)
    A = torch.randn(M, D, device="cuda").to(bf); A4 = torch.randn(M, 4 * D, device="cuda").to(bf)
    if zero:
        A.zero_(); A4.zero_()
    cases = [("fc1+GELU  N1536 K384", A, 4 * D, hip.EPI_BIAS_GELU_BF16), ("GELU-bwd  N1536 K384", A, 4 * D, hip.EPI_GELU_BWD_BF16),
             ("fc2+resid N384 K1536", A4, D, hip.EPI_BIAS_RESID_F32), ("proj+resid N384 K384", A, D, hip.EPI_BIAS_RESID_F32), ("qkv       N1152 K384", A, 3 * D, hip.EPI_BIAS_BF16)]
    print("== all-zero operands" if zero else "== random operands", flush=True)
    for name, a, N, epi in cases:
        K = a.shape[1]
        W = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        if zero:
            W.zero_()
        bias = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, dtype=torch.float32 if epi == hip.EPI_BIAS_RESID_F32 else bf, device="cuda")
        out2 = torch.empty(M, N, dtype=bf, device="cuda") if epi == hip.EPI_BIAS_GELU_BF16 else None
        aux = torch.randn(M, N, device="cuda").to(bf) if epi == hip.EPI_GELU_BWD_BF16 else (torch.randn(M, N, device="cuda") if epi == hip.EPI_BIAS_RESID_F32 else None)

        def call(h, tile):
            return h.dcv_gemm_nt_ex(p(a), K, p(W), K, M, N, K, epi, p(bias), p(out), N, p(out2), N if out2 is not None else 0, p(aux), N if aux is not None else 0,
                                    None, 0, 0, 0, tile, st)
        arms = [("product auto", H["product"], hip.TILE_AUTO)] + [(f"pair offset {k[4:] if k != 'product' else 0}", h, hip.TILE_PAIR) for k, h in H.items()]
        res = {n: [] for n, _, _ in arms}
        for rnd in range(10):
            for n, h, t in arms:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(3):
                    rc = call(h, t)
                e.record(); torch.cuda.synchronize()
                assert rc == 0, (n, rc)
                if rnd >= 2:
                    res[n].append(s.elapsed_time(e) * 1e3 / 3)
        print(f"{name:22s} " + "   ".join(f"{n} {np.median(v):6.1f}" for n, v in res.items()), flush=True)
        # phase table from the stamp builds (fc1 only)
        if name.startswith("fc1"):
            for k, h in H.items():
                if not hasattr(h, "dcv_pair_stamps"):
                    continue
                call(h, hip.TILE_PAIR); torch.cuda.synchronize()
                buf = np.zeros(1024 * 32, dtype=np.uint64)
                assert h.dcv_pair_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
                s_ = buf.reshape(1024, 32)[:512]
                ident = s_[:, 0]
                cu = ((ident >> 32) << 16) | ((ident & 0xFFFF) >> 8)  # (xcc, se/sh/cu)
                second = (ident & 15) != 0
                phases = []
                for c in np.unique(cu):
                    idx = np.nonzero(cu == c)[0]
                    if len(idx) != 2 or second[idx[0]] == second[idx[1]]:
                        continue
                    a_, b_ = (idx[0], idx[1]) if not second[idx[0]] else (idx[1], idx[0])
                    ta, tb = s_[a_, 1:17].astype(np.int64), s_[b_, 1:17].astype(np.int64)
                    per = np.diff(ta).mean()
                    phases.append(((tb - ta) % per) / per)
                ph = np.array(phases)
                print(f"   stamps {k}: CUs with one first-slot and one second-slot workgroup {len(ph)} of {len(np.unique(cu))};  tile period {per:.0f} cycles;"
                      f"  median phase of B's epilogue start in A's period, tiles 1..16: " + " ".join(f"{x:.2f}" for x in np.median(ph, axis=0)) if len(ph) else f"   stamps {k}: no pairs found", flush=True)
