"""Do the matrix pipe and the vector ALU of a SIMD overlap across waves?  Builds nothing: expects tools/probes/liboverlap_probe.so
(hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/overlap_probe.hip).  Prints cycles per iteration and wave for the MFMA phase
alone, the vector phase alone and both in sequence, at 1, 2 and 3 waves per SIMD."""
import ctypes as C, os, numpy as np, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboverlap_probe.so"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 2000
for nv in (128, 192):
    print(f"vector phase: {nv} fma + {nv // 4} exp per iteration; MFMA phase: 16 x 32x32x16 (512 matrix-pipe cycles)")
    for wps in (1, 2):
        blocks = 256 * wps  # one 256-thread block = one wave per SIMD of a CU
        out = torch.empty(blocks * 256, device="cuda"); cyc = torch.zeros(blocks * 4, dtype=torch.int64, device="cuda")
        res = []
        for mode in (0, 1, 2, 3):
            for _ in range(2):
                rc = lib.run_probe(mode, nv, C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), iters, blocks, st)
            torch.cuda.synchronize(); assert rc == 0
            res.append(float(np.median(cyc.cpu().numpy())) / iters)
        print(f"  {wps} wave(s)/SIMD: MFMA only {res[0]:7.1f}  vector only {res[1]:7.1f}  both in sequence {res[2]:7.1f} cycles per iteration and wave"
              f"  hand-interleaved in one wave {res[3]:7.1f}   -> per SIMD and wave-iteration: in sequence {res[2] / wps:6.1f}, interleaved {res[3] / wps:6.1f}"
              f" (sum of the phases {(res[0] + res[1]) / wps:6.1f}, larger phase {max(res[0], res[1]) / wps:6.1f})")
