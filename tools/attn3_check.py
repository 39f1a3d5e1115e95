"""Round-5 harness for the third dK/dV kernel (csrc/attn_bwd3.hip): bitwise comparison with the second form (attn_bwd.hip, same summation order; a
-DDCV_DKDV_FORM=2 build, libdcv_hip_dkdv2.so) on several shapes, then interleaved timing at the headline shape.  python tools/attn3_check.py [lib.so ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip.so")]
hs = [C.CDLL(l) for l in libs]
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SC = 0.125 * 1.4426950408889634


def make(B, N, H, seed=0, zero=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    D = 64 * H
    qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g)
    qkv[:, :, :D] *= SC
    qkv = qkv.to(torch.bfloat16)
    dO = torch.randn(B, N, D, device="cuda", generator=g).to(torch.bfloat16)
    if zero:
        qkv.zero_(); dO.zero_()
    return dict(B=B, N=N, H=H, qkv=qkv, dO=dO, o=torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"), lse=torch.empty(B, H, N, device="cuda"),
                ws=torch.empty(2, B, H, N, device="cuda"))


def prep(h, d, nq):
    rc = h.dcv_attn_fwd_rows_ps(p(d["qkv"]), p(d["o"]), p(d["lse"]), d["B"], d["N"], nq, d["H"], 64, st)
    assert rc == 0, rc
    dq = torch.zeros_like(d["qkv"])
    rc = h.dcv_attn_bwd_dq_rows_ps(p(d["qkv"]), p(d["o"]), p(d["dO"]), p(d["lse"]), p(d["ws"]), p(dq), d["B"], d["N"], nq, d["H"], 64, C.c_float(0.125), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return dq, d["ws"].clone()


def run_dq(h, d, nq):
    dq = torch.full_like(d["qkv"], float("nan"))
    ws = torch.full_like(d["ws"], float("nan"))
    rc = h.dcv_attn_bwd_dq_rows_ps(p(d["qkv"]), p(d["o"]), p(d["dO"]), p(d["lse"]), p(ws), p(dq), d["B"], d["N"], nq, d["H"], 64, C.c_float(0.125), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return dq, ws


def run(h, name, d, nq):
    out = torch.full_like(d["qkv"], float("nan"))
    rc = getattr(h, name)(p(d["qkv"]), p(d["dO"]), p(d["lse"]), p(d["ws"]), p(out), d["B"], d["N"], nq, d["H"], 64, C.c_float(0.125), st)
    assert rc == 0, (name, rc)
    torch.cuda.synchronize()
    return out


h0 = hs[0]
href = C.CDLL(os.path.join(ROOT, "diverse_channel_vit_amd", "libdcv_hip_dkdv2.so"))  # -DDCV_DKDV_FORM=2 build: the second form behind the same entry
ok = True
for (B, N, H, nq) in [(1, 64, 1, 64), (2, 256, 2, 256), (1, 77, 3, 77), (2, 320, 6, 320), (3, 1569, 6, 1569), (2, 1569, 6, 1), (2, 600, 6, 33), (1, 4100, 2, 4100)]:
    d = make(B, N, H, seed=N)
    dq_ref, ws_ref = prep(href, d, nq)
    for h, l in zip(hs, libs):  # dQ (query rows < nq; the rest zero) and the workspace rows, third form against second
        dq_new, ws_new = run_dq(h, d, nq)
        D_ = 64 * H
        a, b_ = dq_ref[:, :, :D_].float(), dq_new[:, :, :D_].float()
        bad = int((torch.isnan(b_) | (a != b_)).sum())
        wa, wb = ws_ref[:, :, :, :nq], ws_new[:, :, :, :nq]
        wbad = int((torch.isnan(wb) | (wa != wb)).sum())
        print(f"B{B} N{N} H{H} Nq{nq} {os.path.basename(l)}: dQ mismatching {bad} of {a.numel()}  max|diff| {float((a - b_).abs().nan_to_num(1e30).max()):.3g}   workspace mismatching {wbad} of {wa.numel()}", flush=True)
        ok &= bad == 0 and wbad == 0
    ref = run(href, "dcv_attn_bwd_dkdv_rows_ps", d, nq)
    for h, l in zip(hs, libs):
        new = run(h, "dcv_attn_bwd_dkdv_rows_ps", d, nq)
        D = 64 * H
        rem = N % 256
        tail = rem if (0 < rem <= 64 and N > 256) else 0  # these keys go through the second form's TAIL2 mode: two partial sums added, not the sequential order
        a, b_ = ref[:, :N - tail, D:].float(), new[:, :N - tail, D:].float()
        bad = int((torch.isnan(b_) | (a != b_)).sum())
        mx = float((a - b_).abs().nan_to_num(1e30).max())
        msg = f"B{B} N{N} H{H} Nq{nq} {os.path.basename(l)}: mismatching elements {bad} of {a.numel()}  max|diff| {mx:.3g}  max|ref| {float(a.abs().max()):.3g}"
        if tail:
            ta, tb = ref[:, N - tail:, D:].float(), new[:, N - tail:, D:].float()
            terr = float((ta - tb).abs().nan_to_num(1e30).max())
            msg += f"   tail keys ({tail}): max|diff| {terr:.3g} of max|ref| {float(ta.abs().max()):.3g}"
            ok &= terr <= 1.6e-2 * float(ta.abs().max()) + 1e-6  # two bf16 roundings of differently ordered fp32 sums
        print(msg, flush=True)
        ok &= bad == 0
print("BITWISE", "OK" if ok else "MISMATCH", flush=True)

if os.environ.get("A3_TIME", "1") != "0":
    for zero in (False, True):
        for N in (1536, 1569):
            d = make(64, N, 6, zero=zero)
            prep(href, d, N)
            out = torch.empty_like(d["qkv"])
            names = [(href, "dcv_attn_bwd_dkdv_rows_ps", "dkdv2")] + [(h, "dcv_attn_bwd_dkdv_rows_ps", "dkdv3:" + os.path.basename(l)) for h, l in zip(hs, libs)]
            names += [(href, "DQ", "dq2")] + [(h, "DQ", "dq3:" + os.path.basename(l)) for h, l in zip(hs, libs)]
            res = {n[2]: [] for n in names}
            for rnd in range(14):
                for h, fn, tag in names:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(3):
                        if fn == "DQ":
                            h.dcv_attn_bwd_dq_rows_ps(p(d["qkv"]), p(d["o"]), p(d["dO"]), p(d["lse"]), p(d["ws"]), p(out), 64, N, N, 6, 64, C.c_float(0.125), st)
                        else:
                            getattr(h, fn)(p(d["qkv"]), p(d["dO"]), p(d["lse"]), p(d["ws"]), p(out), 64, N, N, 6, 64, C.c_float(0.125), st)
                    e.record()
                    torch.cuda.synchronize()
                    if rnd >= 2:
                        res[tag].append(s.elapsed_time(e) / 3 * 1e3)
            for tag, v in res.items():
                v = sorted(v)
                print(f"{'zeros ' if zero else 'random'} N{N} {tag:40s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}", flush=True)
