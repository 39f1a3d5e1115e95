"""Is the chip power-limited on the step's kernels?  Samples board power and the shader clock (hwmon / rocm-smi sysfs nodes, read-only) every
~20 ms while one kernel is launched back to back for a few seconds: an NT GEMM with random and with all-zero operands, the attention forward,
LayerNorm (an HBM-bound kernel), and idle.  Prints mean / max power, the power cap, mean sclk and the kernel's time."""
import glob, os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()


def my_card_dirs():
    """The sysfs device directory of the GPU this process computes on (several cards are visible on a shared host): matched by PCI address."""
    bus = None
    try:
        import ctypes
        torch.cuda.init()
        hiprt = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hiprt.hipDeviceGetPCIBusId(buf, 64, torch.cuda.current_device()) == 0:
            bus = buf.value.decode()  # e.g. "0000:75:00.0"
    except Exception:
        bus = None
    dirs = []
    for d in glob.glob("/sys/class/drm/card*/device"):
        real = os.path.realpath(d)
        if bus and os.path.basename(real).lower() == str(bus).lower():
            dirs.append(d)
    return dirs, bus


def find_nodes():
    out = {}
    dirs, bus = my_card_dirs()
    pats = [d + "/hwmon/hwmon*" for d in dirs] or ["/sys/class/drm/card*/device/hwmon/hwmon*"]
    out["_card"] = [f"pci {bus}: {dirs}" if dirs else f"pci {bus}: NOT matched, reading the maximum over all visible cards"]
    for hw in (h for p in pats for h in glob.glob(p)):
        for key, name in (("power", "power1_average"), ("power_in", "power1_input"), ("cap", "power1_cap"), ("sclk", "freq1_input"), ("mclk", "freq2_input")):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(key, []).append(p)
    return out


NODES = find_nodes()


def read(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        pw = (NODES.get("power") or NODES.get("power_in") or [None])
        sc = NODES.get("sclk") or [None]
        while not self.stop:
            self.rows.append((max((read(p) for p in pw if p), default=float("nan")) / 1e6, max((read(p) for p in sc if p), default=float("nan")) / 1e6))
            time.sleep(0.02)


def run(name, fn, seconds=3.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    s.stop = True; s.join()
    r = np.array(s.rows[len(s.rows) // 3:])  # the last two thirds: past the ramp
    us = e0.elapsed_time(e1) * 1e3 / max(n, 1)
    print(f"{name:44s} {us:8.1f} us/launch | power mean {np.nanmean(r[:, 0]):7.1f} W  max {np.nanmax(r[:, 0]):7.1f} W | sclk mean {np.nanmean(r[:, 1]):7.1f} MHz  min {np.nanmin(r[:, 1]):7.1f}", flush=True)


print("card:", NODES.get("_card"))
print("nodes:", {k: v[:2] for k, v in NODES.items() if k != "_card"})
for p in NODES.get("cap", []):
    print("power cap:", read(p) / 1e6, "W", p)
M, D = 64 * 1569, 384
bf = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(M, D, device="cuda").to(bf); W = (torch.randn(3 * D, D, device="cuda") * 0.05).to(bf)
A0, W0 = torch.zeros_like(A), torch.zeros_like(W)
bias = torch.zeros(3 * D, device="cuda"); out = torch.empty(M, 3 * D, dtype=bf, device="cuda")
B, N, H = 64, 1569, 6
qkv = torch.randn(B, N, 3 * D, device="cuda").to(bf); o = torch.empty(B, N, D, dtype=bf, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda"); u = torch.empty(M, D, dtype=bf, device="cuda")
mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
time.sleep(1.0)
s = Sampler(); s.start(); time.sleep(1.5); s.stop = True; s.join()
r = np.array(s.rows)
print(f"{'idle':44s} {'':8s}           | power mean {np.nanmean(r[:, 0]):7.1f} W  max {np.nanmax(r[:, 0]):7.1f} W | sclk mean {np.nanmean(r[:, 1]):7.1f} MHz")
run("qkv GEMM wide, random operands", lambda: hip.gemm_nt(A, W, hip.EPI_BIAS_BF16, out, bias=bias, tile=hip.TILE_WIDE))
run("qkv GEMM wide, ALL-ZERO operands", lambda: hip.gemm_nt(A0, W0, hip.EPI_BIAS_BF16, out, bias=bias, tile=hip.TILE_WIDE))
run("qkv GEMM narrow, random operands", lambda: hip.gemm_nt(A, W, hip.EPI_BIAS_BF16, out, bias=bias, tile=hip.TILE_NARROW))
run("attention forward B64 N1569 H6", lambda: hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125))
dO = torch.randn(B, N, D, device="cuda").to(bf); dqkv = torch.empty_like(qkv); ws = torch.empty(2, B, H, N, device="cuda")
hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125)
lib = hip.load()
import ctypes as C
p_ = lambda t: C.c_void_p(t.data_ptr())
st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run("attention backward dQ", lambda: lib.dcv_attn_bwd_dq_rows(p_(qkv), p_(o), p_(dO), p_(lse), p_(ws), p_(dqkv), B, N, N, H, 64, C.c_float(0.125), st_))
run("attention backward dK/dV", lambda: lib.dcv_attn_bwd_dkdv_rows(p_(qkv), p_(dO), p_(lse), p_(ws), p_(dqkv), B, N, N, H, 64, C.c_float(0.125), st_))
qz = torch.zeros_like(qkv); dz = torch.zeros_like(dO)
run("attention backward dK/dV, ALL-ZERO operands", lambda: lib.dcv_attn_bwd_dkdv_rows(p_(qz), p_(dz), p_(lse), p_(ws), p_(dqkv), B, N, N, H, 64, C.c_float(0.125), st_))
run("attention forward, ALL-ZERO operands", lambda: hip.attn_fwd(qz, o, lse, B, N, H, 64, 0.125))
gq = lambda m, n, sc=1.0: (torch.randn(m, n, device="cuda") * sc).to(bf)
tn_ops = [(gq(M, D, 0.1), gq(M, 4 * D)), (gq(M, 4 * D, 0.1), gq(M, D)), (gq(M, D, 0.1), gq(M, D)), (gq(M, 3 * D, 0.1), gq(M, D))]
tn_out = [(torch.zeros(Y.shape[1], X.shape[1], device="cuda"), torch.zeros(Y.shape[1], device="cuda")) for Y, X in tn_ops]
run("weight gradients of a block, one grouped launch", lambda: hip.gemm_tn_acc_group([(Y, X, dW, db) for (Y, X), (dW, db) in zip(tn_ops, tn_out)]))
tn_zero = [(torch.zeros_like(Y), torch.zeros_like(X)) for Y, X in tn_ops]
run("the same on ALL-ZERO operands", lambda: hip.gemm_tn_acc_group([(Y, X, dW, db) for (Y, X), (dW, db) in zip(tn_zero, tn_out)]))
run("weight gradient fc1 alone (P1536 Q384)", lambda: hip.gemm_tn_acc(tn_ops[1][0], tn_ops[1][1], tn_out[1][0], tn_out[1][1]))
run("LayerNorm forward (HBM-bound)", lambda: hip.ln_fwd(x, g, b, u, mean, rstd, M, D, 1e-6))
