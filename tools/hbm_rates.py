"""HBM rates this box sustains for the access mixes of the step's memory-bound phases (torch kernels, 1 GiB buffers):
pure write (fill), pure read (sum), copy (read + write).  python tools/hbm_rates.py"""
import torch
n = 1 << 28  # 1 GiB of f32
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
def t(fn, bytes_, name, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{name:28s} {bytes_ / ms / 1e9:7.2f} TB/s  ({ms * 1e3:7.1f} us)")
t(lambda: a.fill_(1.0), 4 * n, "write (fill_)")
t(lambda: a.sum(), 4 * n, "read (sum)")
t(lambda: b.copy_(a), 8 * n, "copy (read + write)")
ah = a.view(torch.bfloat16 if False else torch.float32)
h = torch.empty(n, dtype=torch.bfloat16, device="cuda")
t(lambda: h.copy_(a), 6 * n, "f32 -> bf16 cast (4r + 2w)")
