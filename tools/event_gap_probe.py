"""What does an event record between two kernels cost on this stack?  Back-to-back launches of one ~40 us kernel on one stream, with after every launch:
nothing / a torch.cuda.Event record / a raw HIP event record with hipEventDisableSystemFence / record + a second stream waiting on it (torch and raw).
Prints microseconds per iteration."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
rt = C.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventDisableSystemFence = 0x2, 0x20000000


class RawEvent:
    def __init__(self, flags):
        self.h = C.c_void_p()
        assert rt.hipEventCreateWithFlags(C.byref(self.h), C.c_uint(flags)) == 0

    def record(self, stream):
        assert rt.hipEventRecord(self.h, C.c_void_p(stream.cuda_stream)) == 0

    def wait(self, stream):
        assert rt.hipStreamWaitEvent(C.c_void_p(stream.cuda_stream), self.h, 0) == 0


M, D = 64 * 1569, 384
x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
u = torch.empty(M, D, dtype=torch.bfloat16, device="cuda"); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
u2 = torch.empty_like(u); mean2 = torch.empty_like(mean); rstd2 = torch.empty_like(rstd)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
kern = lambda: hip.ln_fwd(x, g, b, u, mean, rstd, M, D, 1e-6)


def kern_side():
    with torch.cuda.stream(side):
        hip.ln_fwd(x, g, b, u2, mean2, rstd2, M, D, 1e-6)


def run(name, body, n=300):
    for _ in range(20):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    print(f"{name:70s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per iteration", flush=True)


def torch_rec():
    kern(); torch.cuda.Event().record(main)


raw_pool = [RawEvent(hipEventDisableTiming | hipEventDisableSystemFence) for _ in range(64)]
raw_pool_sys = [RawEvent(hipEventDisableTiming) for _ in range(64)]
cnt = [0]


def raw_rec(pool):
    def f():
        kern(); cnt[0] += 1; pool[cnt[0] % 64].record(main)
    return f


def torch_fork():
    kern(); ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev); kern_side()


def raw_fork(pool):
    def f():
        kern(); cnt[0] += 1; ev = pool[cnt[0] % 64]; ev.record(main); ev.wait(side); kern_side()
    return f


run("kernel only", kern)
run("kernel + torch.cuda.Event record", torch_rec)
run("kernel + raw event record (DisableTiming)", raw_rec(raw_pool_sys))
run("kernel + raw event record (DisableTiming | DisableSystemFence)", raw_rec(raw_pool))
run("kernel, torch event, second stream waits and runs a kernel", torch_fork)
run("kernel, raw event (DisableTiming), second stream waits + kernel", raw_fork(raw_pool_sys))
run("kernel, raw event (DisableSystemFence), second stream waits + kernel", raw_fork(raw_pool))
