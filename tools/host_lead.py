"""How far ahead of the device is the host at the start of every step of a free-running training loop (no synchronisation between steps)?
Per step: host time spent enqueueing it, and the host's lead = (device time at which the step's first kernel could have started at the
earliest, i.e. when the previous step's work ended) - (host time at which the step's enqueue began).  A lead near zero means the device
waits for the host at the step boundary."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
dev = torch.device("cuda", 0)
cfg = bench.model_cfg("small", 8, 224, 16, 161)
torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((64, 8, 224, 224)).astype(np.float32)).to(dev); y = torch.from_numpy(rs.randint(0, 161, 64)).to(dev)
ce = torch.nn.CrossEntropyLoss()
tm = {}


def step():
    t0 = time.perf_counter()
    opt.zero_grad()
    out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    t1 = time.perf_counter()
    loss = ce(out, y) + extra
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    return t0, t1, t2, t3


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
base = time.perf_counter()
ev[0].record()
rows = []
for i in range(N):
    rows.append(step())
    ev[i + 1].record()
torch.cuda.synchronize()
dev_end = [ev[0].elapsed_time(ev[i + 1]) for i in range(N)]  # ms after ev[0] at which step i finished on the device
print("step: host enqueue ms (forward / backward / optimiser) | host began at ms | device finished previous step at ms | host lead ms")
for i, (t0, t1, t2, t3) in enumerate(rows):
    began = 1e3 * (t0 - base)
    prev_end = dev_end[i - 1] if i else 0.0
    print(f"{i:3d}: {1e3 * (t1 - t0):6.2f} / {1e3 * (t2 - t1):6.2f} / {1e3 * (t3 - t2):6.2f} | {began:8.2f} | {prev_end:8.2f} | {prev_end - began:8.2f}")
print(f"device time per step {dev_end[-1] / N:.2f} ms; host enqueue per step {1e3 * np.mean([r[3] - r[0] for r in rows]):.2f} ms")
