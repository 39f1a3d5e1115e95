"""Which Python lines issue the step's SMALL device operations (fills, copies, elementwise glue)?  One headline train step under torch.profiler
with stacks; prints aten ops that are not the C-ABI kernels, grouped by the innermost frame inside this repository."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
cfg = bench.model_cfg("small", 8, 224, 16, 161)
torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((64, 8, 224, 224)).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randint(0, 161, 64)).to(dev)
ce = torch.nn.CrossEntropyLoss()


def step():
    opt.zero_grad()
    out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    (ce(out, y) + extra).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if not ev.kernels:
        continue
    frame = next((f for f in (ev.stack or []) if root in f and "tools/glue_ops" not in f), "(autograd / no repo frame)")
    key = (ev.name, frame.replace(root + "/", "").split(",")[0])
    agg[key] += len(ev.kernels)
    dur[key] += kt
tot = sum(agg.values())
print(f"aten ops with device work in one step: {tot} launches, {sum(dur.values()) / 1e3:.2f} ms of kernel time")
for k, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    print(f"  {n:4d} x {k[0]:28s} {dur[k]:8.1f} us  {k[1]}")
