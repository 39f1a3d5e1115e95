"""Host time to ENQUEUE one training step (eager launches) against the GPU time of the step: python tools/host_overhead.py [batch]
The host must stay ahead of the GPU; this prints by how much, and the top Python-level costs (cProfile) of one enqueued step."""
import cProfile, pstats, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
cfg = bench.model_cfg("small", 8, 224, 16, 161)
torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((B, 8, 224, 224)).astype(np.float32)).to(dev); y = torch.from_numpy(rs.randint(0, 161, B)).to(dev)
ce = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad()
    out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    (ce(out, y) + extra).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
host = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); host.append(time.perf_counter() - t0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / 10
print(f"batch {B}: host enqueue {1e3 * np.median(host):.2f} ms per step (GPU idle at start), steady-state step {1e3 * gpu:.2f} ms")
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
