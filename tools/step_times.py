import sys, os, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench, diverse_channel_vit_amd as dcv
B = int(sys.argv[1]); dev = torch.device("cuda", 0)
cfg = bench.model_cfg("small", 8, 224, 16, 161); torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((B, 8, 224, 224)).astype(np.float32)).to(dev); y = torch.from_numpy(rs.randint(0, 161, B)).to(dev)
ce = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad(); out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0); (ce(out, y) + extra).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
host = []
marks[0].record()
for i in range(40):
    t0 = time.perf_counter(); step(); host.append(1e3 * (time.perf_counter() - t0)); marks[i + 1].record()
torch.cuda.synchronize()
gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(40)]
print("gpu ms :", " ".join(f"{v:5.1f}" for v in gpu))
print("host ms:", " ".join(f"{v:5.1f}" for v in host))
