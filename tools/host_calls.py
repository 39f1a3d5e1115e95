"""Host cost of every hip.* wrapper call in one training step (monkey-patched timers; includes the autograd thread):
python tools/host_calls.py [batch]"""
import sys, os, time, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
from diverse_channel_vit_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
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
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); d = time.perf_counter() - t0
        acc[name][0] += 1; acc[name][1] += d
        return r
    return w
for name in dir(hip):
    fn = getattr(hip, name)
    if callable(fn) and not name.startswith("_") and name not in ("load", "set_profiler") and getattr(fn, "__module__", "") == hip.__name__ and not isinstance(fn, type):
        setattr(hip, name, wrap(name, fn))
tt = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); tt.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = sum(v[1] for v in acc.values()) / 5
print(f"batch {B}: host enqueue {1e3 * np.median(tt):.2f} ms per step; inside hip.* wrappers {1e3 * tot:.2f} ms ({sum(v[0] for v in acc.values()) // 5} calls)")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} {n // 5:4d} calls  {1e6 * t / n:7.1f} us each  {1e3 * t / 5:6.2f} ms per step")
