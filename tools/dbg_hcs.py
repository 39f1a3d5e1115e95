import sys, os, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import diverse_channel_vit_amd as dcv
from bench import model_cfg
dev = torch.device("cuda:0")
cfg = model_cfg(); cfg.update(enable_sample=True, hcs_sampling="lowest_cosine_prob", hcs_sampling_temp=1000.0)
random.seed(1); torch.manual_seed(1)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(1234)
x = torch.from_numpy(rs.standard_normal((64, 8, 224, 224)).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randint(0, 161, 64)).to(dev)
ce = torch.nn.CrossEntropyLoss()
pe = model.feature_extractor.patch_embed
for s in range(14):
    before = sum(pe.counter.values())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(); out, extra = model(x, "train", None); loss = ce(out, y) + extra; loss.backward(); opt.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {s}: C={sum(pe.counter.values())-before} {dt*1e3:.1f} ms", flush=True)
