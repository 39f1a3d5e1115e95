import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import diverse_channel_vit_amd as dcv
from bench import model_cfg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = model_cfg()
cfg["ortho_loss_v1_lambda"] = float(os.environ.get("LO", "0.001")); cfg["proxy_loss_lambda"] = float(os.environ.get("LP", "0.001"))
torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model, capturable=True)
rs = np.random.RandomState(1234)
x = torch.from_numpy(rs.standard_normal((B, 8, 224, 224)).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randint(0, 161, B)).to(dev)
ce = torch.nn.CrossEntropyLoss()
for s in range(3):
    opt.advance(); opt.zero_grad()
    out, extra = model(x, "train", None); loss = ce(out, y) + extra; loss.backward(); opt.step()
    print("eager", s, loss.item(), extra.item(), flush=True)
del out, extra, loss
gs = dcv.GraphedTrainStep(model, opt, "train", None, ce, 1.0)
for s in range(4):
    l = gs(x, y)
    torch.cuda.synchronize()
    o = gs.static_out
    print("graph", s, l.item(), "extra", gs.static_extra.item(), "out absmax", o.abs().max().item(), "finite", bool(torch.isfinite(o).all()),
          "ce", ce(o, y).item(), "|arena|", model._arena.abs().max().item(), flush=True)
model.eval()
with torch.no_grad():
    o = model(x, "train", None)
print("eager eval after replays: ce", ce(o, y).item(), "absmax", o.abs().max().item())
