"""Diagnostic (GPU box): after ONE HipAdamW step, compare the updated parameters with the oracle's and
evaluate batch 1 with both parameter sets through both forwards."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dichavit_oracle as orc
import diverse_channel_vit_amd as dcv
z = np.load(os.path.join(ROOT, "tests/golden/curve100_so2sat_s.npz")); meta = json.loads(str(z["meta"])); ref = z["losses"][:, 0]
class Cfg(dict): __getattr__ = dict.get
dev = torch.device("cuda:0")
cfg = Cfg(meta["cfg"], in_channel_names=[f"c{i}" for i in range(18)], img_size=[32], num_classes=17)
model = dcv.dichavit(cfg, mapper={"train": list(range(18))})
shapes = orc.state_shapes(meta["cfg"], 18, 32, 17)
st = orc.make_state(shapes, meta["seed"])
model.load_state_dict({**st, "adaptive_interface.0": st["proxies"]}); model = model.to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=meta["lr"], weight_decay=meta["wd"], model=model)
batches = [orc.make_batch(meta["seed"] + 100 + i, meta["B"], 18, 32, 17) for i in range(4)]
ch = list(range(18))
# oracle step 0 (fp32 like the reference)
sd = {k: v.clone().requires_grad_(k != "proxies") for k, v in st.items()}
x, y = batches[0]
loss, _, _, _ = orc.train_loss(sd, x, y, meta["cfg"], ch, ch); loss.backward()
g_ref = {k: sd[k].grad.clone() for k in sd if sd[k].grad is not None}
with torch.no_grad():
    for k in g_ref:
        orc.adamw_step(sd[k], sd[k].grad, torch.zeros_like(sd[k]), torch.zeros_like(sd[k]), 1, meta["lr"], 0.9, 0.999, 1e-8, meta["wd"])
print("oracle step0 loss", loss.item(), "golden", ref[0])
# hip step 0
opt.zero_grad()
out, extra = model(x.to(dev), "train", None); l0 = torch.nn.functional.cross_entropy(out, y.to(dev)) + extra; l0.backward()
g_hip = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None and not k.startswith("adaptive")}
opt.step(); torch.cuda.synchronize()
print("hip step0 loss", l0.item())
tot_flip = tot_n = 0
worst = []
for k, p in model.named_parameters():
    if k.startswith("adaptive") or k not in g_ref: continue
    d_h = (p.detach().cpu() - st[k]); d_r = (sd[k].detach() - st[k])
    flip = ((d_h * d_r) < 0).float().mean().item()
    rel = (d_h - d_r).norm().item() / (d_r.norm().item() + 1e-30)
    tot_flip += flip * p.numel(); tot_n += p.numel()
    worst.append((rel, flip, k, p.numel()))
worst.sort(reverse=True)
print("overall sign-flip fraction of the update:", tot_flip / tot_n)
for w in worst[:12]: print("  rel upd err %.3f flip %.3f %s (%d)" % w)
# evaluate batch 1 four ways
x1, y1 = batches[1]
def ofwd(params):
    with torch.no_grad():
        l, _, _, _ = orc.train_loss({k: v.detach() for k, v in params.items()}, x1, y1, meta["cfg"], ch, ch)
    return l.item()
hip_params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k != "adaptive_interface.0"}
print("batch1 loss: oracle-fwd(oracle params) %.5f | oracle-fwd(hip params) %.5f | golden %.5f" % (ofwd(sd), ofwd(hip_params), ref[1]))
with torch.no_grad():
    model.train()
    o, e = model(x1.to(dev), "train", None); lh = torch.nn.functional.cross_entropy(o, y1.to(dev)) + e
    print("             hip-fwd(hip params) %.5f" % lh.item())
    model.load_state_dict({**{k: v.detach() for k, v in sd.items()}, "adaptive_interface.0": sd["proxies"].detach()})
    o, e = model(x1.to(dev), "train", None); lh = torch.nn.functional.cross_entropy(o, y1.to(dev)) + e
    print("             hip-fwd(oracle params) %.5f" % lh.item())
# gradient error structure
for k in ["feature_extractor.blocks.0.attn.qkv.weight", "feature_extractor.blocks.11.mlp.fc2.weight", "feature_extractor.blocks.5.mlp.fc1.weight", "feature_extractor.patch_embed.proj.weight"]:
    a, b = g_hip[k].double(), g_ref[k].double()
    print(k, "rel", ((a - b).norm() / b.norm()).item(), "flip", ((a * b) < 0).double().mean().item(), "rms", b.pow(2).mean().sqrt().item(), "median|g|", b.abs().median().item())
