"""Unprofiled length of the region between the encoder's last forward attention kernel and its first backward attention kernel (the last block's
CLS-only tail, the head, the losses and their backward, the tail's backward): events around those two launches only, 10 steps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, diverse_channel_vit_amd as dcv
from diverse_channel_vit_amd import hip
dev = torch.device("cuda", 0)
cfg = bench.model_cfg("small", 8, 224, 16, 161)
torch.manual_seed(0)
model = dcv.dichavit(cfg, mapper={"train": list(range(8))}).to(dev).train()
opt = dcv.HipAdamW([p for p in model.parameters() if p.requires_grad], lr=4.9e-5, weight_decay=0.04, model=model)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((64, 8, 224, 224)).astype(np.float32)).to(dev); y = torch.from_numpy(rs.randint(0, 161, 64)).to(dev)
ce = torch.nn.CrossEntropyLoss()


def step():
    opt.zero_grad()
    out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    (ce(out, y) + extra).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
res = []
for _ in range(10):
    hip.set_profiler(True, only=["attn_fwd3_kernel<true>", "attn_bwd_dq2_kernel<true>", "attn_fwd3_kernel<false>", "attn_bwd_dq2_kernel<false>"])
    step()
    torch.cuda.synchronize()
    prof = hip.set_profiler(False)
    fwd = [ev for k, v in prof.items() if k[0].startswith("attn_fwd3_kernel") for ev in v["ev"]]
    bwd = [ev for k, v in prof.items() if k[0].startswith("attn_bwd_dq2_kernel") for ev in v["ev"]]
    # launches are recorded in issue order per key; the full-size forward kernels come first, the Nq=1 one (last block) last
    last_fwd_end = max(fwd, key=lambda e: fwd[0][0].elapsed_time(e[1]))[1]
    first_bwd_start = min(bwd, key=lambda e: fwd[0][0].elapsed_time(e[0]))[0]
    res.append(last_fwd_end.elapsed_time(first_bwd_start) * 1e3)
print(f"last attention forward end -> first attention backward start: median {np.median(res):.0f} us (min {min(res):.0f}, max {max(res):.0f}) over 10 unprofiled steps")
