"""Runs only the attention entries at the headline shape (for rocprofv3 --pmc passes).  argv: repeats [pair|all|ps]
(ps: the plain entries AND the pre-scaled-q entries on the same operand with its q part scaled — the counters of both forms side by side)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
B, N, H, D = 64, 1569, 6, 384
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16); o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
dO = torch.randn(B, N, D, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device="cuda")
what = sys.argv[2] if len(sys.argv) > 2 else "all"
qkv_ps = qkv.clone()
qkv_ps[:, :, :D] = (qkv[:, :, :D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    if what == "ps":
        hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125)
        hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, 0.125)
        hip.attn_fwd(qkv_ps, o, lse, B, N, H, 64, 0.125, prescaled=True)
        hip.attn_bwd(qkv_ps, o, dO, lse, delta, dqkv, B, N, H, 64, 0.125, prescaled=True)
        continue
    hip.attn_fwd(qkv, o, lse, B, N, H, 64, 0.125)
    if what in ("pair", "all"):
        hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, 0.125)
torch.cuda.synchronize()
