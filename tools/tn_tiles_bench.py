"""Times the weight-gradient product at the headline step's shapes for every legal tile (narrow 128x128, wide 384x128; a 384x192 kernel with
XCD-aligned splits was measured with this script in round 3 and removed: profiles/r03_x7_*) in both reduction modes (fp32 atomics / deterministic partial slabs + fixed-order pass), interleaved in one process."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
lib = hip.load()
M = 64 * 1569
bf = torch.bfloat16
torch.manual_seed(0)
T = {d: torch.randn(M, d, device="cuda").to(bf) for d in (384, 1152, 1536)}
tiles = {"narrow": hip.TILE_NARROW, "wide": hip.TILE_WIDE}
rounds = int(os.environ.get("TB_ROUNDS", 10))
for (P, Q) in [(1536, 384), (384, 1536), (1152, 384), (384, 384)]:
    Y, X = T[P], (T[Q] if Q != P else torch.randn(M, Q, device="cuda").to(bf))
    dW = torch.zeros(P, Q, device="cuda"); db = torch.zeros(P, device="cuda")
    cases = [(n, t, det) for n, t in tiles.items() if lib.dcv_gemm_tn_pick(M, P, Q, t) == t for det in (False, True)]
    res = {c: [] for c in cases}
    for rnd in range(rounds):
        for c in cases:
            n, t, det = c
            hip.set_deterministic(det)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                hip.gemm_tn_acc(Y, X, dW, None if os.environ.get("TB_NOBIAS") else db, tile=t)
            e.record(); torch.cuda.synchronize()
            if rnd >= 2:
                res[c].append(s.elapsed_time(e) * 1e3 / 3)
    auto = {v: k for k, v in tiles.items()}[lib.dcv_gemm_tn_pick(M, P, Q, hip.TILE_AUTO)]
    print(f"P{P:5d} Q{Q:5d} (AUTO = {auto}): " + "  ".join(f"{n}/{'det' if det else 'atomic'} {np.median(v):6.1f}" for (n, t, det), v in res.items()), flush=True)
