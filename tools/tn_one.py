"""One weight-gradient product at the headline shape, REPS launches (for rocprofv3 --pmc passes; DCV_LIB picks the build):
python tools/tn_one.py [P Q reps]   (default fc2's: P = 384, Q = 1536)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diverse_channel_vit_amd import hip
hip.load()
hip.set_deterministic(False)
P, Q, reps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (384, 1536, 6)
M = 64 * 1569
torch.manual_seed(0)
Y = torch.randn(M, P, device="cuda").to(torch.bfloat16); X = torch.randn(M, Q, device="cuda").to(torch.bfloat16)
dW = torch.zeros(P, Q, device="cuda"); db = torch.zeros(P, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    hip.gemm_tn_acc_group([(Y, X, dW, db)])  # the grouped entry: the only one that picks the 384 x 256 tile in the variant build
torch.cuda.synchronize()
ref = Y.float().t() @ X.float()
print("max rel err", float((dW / reps - ref).abs().max() / ref.abs().max()))
