"""CX core alone on loop-size features (6 x 256 x 24 x 24): for rocprofv3 kernel stats / PMC."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import ops
torch.manual_seed(0)
N, C, h = 6, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 24
fx = torch.relu(torch.randn(N, C, h, h, device="cuda"))
fy = torch.relu(torch.randn(N, C, h, h, device="cuda"))
loss = torch.zeros(1, device="cuda")
for _ in range(30):
    ops.cx_fwd_bwd(fx, fy, 0.5, None, 1e-3, loss, True)
torch.cuda.synchronize()
