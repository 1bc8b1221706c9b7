"""Wall time of the MLP-only step (bench.py's mlp_only_step loop) -- no events between the launches."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn
from npp_amd.model import NPPNet
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0), device=dev)
bp = 26624
yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
c = grid[torch.randint(0, H * H, (bp,), device=dev)].contiguous(); gt = torch.rand(bp, 3, device=dev)
net.workspace(bp)["dpred"].zero_()
def step():
    net.zero_grad(); net.forward_train(c); net.pixel_loss(bp, bp, gt); net.backward(bp); net.optimizer_step(bp)
out = []
for rep in range(3):
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 200 * 1e6)
print(os.environ.get("NPP_LIB_PATH", "in-tree").split("/")[-1], "fused_repack" if net.fused_repack else "adam+pack", " step us:", " ".join(f"{v:.1f}" for v in out))
