"""Timing of the fused exact-fp32 render (npp_mlp_fwd32) at 1024^2, K = 3, against the bf16 chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from npp_amd import synthetic as syn
from npp_amd.model import NPPNet
H, K = 1024, 3
dev = torch.device("cuda:0")
a, p, _ = syn.synthetic_periodicity(H, K)
net = NPPNet(a, p, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0), device=dev)
yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
fwd_macs, _ = syn.mlp_macs_per_pixel(K)
for name, fn in (("fp32_fused", net.render_fp32), ("bf16_fused", net.render)):
    fn(grid); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn(grid)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(name, f"{dt*1e3:.2f} ms  {H*H/dt/1e6:.1f} Mpx/s  {2*fwd_macs*H*H/dt/1e12:.1f} TFLOP/s")
