"""In-sequence times of the MLP launches of MLP-only steps at c2's row count (HIP events between the launches, median of 40 steps):
   python tools/r6_fwd_time.py        (NPP_LIB_PATH=build_ab/libnpp_<variant>.so, NPP_STASH8=0/1)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn  # noqa: E402
from npp_amd.model import NPPNet          # noqa: E402

dev = torch.device("cuda", 0)
H, K, n = 512, 3, 26624
angles, periods, _ = syn.synthetic_periodicity(H, K)
net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0), device=dev,
             ksplit=int(os.environ['R6_KSPLIT']) if os.environ.get('R6_KSPLIT') else None)
bp = ops.pad_rows(n)
rng = np.random.RandomState(0)
c = torch.from_numpy(np.stack([rng.randint(0, H, bp), rng.randint(0, H, bp)], 1).astype(np.int32)).to(dev)
gt = torch.rand(n, 3, device=dev)
ws = net.workspace(bp)
ws["dpred"].zero_()
R = 40
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(R)]
for r in range(R + 5):
    e = evs[r - 5] if r >= 5 else None
    net.zero_grad()
    if e: e[0].record()
    net.forward_train(c)
    if e: e[1].record()
    net.pixel_loss(bp, n, gt)
    if e: e[2].record()
    ops.mlp_bwd(ws["dpred"], ws["pred"], K, net.wb, net.params, ws["actT"], ws["dzT"])
    if e: e[3].record()
    ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, K, net.ksplit, ws["gslabs"])
    if e: e[4].record()
    net.optimizer_step(bp)
    if e: e[5].record()
torch.cuda.synchronize()
t = np.median(np.array([[e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(5)] for e in evs]), 0)
print(f"{os.environ.get('NPP_LIB_PATH', 'default'):32s} stash8={ops.tune('stash8')} ksplit={net.ksplit}  fwd {t[0]:6.1f}  loss {t[1]:5.1f}  bwd {t[2]:6.1f}  wgrad {t[3]:6.1f}  adam {t[4]:5.1f} us")
