"""Round 5: what the workgroup quantisation of the fused MLP launches costs.  The forward / backward chains run
64-row workgroups, two resident per CU (512 slots on 256 CUs): 26 624 rows = 416 workgroups leave 96 CUs with ONE workgroup.  Timed
here: complete MLP-only steps (forward -> pixel loss -> backward -> weight gradients -> Adam) at row counts around that point, HIP
events between the launches, median of 30 steps.  If T(26 624) = T(32 768) the launch time is set by the CUs that carry two workgroups
and a balanced partition (104 rows per CU) could at best reach 0.8125 x."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn  # noqa: E402
from npp_amd.model import NPPNet          # noqa: E402

dev = torch.device("cuda", 0)
H, K = 512, 3
angles, periods, _ = syn.synthetic_periodicity(H, K)
fwd_macs, train_macs = syn.mlp_macs_per_pixel(K)
print("rows   WGs |  fwd us (frac)   bwd us (frac)   wgrad us (frac)  adam us | us per 1000 rows fwd / bwd / wgrad")
for n in (8192, 16384, 20480, 24576, 26624, 28672, 32768, 36864, 49152, 65536):
    net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0), device=dev)
    bp = ops.pad_rows(n)
    rng = np.random.RandomState(0)
    c = torch.from_numpy(np.stack([rng.randint(0, H, bp), rng.randint(0, H, bp)], 1).astype(np.int32)).to(dev)
    gt = torch.rand(n, 3, device=dev)
    ws = net.workspace(bp)
    ws["dpred"].zero_()
    names = ["fwd", "loss", "bwd", "wgrad", "adam"]
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(30)]
    for r in range(33):
        e = evs[r - 3] if r >= 3 else None
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
    fl = [2 * fwd_macs * n, 0, 2 * (train_macs - 2 * fwd_macs) * n, 2 * fwd_macs * n, 0]
    fr = [fl[i] / (t[i] * 1e-6) / 2.5e15 if fl[i] else 0 for i in range(5)]
    print(f"{n:6d} {bp // 64:4d} | {t[0]:7.1f} ({fr[0]:.3f})  {t[2]:7.1f} ({fr[2]:.3f})  {t[3]:7.1f} ({fr[3]:.3f})  {t[4]:6.1f} | "
          f"{t[0] / n * 1e3:.2f} / {t[2] / n * 1e3:.2f} / {t[3] / n * 1e3:.2f}   ksplit {net.ksplit}", flush=True)
    del net, ws
    torch.cuda.empty_cache()
