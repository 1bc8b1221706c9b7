"""Round-3 probe: does the training stash's residency in the 256 MiB Infinity Cache matter?
In-sequence per-kernel times (HIP events between the launches of whole MLP-only steps) against the batch size:
the stash is ~15 KB/row, so 8192 rows = 123 MB (resident), 16384 = 246 MB (edge), 26624 = 400 MB (not).
Per-row times are compared at row counts that fill whole rounds of workgroups (multiples of 512 x 64 / 2 ...) so that the
workgroup quantisation does not hide the effect:  python tools/r3_l3_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn          # noqa: E402
from npp_amd.model import NPPNet                   # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
WIDTH = int(os.environ.get("R3_WIDTH", "256"))
net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0, width=WIDTH), device=dev, width=WIDTH)
yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)


def seq_times(bp, ksplit, reps=30):
    net.ksplit = ksplit
    net._ws = {}
    c = grid[torch.randint(0, H * H, (bp,), device=dev)].contiguous()
    gt = torch.rand(bp, 3, device=dev)
    ws = net.workspace(bp)
    ws["dpred"].zero_()
    names = ["fwd", "loss", "bwd", "wgrad", "adam", "pack"]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(reps)]

    def one(e):
        net.zero_grad()
        if e: e[0].record()
        ops.mlp_fwd(c, net.cfg, net.wf, net.params, ws["pred"], ws["actT"], net.width)
        if e: e[1].record()
        net.pixel_loss(bp, bp, gt)
        if e: e[2].record()
        ops.mlp_bwd(ws["dpred"], ws["pred"], net.K, net.wb, net.params, ws["actT"], ws["dzT"], net.width)
        if e: e[3].record()
        ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, net.K, net.ksplit, ws["gslabs"], net.width)
        if e: e[4].record()
        net.opt_step += 1
        idle = net._loss_bufs[1 - net._loss_idx:2 - net._loss_idx]
        ops.adam_step_net(net.params, net.m, net.v, ws["gslabs"], net.ksplit, ws["gslabs"].numel() // net.ksplit, net.latents,
                          net.lat_m, net.lat_v, net.dlatent, idle, 0.0, net.opt_step)
        net._clean = True
        if e: e[5].record()
        net.repack()
        if e: e[6].record()
    for _ in range(5):
        one(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one(None)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    for r in range(reps):
        one(ev[r])
    torch.cuda.synchronize()
    t = np.array([[ev[r][i].elapsed_time(ev[r][i + 1]) * 1e3 for i in range(len(names))] for r in range(reps)])
    med = np.median(t, 0)
    return wall, dict(zip(names, med))


print("rows    wgs  ks  stash_MB |  step_us  us/krow |  fwd   loss   bwd   wgrad  adam  pack  (in-sequence, event-bracketed)")
cases = ((8192, 12), (16384, 12), (24576, 12), (26624, 12), (32768, 12), (49152, 12), (65536, 12), (8192, 4), (16384, 6))
if os.environ.get("R3_ROWS"):
    cases = tuple((int(v), 12) for v in os.environ["R3_ROWS"].split(","))
for bp, ks in cases:
    wall, t = seq_times(bp, ks)
    print(f"{bp:6d} {bp // 64:5d} {ks:3d} {bp * 15.2e-3:8.0f} | {wall:8.1f} {wall / bp * 1e3:7.2f} | "
          + " ".join(f"{t[k]:6.1f}" for k in t)
          + "   per-krow: " + " ".join(f"{k}={t[k] / bp * 1e3:.2f}" for k in ("fwd", "bwd", "wgrad")))
