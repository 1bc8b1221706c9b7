#!/usr/bin/env python3
"""Fused chain at both compiled widths: 512^2 K=3 render and one MLP-only training step (forward with stash, pixel loss,
backward chain, grouped wgrad, Adam + repack) on 26 624 rows, with MFMA fractions of the 2.5 PFLOP/s bf16 peak; the W = 512
numbers next to the unfused dense-layer path that served that width before.      python tools/width_time.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import synthetic as syn  # noqa: E402
from npp_amd.model import NPPNet  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 2.5e15


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


out = {}
H, K, B = 512, 3, 26624
yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
a, p, _ = syn.synthetic_periodicity(H, K)
for W in (256, 512):
    net = NPPNet(a, p, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0, width=W), device=dev, width=W)
    fwd_macs, train_macs = syn.mlp_macs_per_pixel(K, W)
    t_r = timed(lambda: net.render(grid))
    c = grid[torch.randint(0, H * H, (B,), device=dev)].contiguous()
    gt = torch.rand(B, 3, device=dev)

    def step():
        net.zero_grad()
        net.forward_train(c)
        net.pixel_loss(B, B, gt)
        net.backward(B)
        net.optimizer_step(B)
    net.workspace(B)["dpred"].zero_()
    t_s = timed(step, reps=20)
    t_f = timed(lambda: net.forward_train(c), reps=20)
    from npp_amd import ops
    ws = net.workspace(B)
    t_b = timed(lambda: ops.mlp_bwd(ws["dpred"], ws["pred"], net.K, net.wb, net.params, ws["actT"], ws["dzT"], W), reps=20)
    t_w = {}
    for ks in sorted({net.ksplit, 2, 3, 4, 6, 8, 12}):
        gs = torch.empty(ops.train_workspace(net.K, B, ks, W)[3] // 4, dtype=torch.float32, device=dev)
        t_w[ks] = timed(lambda: ops.mlp_wgrad(ws["dzT"], ws["actT"], B, net.K, ks, gs, W), reps=20) * 1e6
    t_a = timed(lambda: net.optimizer_step(B), reps=20)
    out[f"W{W}"] = {"ksplit": net.ksplit, "render_512sq_ms": t_r * 1e3, "render_mfma_frac": 2 * fwd_macs * H * H / t_r / PEAK,
                    "train_fwd_us": t_f * 1e6, "bwd_us": t_b * 1e6, "wgrad_us_by_ksplit": t_w, "adam_repack_us": t_a * 1e6,
                    "wgrad_tiles": int(ops.lib(W).npp_mlp_wgrad_tiles(K)), "train_fwd_mfma_frac": 2 * fwd_macs * B / t_f / PEAK,
                    "mlp_step_ms": t_s * 1e3, "mlp_step_mfma_frac": 2 * train_macs * B / t_s / PEAK, "rows_per_s": B / t_s}
    del net
print(json.dumps(out, indent=1))
