"""Full-loop probe (device side of train.py:183-264 with pre-drawn sampler output): per-source ms/iteration, host
sampling cost, and a trunk / CX breakdown.  Run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
from npp_amd import ops

H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, shifts=shifts, ksplit=12)
print("patch", fit.patch_size)
batches = []
while len(batches) < 10:
    b = fit.sample_batch()
    if b is not None:
        batches.append(b)
print([b["source"] for b in batches], [b["k"] for b in batches])
for b in batches:
    fit.step_from(b)
torch.cuda.synchronize()
for src in ("val", "train", "same"):
    bs = [b for b in batches if b["source"] == src]
    if not bs:
        continue
    t0 = time.perf_counter()
    for r in range(10):
        for b in bs:
            fit.step_from(b)
    torch.cuda.synchronize()
    print(src, "ms/step", (time.perf_counter() - t0) / (10 * len(bs)) * 1e3)
t0 = time.perf_counter()
for r in range(20):
    fit.sample_batch()
torch.cuda.synchronize()
print("sample_batch ms", (time.perf_counter() - t0) / 20 * 1e3)
P = fit.patch_size
x = torch.rand(6, 3, P, P, device="cuda")
y = torch.rand(6, 3, P, P, device="cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tr = fit.contextualLoss.hip_trunk
xy = torch.cat([x, y])
f = tr(xy)[0]
print("vgg19 fwd 12 imgs ms", t(lambda: tr(xy)))
fx, fy = f[:6].contiguous(), f[6:].contiguous()
print("cx core fwd+bwd ms", t(lambda: ops.cx_fwd_bwd(fx, fy)))
xr = x.clone().requires_grad_(True)


def full():
    xr.grad = None
    fit.contextualLoss(xr, y).backward()


print("CX module fwd+bwd ms", t(full))
