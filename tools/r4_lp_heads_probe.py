"""Round-4 probe: the grouped LPIPS heads launch (npp_lpips_layers) on VGG16-shaped taps of two 96^2 patches against each tap alone
(which tap sets the launch's duration), with and without the gradient.  usage: r4_lp_heads_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops            # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(11)
N, shapes = 2, [(64, 96), (128, 48), (256, 24), (512, 12), (512, 6)]
f0s = [torch.rand(N, C, h, h, generator=g).to(dev) for C, h in shapes]
f1s = [torch.rand(N, C, h, h, generator=g).to(dev) for C, h in shapes]
lins = [(torch.rand(C, generator=g) * 0.1).to(dev) for C, _ in shapes]
lats = [torch.cat([torch.randn(C, generator=g) * 0.5, torch.randn(C, generator=g) * 0.3]).to(dev) for C, _ in shapes]
dl = [torch.zeros_like(t) for t in lats]
df = [torch.empty_like(f) for f in f0s]
spline, n_knots, xs = ops.load_spline(dev)
loss = torch.zeros(1, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("all five, with gradient: %.1f us" % timed(lambda: ops.lpips_layers(f0s, f1s, lins, lats, spline, n_knots, xs, 0.7, loss, df, dl)))
print("all five, plain head   : %.1f us" % timed(lambda: ops.lpips_layers(f0s, f1s, lins, None, spline, n_knots, xs, 0.7, loss, df, None)))
for k, (C, h) in enumerate(shapes):
    sl = slice(k, k + 1)
    t = timed(lambda: ops.lpips_layers(f0s[sl], f1s[sl], lins[sl], lats[sl], spline, n_knots, xs, 0.7, loss, df[sl], dl[sl]))
    print(f"tap {k} ({C} ch, {h}^2) alone: {t:.1f} us")
