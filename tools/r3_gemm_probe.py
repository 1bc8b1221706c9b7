"""Round-3 probe: the stacked dense-layer launches of the ranking fit in a tight loop (9 x 2048 x 256 x 256)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
C, B, W = int(os.environ.get("R3_C", "9")), 2048, 256
x = torch.randn(C, B, W, device=dev); w = torch.randn(C, W, W, device=dev) / 16; b = torch.randn(C, W, device=dev)
y = torch.empty(C, B, W, device=dev); z = torch.empty(C, B, W, device=dev); dw = torch.zeros(C, W, W, device=dev); db = torch.zeros(C, W, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fl = 2.0 * C * B * W * W
for name, fn in (("fwd snake+z", lambda: ops.linear_fwd_batched(x, w, b, 1, y, z)), ("fwd linear ", lambda: ops.linear_fwd_batched(x, w, b, 0, y)),
                 ("bwd data   ", lambda: ops.linear_bwd_data_batched(x, w, y)), ("bwd weight ", lambda: ops.linear_bwd_weight_batched(x, y, dw, db))):
    us = t(fn)
    print(f"{name}: {us:7.1f} us  {fl / us * 1e-6:6.1f} TF  ({fl / us * 1e-6 / 157.3:.2f} of the fp32 MFMA peak)")
