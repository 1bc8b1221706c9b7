"""Round-4 probe: UPPER BOUND on what folding the four max-pool launches of the VGG19 trunk (two forward, two backward) into their
neighbouring convolutions could save -- the same iterations timed with the pool launches simply skipped (results wrong, timing only;
the switch lives in this script, not in the product).  A fold adds epilogue / loader work to a convolution, so the real gain is smaller.
    python3 tools/r4_pool_bound.py [val|train|same] [iters]"""
import os
import sys
import time

import torch

os.environ.setdefault("NPP_LP_GRAPH", "0")          # a captured LPIPS graph would keep its pool nodes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn            # noqa: E402
from npp_amd.fit import CompletionFit                # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "val"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
img, mask = syn.synthetic_image(H, seed=0)
f = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts, seed=0,
                  rng_mode="fast", use_perceptual_loss=True)
pool = []
while len(pool) < 8:
    b = f.sample_batch()
    if b is not None and b["source"] == src:
        pool.append(b)
real = (ops.maxpool2_fwd, ops.maxpool2_bwd)
noop = (lambda *a, **k: None, lambda *a, **k: None)


def timed(fns):
    ops.maxpool2_fwd, ops.maxpool2_bwd = fns
    for i in range(10):
        f.step_from(pool[i % len(pool)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        f.step_from(pool[i % len(pool)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for rep in range(3):
    a = timed(real)
    b = timed(noop)
    print(f"{src}: with pools {a:.4f} ms, pool launches skipped {b:.4f} ms  (bound {1e3 * (a - b):.1f} us)", flush=True)
ops.maxpool2_fwd, ops.maxpool2_bwd = real
