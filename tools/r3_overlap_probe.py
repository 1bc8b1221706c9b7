"""Round-3 probe: complete iteration with the pixel rows' path on a side stream under the patch-loss chain
(CompletionFit.overlap) against the single-stream sequence, same pool, interleaved rounds in one process."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn          # noqa: E402
from npp_amd.fit import CompletionFit              # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, K)


def make():
    return CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, seed=0, shifts=shifts)


fit = make()
quota = {"val": 10, "train": 6, "same": 4}
pool = []
while len(pool) < 20:
    b = fit.sample_batch()
    if b is not None and quota[b["source"]] > 0:
        quota[b["source"]] -= 1
        pool.append(b)

# ---- correctness: one step from identical state, split vs unsplit: gradients and updated parameters
fa, fb = make(), make()
fa.overlap, fb.overlap = False, True
for f in (fa, fb):
    f.step_from(pool[0])
torch.cuda.synchronize()
ga, gb = fa.net.grads(), fb.net.grads()
worst = max(float(np.linalg.norm(ga[k_] - gb[k_]) / (np.linalg.norm(ga[k_]) + 1e-30)) for k_ in ga)
print("split vs unsplit: worst rel-L2 gradient difference", worst, " params max |d|", float((fa.net.params - fb.net.params).abs().max()),
      " loss", float(fa.net.loss_buf), float(fb.net.loss_buf))


PIPE = os.environ.get("R3_PIPE", "0") == "1"


def run(mode, ks, n=100):
    fit.overlap = mode
    fit.overlap_ks = ks
    for i in range(len(pool)):
        fit.step_from(pool[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fit.step_from(pool[i % len(pool)], pool[(i + 1) % len(pool)] if PIPE else None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


variants = [(False, (4, 12)), (True, (4, 12)), (True, (3, 12)), (True, (2, 12)), (True, (4, 8)), (True, (4, 10)), (True, (6, 12)), (True, (3, 9))]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        res[v].append(run(*v))
for v in variants:
    print(("overlap ks=%d/%d" % v[1]) if v[0] else "single stream  ", " ms/iter: ", " ".join(f"{x:.4f}" for x in res[v]), "  min", f"{min(res[v]):.4f}")

# per source
for mode in (False, True):
    fit.overlap, fit.overlap_ks = mode, (4, 12)
    out = {}
    for src in ("val", "train", "same"):
        bs = [b for b in pool if b["source"] == src]
        for b in bs:
            fit.step_from(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            for b in bs:
                fit.step_from(b)
        torch.cuda.synchronize()
        out[src] = (time.perf_counter() - t0) / (10 * len(bs)) * 1e3
    print("overlap" if mode else "single ", {k_: round(v_, 4) for k_, v_ in out.items()})
