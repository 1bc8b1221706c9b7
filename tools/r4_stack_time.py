"""Round-4 probe: M stacked images per launch sequence (npp_amd.stack.StackedFit) against the single-image loop, c2 shape
(512^2, K = 3, W = 256, 8192 pixel rows + 2 x 96^2 patch rows per image and iteration).
  device-only: fixed pre-drawn batch sets (each with its own random mix of patch sources), ms per stacked iteration
  e2e:         step_full() including every image's host draw (fast generator) and the sampler's device launches
    python tools/r4_stack_time.py [M ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn                 # noqa: E402
from npp_amd.fit import CompletionFit                # noqa: E402
from npp_amd.stack import StackedFit                 # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = int(os.environ.get("R4_H", "512")), int(os.environ.get("R4_K", "3"))
Ms = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
SETS, REPS = int(os.environ.get("R4_SETS", "10")), int(os.environ.get("R4_REPS", "20"))


def fits(M):
    angles, periods, shifts = syn.synthetic_periodicity(H, K)
    out = []
    for i in range(M):
        img, mask = syn.synthetic_image(H, seed=i)
        out.append(CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=i), device=dev, N_rand=8192,
                                 shifts=shifts, seed=i, rng_mode="fast"))
    return out


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# single-image loop (CompletionFit.step_from), the same way
f1 = fits(1)[0]
for _ in range(5):
    f1.step_full()
ts = []
for s in range(SETS):
    b = None
    while b is None:
        b = f1.sample_batch()
    for _ in range(3):
        f1.step_from(b)
    ts.append((b["source"], timed(lambda: f1.step_from(b), REPS)))
rows = f1.N_rand + f1.patch_num * f1.patch_size ** 2
single = float(np.mean([t for _, t in ts]))
print(f"single-image loop   : {single:.4f} ms / iteration (mix {[s for s, _ in ts]}) = {rows / single / 1e3:.2f} M rows/s")
e2e1 = timed(lambda: f1.step_full(), 100)
print(f"single-image e2e    : {e2e1:.4f} ms / iteration = {rows / e2e1 / 1e3:.2f} M rows/s")
for M in Ms:
    st = StackedFit(fits(M))
    for _ in range(5):
        st.step_full()
    ts = []
    for s in range(SETS):
        b = st.sample()
        for _ in range(3):
            st.step_from(b)
        ts.append(timed(lambda: st.step_from(b), REPS))
    t = float(np.mean(ts))
    e2e = timed(lambda: st.step_full(), 100)
    print(f"stacked M = {M} (ks {st.ksplit}): {t:.4f} ms / stacked iteration = {t / M:.4f} per image = {M * rows / t / 1e3:.2f} M rows/s "
          f"({single * M / t:.2f} x the single-image loop); e2e {e2e:.4f} ms = {M * rows / e2e / 1e3:.2f} M rows/s ({e2e1 * M / e2e:.2f} x)")
    st.close()
    del st
