"""Round 5: where the end-to-end loop with the reference's random stream loses time against the device-only loop.
 (a) producer: CompletionFit.draw_batch() alone in a loop on the host (the native MT19937 stream: np.random.uniform, the centre choice =
     a full shuffle of the pool, the pixel draw = a full shuffle of the 245 k known pixels) -- what ONE producer thread can deliver;
 (b) the sampler's device half (materialise_batch: one upload, one crop gather, one row assembly) timed alone;
 (c) step_from over pre-drawn batches (device-only), step_full with prefetch (reference stream) and in fast mode."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn          # noqa: E402
from npp_amd.fit import CompletionFit         # noqa: E402

dev = torch.device("cuda", 0)
H, K = 512, 3
img, mask = syn.synthetic_image(H)
angles, periods, shifts = syn.synthetic_periodicity(H, K)


def make(**kw):
    return CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts, seed=0, **kw)


f = make(rng_mode="reference")
for _ in range(20):
    f.draw_batch()
t0 = time.perf_counter()
ds = [f.draw_batch() for _ in range(300)]
t_draw = (time.perf_counter() - t0) / 300
print(f"(a) producer alone, reference stream: {t_draw * 1e3:.3f} ms per draw_batch ({sum(d['k'] > 0 for d in ds)} of 300 with k > 0)")
ff = make(rng_mode="fast")
t0 = time.perf_counter()
for _ in range(300):
    ff.draw_batch()
print(f"    producer alone, fast mode:        {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per draw_batch")
ds = [d for d in ds if d["k"] > 0]
for d in ds[:10]:
    f.materialise_batch(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
bs = [f.materialise_batch(d) for d in ds[10:110]]
t_enq = (time.perf_counter() - t0) / 100
torch.cuda.synchronize()
t_mat = (time.perf_counter() - t0) / 100
print(f"(b) sampler device half: {t_mat * 1e3:.3f} ms per batch alone (host enqueue {t_enq * 1e3:.3f} ms)")
for b in bs[:40]:
    f.step_from(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(5):
    for b in bs[:40]:
        f.step_from(b)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 200
print(f"(c) device-only loop over 40 pre-drawn batches: {t_dev * 1e3:.4f} ms per iteration")
for name, kw in (("reference stream, producer thread (prefetch 4)", dict(rng_mode="reference", prefetch=4)),
                 ("fast mode, producer thread (prefetch 8)", dict(rng_mode="fast", prefetch=8)),
                 ("reference stream, producer thread (prefetch 16)", dict(rng_mode="reference", prefetch=16)),
                 ("reference stream, serial (no producer)", dict(rng_mode="reference", prefetch=0)),
                 ("fast mode", dict(rng_mode="fast"))):
    g = make(**kw)
    for _ in range(30):
        g.step_full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.step_full()
    torch.cuda.synchronize()
    print(f"    end to end, {name}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per iteration")
    g.close()
