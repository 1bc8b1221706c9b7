import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda", 0)
H, K = 512, 3
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, seed=0, shifts=shifts)
pool = []
while len(pool) < 40:
    b = fit.sample_batch()
    if b is not None: pool.append(b)
for i in range(40): fit.step_from(pool[i])
torch.cuda.synchronize()
for rep, idle in enumerate((0.0, 0.0, 0.5, 2.0, 0.0)):
    time.sleep(idle)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(20):
        fit.step_from(pool[i])
        ev[i + 1].record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
    print(f"rep {rep} idle {idle}: wall {dt / 20 * 1e3:.4f} ms/step, host enqueue {t_host / 20 * 1e3:.3f}; per step:", " ".join(f"{m:.2f}" for m in ms), [b["source"][0] for b in pool[:20]] if rep == 0 else "", flush=True)
