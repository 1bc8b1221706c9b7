"""Round-3 probe: real-half prefetch (step_from(b, next_b)) against the plain step, same pool, interleaved; + equality check."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
H, K = 512, 3
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
make = lambda: CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, seed=0, shifts=shifts)
fit = make()
quota = {"val": 20, "train": 12, "same": 8}
pool = []
while len(pool) < 40:
    b = fit.sample_batch()
    if b is not None and quota[b["source"]] > 0:
        quota[b["source"]] -= 1; pool.append(b)
fa, fb = make(), make()
for i in range(12):
    fa.step_from(pool[i]); fb.step_from(pool[i], pool[i + 1])
torch.cuda.synchronize()
pa, pb = fa.net.params.cpu().numpy(), fb.net.params.cpu().numpy()
print("12 steps, plain vs prefetch: rel-L2 of the parameters", float(np.linalg.norm(pa - pb) / np.linalg.norm(pa)))
def run(pipe, n=120):
    for i in range(len(pool)): fit.step_from(pool[i], pool[(i + 1) % len(pool)] if pipe else None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fit.step_from(pool[i % len(pool)], pool[(i + 1) % len(pool)] if pipe else None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {False: [], True: []}
for r in range(3):
    for pipe in (False, True): res[pipe].append(run(pipe))
for pipe in (False, True): print("prefetch" if pipe else "plain   ", " ".join(f"{v:.4f}" for v in res[pipe]))
for pipe in (False, True):
    out = {}
    for src in ("val", "train", "same"):
        bs = [b for b in pool if b["source"] == src]
        f = lambda: [fit.step_from(b, bs[(j + 1) % len(bs)] if pipe else None) for j, b in enumerate(bs)]
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); out[src] = round((time.perf_counter() - t0) / (5 * len(bs)) * 1e3, 4)
    print("prefetch" if pipe else "plain   ", out)
