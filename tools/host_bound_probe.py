"""Is the complete iteration host-enqueue-bound?  Host time to enqueue (no sync) vs device completion time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, shifts=shifts, ksplit=12)
pool = []
while len(pool) < 10:
    b = fit.sample_batch()
    if b is not None and b["source"] != "same":
        pool.append(b)
for b in pool:
    fit.step_from(b)
torch.cuda.synchronize()
for reps in (1, 5, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for b in pool:
            fit.step_from(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = reps * len(pool)
    print(f"reps {reps}: host enqueue {(t1 - t0) / n * 1e3:.3f} ms/iter, total {(t2 - t0) / n * 1e3:.3f} ms/iter")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for r in range(5):
    for b in pool:
        fit.step_from(b)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
