import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
from npp_amd.stack import StackedFit
M = 8
dev = torch.device("cuda", 0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fits = []
for i in range(M):
    img, mask = syn.synthetic_image(H, seed=i)
    fits.append(CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=i), device=dev, N_rand=8192,
                              shifts=shifts, seed=i, rng_mode="reference"))
st = StackedFit(fits)
for _ in range(20):
    st.step_full()
torch.cuda.synchronize()
# host time alone: enqueue 50 iterations, measure the time until the LAST enqueue returns, then the sync
t0 = time.perf_counter()
for _ in range(50):
    st.step_full()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host {1e3 * (t1 - t0) / 50:.3f} ms per stacked iteration, with the final sync {1e3 * (t2 - t0) / 50:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    st.step_full()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
