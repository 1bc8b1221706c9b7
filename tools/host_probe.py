import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda:0")
H, K = 512, 3
img, mask = syn.synthetic_image(H)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts, seed=0, prefetch=4)
for _ in range(100): fit.step_full()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400): fit.step_full()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/400:.3f} ms/iter, total {1e3*(t2-t0)/400:.3f} ms/iter (GPU backlog at the end {1e3*(t2-t1):.1f} ms)")
# where the host time goes
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fit.step_full()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.strip_dirs().sort_stats("tottime").print_stats(30)
fit.close()
