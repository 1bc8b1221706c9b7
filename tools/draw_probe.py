import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda:0")
H, K = 512, 3
img, mask = syn.synthetic_image(H)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts, seed=0, prefetch=0)
for _ in range(20): fit.draw_batch()
t=time.perf_counter()
for _ in range(300): fit.draw_batch()
print(f"draw_batch {1e3*(time.perf_counter()-t)/300:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fit.draw_batch()
pr.disable()
pstats.Stats(pr).strip_dirs().sort_stats("tottime").print_stats(14)
