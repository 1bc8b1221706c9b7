import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, shifts=shifts, ksplit=12)
for _ in range(5): fit.sample_batch()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): fit.sample_batch()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
