"""Full-grid render (mlp_fwd inference kernel) alone: for rocprofv3 --pmc runs."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, _ = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, ksplit=12)
for _ in range(12):
    fit.net.render(fit.i_all_dev)
torch.cuda.synchronize()
