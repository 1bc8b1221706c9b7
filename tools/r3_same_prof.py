"""Profile target: 40 complete iterations of config c2 whose patch source is 'same' (LPIPS + contextual):
rocprofv3 --kernel-trace -- python tools/r3_same_prof.py [val]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
want = sys.argv[1] if len(sys.argv) > 1 else "same"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
H, K = 512, 3
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, seed=0, shifts=shifts)
pool = []
while len(pool) < 10:
    b = fit.sample_batch()
    if b is not None and b["source"] == want:
        pool.append(b)
for i in range(40):
    fit.step_from(pool[i % len(pool)])
torch.cuda.synchronize()
fit.close()
