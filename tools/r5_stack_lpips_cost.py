import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
from npp_amd.stack import StackedFit
dev = torch.device("cuda", 0); H, K, M = 512, 3, 8
angles, periods, shifts = syn.synthetic_periodicity(H, K)
def fits(lp):
    out = []
    for i in range(M):
        img, mask = syn.synthetic_image(H, seed=i)
        out.append(CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=i), device=dev, N_rand=8192, shifts=shifts, seed=i, rng_mode="fast", use_perceptual_loss=lp))
    return out
for lp in (True, False):
    st = StackedFit(fits(lp))
    for _ in range(10): st.step_full()
    ts = []
    for s in range(10):
        b = st.sample()
        for _ in range(3): st.step_from(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): st.step_from(b)
        torch.cuda.synchronize(); ts.append(((time.perf_counter() - t0) / 20 * 1e3, sum(x is not None and x["source"] == "same" for x in b)))
    print("LPIPS", lp, "ms per stacked iteration by number of 'same' images:", sorted((n, round(t, 3)) for t, n in ts), "mean", round(float(np.mean([t for t, _ in ts])), 3))
    st.close(); del st
