"""End-to-end wall time of the complete loop INCLUDING host-side sampling, per RNG mode / prefetch depth."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
for mode, pf in (("numpy", 0), ("reference", 0), ("reference", 4), ("fast", 0), ("fast", 4)):
    fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, shifts=shifts, ksplit=12,
                        rng_mode=mode, prefetch=pf)
    for _ in range(30):
        fit.step_full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        fit.step_full()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:9s} prefetch {pf}: {dt / n * 1e3:.3f} ms/iteration end to end ({n} iterations incl. sampling), psnr {fit.psnr():.2f} dB, skipped {fit.skipped}")
    fit.close()
