"""Largest loop geometry: 1024^2 image (BASELINE c4 size), top-3, P = 160 patches (loaders.py:133-134 clip), CX on 40x40 maps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 1024, 3
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, shifts=shifts, ksplit=12, rng_mode="fast")
print("patch", fit.patch_size, "rows/iter", fit.N_rand + fit.patch_num * fit.patch_size ** 2)
for _ in range(20):
    fit.step_full()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
seen = {}
for _ in range(n):
    fit.step_full()
    seen[fit.last_source] = seen.get(fit.last_source, 0) + 1
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms/iter incl. sampling; sources {seen}; skipped {fit.skipped}; psnr known {fit.psnr():.2f} dB; "
      f"max mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
