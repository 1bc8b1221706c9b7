"""Device time of one complete iteration against the patch size (the loaders' rule gives 64 ... 160 from the detected period):
python tools/patch_size_time.py [width]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.simplefilter("ignore")
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H, K = 512, 3
img, mask = syn.synthetic_image(H)
a, p, sh = syn.synthetic_periodicity(H, K)
for P in (64, 96, 128, 160):
    fit = CompletionFit(img, mask, a, p, syn.SEED0_FREQS, syn.init_params(K, seed=0, width=W), device=dev, N_rand=8192, seed=0, shifts=sh,
                        patch_size=P, width=W)
    by = {}
    for _ in range(200):
        b = fit.sample_batch()
        if b is not None and b["k"] == 3 or (b is not None and b["source"] == "same"):
            by.setdefault(b["source"], b)
        if len(by) == 3:
            break
    out = []
    for src in ("val", "train", "same"):
        b = by[src]
        for _ in range(5):
            fit.step_from(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            fit.step_from(b)
        torch.cuda.synchronize(); out.append((src, (time.perf_counter() - t0) / 30 * 1e3, b["n"]))
    mix = 0.5 * out[0][1] + 0.3 * out[1][1] + 0.2 * out[2][1]
    print(f"W={W} P={P}: rows {out[0][2]}, " + ", ".join(f"{s} {t:.3f} ms" for s, t, _ in out) + f"; 50/30/20 mix {mix:.3f} ms = {out[0][2] / mix / 1e3:.1f} M rows/s", flush=True)
