#!/usr/bin/env python3
"""A complete default-length fit (2001 iterations, NPP_completion defaults) of the synthetic 512^2 lattice with K = 3 at both
compiled widths: wall time incl. host sampling, PSNR on known / unknown pixels along the way.   python tools/long_fit.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import synthetic as syn  # noqa: E402
from npp_amd.fit import CompletionFit  # noqa: E402
from npp_amd.io import patch_size_from_period  # noqa: E402

dev = torch.device("cuda:0")
H, K = 512, 3
img, mask = syn.synthetic_image(H)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
out = {}
for W in (256, 512):
    fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0, width=W), device=dev, N_rand=8192,
                        shifts=shifts, seed=0, prefetch=4, width=W)
    traj = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, 2002):
        fit.step_full()
        if it in (50, 100, 200, 500, 1000, 1500, 2001):
            torch.cuda.synchronize()
            traj.append({"iter": it, "wall_s": round(time.perf_counter() - t0, 3), "psnr_known_dB": round(fit.psnr(), 3),
                         "psnr_unknown_dB": round(fit.psnr("unknown"), 3)})
    fit.close()
    out[f"W{W}"] = traj
print(json.dumps(out, indent=1))
