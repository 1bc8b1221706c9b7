"""Round-4 probe: the 9-candidate proposal-ranking fit (ProposalRanker.fit_candidates, 300 iterations x 2048 rows) in fp32 and bf16,
wall time per iteration of the set; run under rocprofv3 --kernel-trace --stats for the per-kernel durations.
    python tools/r4_light16_probe.py [fp32|bf16 ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn                 # noqa: E402
from npp_amd.light import ProposalRanker             # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
img, mask = syn.synthetic_image(H, seed=0)
pseudo = np.ones((H, H), np.float32)
pseudo[H // 4:H // 4 + 128, H // 4:H // 4 + 160] = 0
cands9 = [(angles[i % K] + 3.0 * (i // K), periods[i % K] * (1.0 + 0.11 * (i // K))) for i in range(9)]
for prec in (sys.argv[1:] or ["fp32", "bf16"]):
    rk = ProposalRanker(img * mask, np.stack(np.nonzero(pseudo * mask[..., 0]), 1), np.stack(np.nonzero((1 - pseudo) * mask[..., 0]), 1),
                        device=dev, rng_mode="fast", precision=prec)
    rk.fit_candidates(cands9)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rk.fit_candidates(cands9)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{prec}: set fit {min(ts):.4f} s = {min(ts) / rk.N_iters * 1e3:.4f} ms per iteration of the set ({[round(t, 4) for t in ts]})", flush=True)
