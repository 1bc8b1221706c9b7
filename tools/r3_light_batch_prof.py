"""Profile target: one stacked 9-candidate fit (60 iterations) -- rocprofv3 --kernel-trace --stats -- python tools/r3_light_batch_prof.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn               # noqa: E402
from npp_amd.light import ProposalRanker           # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H = 512
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, 3)
pseudo = np.ones((H, H), np.float32)
pseudo[H // 4:H // 4 + 128, H // 4:H // 4 + 160] = 0
rk = ProposalRanker(img * mask, np.stack(np.nonzero(pseudo * mask[..., 0]), 1), np.stack(np.nonzero((1 - pseudo) * mask[..., 0]), 1), device=dev,
                    rng_mode="fast", N_iters=int(os.environ.get("R3_ITERS", "60")))
cands = [(angles[i % 3] + 3.0 * (i // 3), periods[i % 3] * (1.0 + 0.11 * (i // 3))) for i in range(9)]
rk.fit_candidates(cands, batched=True)
torch.cuda.synchronize()
