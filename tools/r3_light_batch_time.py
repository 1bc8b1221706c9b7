"""Round-3 probe: the candidates of one image fitted one by one, on side streams, and stacked into every launch
(NPPNetLightBatch): wall time of the 9-candidate set (search.py's default) and per candidate."""
import os
import sys
import time

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
                    rng_mode="fast")
nc = int(os.environ.get("R3_NCAND", "9"))
cands = [(angles[i % 3] + 3.0 * (i // 3), periods[i % 3] * (1.0 + 0.11 * (i // 3))) for i in range(nc)]
rk._pixel_draws()


def timed(fn, reps=2):
    out = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nets = fn()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return min(out[1:]), nets


t_one, net1 = timed(lambda: [rk.fit_candidate(*cands[0])])
print(f"one candidate alone          {t_one:.4f} s = {t_one / rk.N_iters * 1e3:.4f} ms/iter")
t_st, nets_s = timed(lambda: rk.fit_candidates(cands, batched=False))
print(f"{nc} candidates, side streams   {t_st:.4f} s = {t_st / nc:.4f} s per candidate")
t_b, nets_b = timed(lambda: rk.fit_candidates(cands, batched=True))
print(f"{nc} candidates, stacked        {t_b:.4f} s = {t_b / nc:.4f} s per candidate, {t_b / rk.N_iters * 1e3:.4f} ms per iteration of the set")
for a, b in zip(nets_s, nets_b):
    pa, pb = a.params.cpu().numpy(), b.params.cpu().numpy()
    print("  rel-L2 params stacked vs streams", float(np.linalg.norm(pa - pb) / np.linalg.norm(pa)), " score", rk.score(a)[0], rk.score(b)[0])
