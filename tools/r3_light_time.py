"""candidate fit time, graph vs eager"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn
from npp_amd.light import ProposalRanker
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
H = 512
img, mask = syn.synthetic_image(H, seed=0)
angles, periods, shifts = syn.synthetic_periodicity(H, 3)
pseudo = np.ones((H, H), np.float32); pseudo[H // 4:H // 4 + 128, H // 4:H // 4 + 160] = 0
rk = ProposalRanker(img * mask, np.stack(np.nonzero(pseudo * mask[..., 0]), 1), np.stack(np.nonzero((1 - pseudo) * mask[..., 0]), 1), device=dev, rng_mode="fast")
for g in (True, False, True, False):
    rk.fit_candidate(angles[0], periods[0], use_graph=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net = rk.fit_candidate(angles[0], periods[0], use_graph=g)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("graph" if g else "eager", f"candidate fit {t:.4f} s = {t / rk.N_iters * 1e3:.4f} ms/iter  score", rk.score(net)[0])
