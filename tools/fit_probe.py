import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
img, mask = oracle.synthetic_image(H)
angles, periods, shifts = oracle.synthetic_periodicity(H, K)
P = oracle.init_params(K, seed=0)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, P, N_rand=8192)
print("init psnr", fit.psnr())
t0 = time.time()
for it in range(1, 301):
    fit.step()
    if it in (1, 10, 25, 50, 100, 200, 300):
        torch.cuda.synchronize()
        print(it, "psnr known %.2f unknown %.2f loss %.4f  t=%.2fs" % (fit.psnr(), fit.psnr("unknown"), fit.net.loss_buf.item(), time.time() - t0), flush=True)
