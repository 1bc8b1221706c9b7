"""remapping task at 1024^2 incl. host sampling: reference RNG stream vs fast mode (bench.py's remapping_task_1024sq extra)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
Hr, K = 1024, 3
im_r, _ = syn.synthetic_image(Hr, seed=7)
a_r, p_r, sh_r = syn.synthetic_periodicity(Hr, K)
clear = np.ones((Hr, Hr, 1), np.float32); clear[Hr // 3:Hr // 2] = 0.0
for mode, pf in (("reference", 4), ("fast", 0), ("reference", 4)):
    fr = CompletionFit(im_r, np.ones((Hr, Hr, 1), np.float32), a_r, p_r, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev,
                       N_rand=8192, seed=0, shifts=sh_r, task="remapping", clear_mask=clear, prefetch=pf, rng_mode=mode,
                       contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False)
    for _ in range(20): fr.step_full()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): fr.step_full()
    torch.cuda.synchronize()
    print("NPP_RNG_THREADS=" + os.environ.get("NPP_RNG_THREADS", "default"), "NPP_RNG_AVX2=" + os.environ.get("NPP_RNG_AVX2", "default"), mode, f"{(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/iter")
    fr.close()
