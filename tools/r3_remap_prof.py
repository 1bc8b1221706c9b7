"""Profile target: 40 iterations of the remapping task at 1024^2 (fast sampler) -- rocprofv3 --kernel-trace -- python tools/r3_remap_prof.py"""
import os, sys
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
fr = CompletionFit(im_r, np.ones((Hr, Hr, 1), np.float32), a_r, p_r, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev,
                   N_rand=8192, seed=0, shifts=sh_r, task="remapping", clear_mask=clear, prefetch=0, rng_mode="fast",
                   contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False)
for _ in range(int(os.environ.get("R3_ITERS", "40"))): fr.step_full()
torch.cuda.synchronize()
fr.close()
