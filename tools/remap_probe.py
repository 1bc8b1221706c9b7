import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
dev = torch.device("cuda:0")
Hr, K = 1024, 3
im_r, _ = syn.synthetic_image(Hr, seed=7)
a_r, p_r, sh_r = syn.synthetic_periodicity(Hr, K)
clear = np.ones((Hr, Hr, 1), np.float32); clear[Hr // 3:Hr // 2] = 0.0
for mode, pf in (("reference", 4), ("fast", 0)):
    fr = CompletionFit(im_r, np.ones((Hr, Hr, 1), np.float32), a_r, p_r, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, seed=0,
                       shifts=sh_r, task="remapping", clear_mask=clear, prefetch=pf, contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False, rng_mode=mode)
    for _ in range(20): fr.step_full()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(60): fr.step_full()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/60
    # device-only: replay one batch
    b = None
    while b is None: b = fr.sample_batch()
    for _ in range(5): fr.step_from(b)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(40): fr.step_from(b)
    torch.cuda.synchronize(); dd=(time.perf_counter()-t)/40
    t=time.perf_counter()
    for _ in range(10): fr.draw_batch()
    dh=(time.perf_counter()-t)/10
    print(mode, f"loop {dt*1e3:.3f} ms/iter, device-only step {dd*1e3:.3f} ms ({b['source']}, P={b['P']}), host draw {dh*1e3:.3f} ms")
    fr.close()
