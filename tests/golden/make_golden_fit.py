#!/usr/bin/env python3
"""Golden G8: the REFERENCE's own optimisation trajectory -- its modules driven exactly like the loop body of
NPP_completion/train.py:164-263 (pixel-loss part: np.random.choice of N_rand known pixels, table gather, render = sigmoid(
NPP_Net_top1(None, emb)), zero_grad, img2mse('robust_loss_adaptive', adaptive_pix), backward, Adam over net + adaptive
latents, LR rule set after the step, global_step += 1) on the synthetic 256^2 lattice image, PyTorch CPU fp32, anomaly
detection off.  Stores the PSNR over known / unknown pixels at a few checkpoints plus what is needed to start from the same
point (freqs; the weights are torch.manual_seed(0) default init in the reference's construction order = tests/refinit.py).
Takes ~1 min on 8 cores.      python tests/golden/make_golden_fit.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from make_golden import import_reference, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS, _net  # noqa: E402
import oracle  # noqa: E402  (only for the synthetic image / periodicity definition)


def main(K=1, n_iters=300, out_name="g8_fit.npz", checkpoints=(1, 5, 10, 20, 30, 40, 50, 75, 100, 150, 200, 300), W=256, H=256):
    """K = 1: NPP_Net_top1 (g8_fit.npz).  K = 3: NPP_Net with the coarse-level proposals of the synthetic lattice, BASELINE
    config c2's network (g8k3_fit.npz; models/networks.py:56-95, table = cat of the K proposals' embeddings, train.py:103-105).
    H = 512 with K = 3, W = 256: BASELINE config c2 at its REAL size (g8c2_fit.npz; the tables are 1.45 GB each)."""
    R = import_reference()
    emb, msec = R["emb"], R["msec"]
    N_rand = 8192
    img, mask = oracle.synthetic_image(H)
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    masked = img * mask
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_all = np.stack(np.meshgrid(np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 2)
    torch.manual_seed(0)
    embedder, freq_nerf = emb.get_embedder(10, 0, (H, H))                       # draws freqs from the seed-0 generator
    freqs = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    eps = [emb.get_embedder(10, 0, (H, H), selected_angles=torch.Tensor(angles[k]), selected_periods=torch.Tensor(periods[k]),
                            freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS)[0] for k in range(K)]
    torch.manual_seed(0)                                                         # weights: seed-0 default init (tests/refinit.py)
    net = _net(R, K, W, int(freq_nerf))                                         # W = 512: the reference's default --netwidth
    adaptive = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
    opt = torch.optim.Adam(list(net.parameters()) + list(adaptive.parameters()), lr=5e-4, betas=(0.9, 0.999))
    with torch.no_grad():
        tab_train = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_train))) for ep in eps], 1)   # train.py:93-105 tables
        tab_all = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_all))) for ep in eps], 1)
    masked_t, img_t, mask_t = torch.Tensor(masked), torch.Tensor(img), torch.Tensor(mask)

    def psnr():
        with torch.no_grad():
            pred = torch.cat([torch.sigmoid(net(None, tab_all[j:j + 20000])) for j in range(0, tab_all.shape[0], 20000)]).reshape(H, H, 3)
        out = []
        for m in (mask_t, 1 - mask_t):
            mse = (((pred - img_t) ** 2) * m).sum() / (m.sum() * 3)
            out.append(float(-10 * torch.log10(mse)))
        return out
    np.random.seed(0)
    traj, global_step, t0 = [], 0, time.time()
    for i in range(1, n_iters + 1):
        sel = np.random.choice(i_train.shape[0], size=[N_rand], replace=False)   # train.py:172
        c = i_train[sel]
        gt = masked_t[c[:, 0], c[:, 1], :]
        pred = torch.sigmoid(net(None, tab_train[sel]))                          # render, helpers.py:41-62
        opt.zero_grad()
        loss = msec.img2mse(pred, gt, "robust_loss_adaptive", adaptive, torch.ones_like(gt[:, :1]))   # train.py:176,195
        loss.backward()
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))                     # train.py:256-262
        for g in opt.param_groups:
            g["lr"] = new_lr
        global_step += 1
        if i in checkpoints:
            traj.append([i] + psnr() + [float(loss)])
            print(traj[-1], f"{time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, out_name), K=np.int64(K), W=np.int64(W), traj=np.array(traj, np.float64), freqs=freqs, H=np.int64(H), N_rand=np.int64(N_rand),
                        latent_alpha=adaptive.latent_alpha.detach().numpy(), latent_scale=adaptive.latent_scale.detach().numpy())


if __name__ == "__main__":
    if "--c2" in sys.argv:                        # g8c2_fit.npz: config c2 at full size (512^2, K = 3, W = 256), ~0.3 s per iteration
        main(K=3, n_iters=60, out_name="g8c2_fit.npz", checkpoints=(1, 5, 10, 20, 30, 40, 50, 60), H=512)
    elif "--c5" in sys.argv:                      # g8c5_fit.npz: config c5's network at its real size (512^2, top-5 proposals, W = 256)
        main(K=5, n_iters=60, out_name="g8c5_fit.npz", checkpoints=(1, 5, 10, 20, 30, 40, 50, 60), H=512)
    elif "--s1024" in sys.argv:                   # g8s1024_fit.npz: a 1024^2 image (configs c2 / c4's grid size), K = 3, W = 256
        main(K=3, n_iters=40, out_name="g8s1024_fit.npz", checkpoints=(1, 5, 10, 20, 30, 40), H=1024)
    elif "--w512" in sys.argv:                      # g8w512_fit.npz: NPP_Net K = 3 at the reference's default width
        main(K=3, n_iters=100, out_name="g8w512_fit.npz", checkpoints=(1, 5, 10, 20, 30, 50, 75, 100), W=512)
    elif "--k5" in sys.argv:                      # g8k5_fit.npz: BASELINE config c5's network (top-5 proposals)
        main(K=5, n_iters=100, out_name="g8k5_fit.npz", checkpoints=(1, 5, 10, 20, 30, 50, 75, 100))
    elif "--k3" in sys.argv:
        main(K=3, n_iters=150, out_name="g8k3_fit.npz", checkpoints=(1, 5, 10, 20, 30, 50, 75, 100, 150))
    else:
        main()
