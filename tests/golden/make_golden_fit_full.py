#!/usr/bin/env python3
"""Golden G8c2_full: BASELINE config c2's COMPLETE fit from the reference's own modules, for the reference's full iteration
count -- `for i in trange(1, N_iters = 2001)` (NPP_completion/train.py:133, options/arg_config.py:96), i.e. iterations 1 .. 2000,
INCLUDING the patch-size decay that fires at i = 2000 (train.py:137-141: patch 96 -> 48, patch_num 2 -> 4, sampler tables
rebuilt, pools reset) -- on the synthetic 512^2 image with top-3 proposals, W = 256, PyTorch CPU fp32, NumPy seed 0.

The loop body is make_golden_fit_patch.py's (GridPatchSampler, table gathers, NPP_Net, img2mse(robust_loss_adaptive), the patch
plumbing of train.py:200-236, contextual_loss on VGG19[0:18]-shaped features, the reference's LPIPS.forward on 'same' iterations,
Adam + LR rule), with torch.topk's tie order DEFINED as in make_golden_fit_tasks.py (stable: one of the orders the reference may
produce, and the one the build's sampler implements), so the patch losses of EVERY source are comparable value by value.

Stored: the FINAL RENDERED IMAGE (uint8, round(255 * pred)) -- the output, not a prefix; PSNR (known, unknown) vs the ground
truth at 100 / 250 / 500 / 1000 / 1500 / 2000; the (source, k) sequence and (weighted patch loss, raw CX) of every iteration;
the LPIPS values of the 'same' iterations; the adaptive latents at the end.  Absent offline: pretrained VGG
weights -- the trunks are the build's fixed-seed stand-ins (losses._Trunk, seeds 1234 / 4321) behind the reference's forward code.

    python tests/golden/make_golden_fit_full.py [--threads N] [--iters N]        (~1.5-2.5 s per iteration: about an hour)
A partial file (g8c2_full.partial.npz) is rewritten every 250 iterations so a killed run leaves its progress.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))                 # tests/: comparators.py
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS, _net  # noqa: E402
from make_golden_fit_patch import reference_lpips  # noqa: E402
from make_golden_fit_tasks import stable_topk  # noqa: E402
import oracle  # noqa: E402


def main(n_last=2000, out_name="g8c2_full.npz", H=512, K=3, decay=2000, checkpoints=(100, 250, 500, 1000, 1500, 2000)):
    R = import_reference()
    stable_topk()
    emb, msec, cxf = R["emb"], R["msec"], R["cxf"]
    from npp_amd.losses import _VGG19
    from comparators import TorchTrunk as _Trunk
    percep = reference_lpips(R)
    N_rand, topk = 8192, 3
    patch_num = 2                                                                # arg_config.py:63
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    patch_size = int(oracle.patch_size_from_period(periods[0]))
    P0 = patch_size
    masked = img * mask
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_val = np.stack(np.nonzero(1 - mask[..., 0]), 1)
    i_all = np.stack(np.meshgrid(np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 2)
    torch.manual_seed(0)
    embedder, freq_nerf = emb.get_embedder(10, 0, (H, H))
    freqs = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    eps = [emb.get_embedder(10, 0, (H, H), selected_angles=torch.Tensor(angles[k_]), selected_periods=torch.Tensor(periods[k_]),
                            freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS)[0] for k_ in range(K)]
    torch.manual_seed(0)
    net = _net(R, K, 256, int(freq_nerf))
    adaptive = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
    grad_vars = list(net.parameters()) + list(adaptive.parameters())
    for a_ in percep.adaptive_perceps:                                            # helpers.py:147-151
        grad_vars += list(a_.parameters())
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
    vgg = _Trunk(_VGG19, taps=(17,))
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(3, 1, 1)                  # contextual.py:41-46
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    with torch.no_grad():
        tab_train = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_train))) for ep in eps], 1)            # train.py:93-105
        tab_all = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_all))) for ep in eps], 1).reshape(H, H, -1)
    masked_t, img_t, mask_t = torch.Tensor(masked), torch.Tensor(img), torch.Tensor(mask)
    np.random.seed(0)
    S = R["sampler"].GridPatchSampler(img=masked_t[None], mask=mask_t[None], N_samples=patch_num, patch_size=patch_size, height=H,
                                      width=H, pool_train=torch.Tensor(i_train), pool_val=torch.Tensor(i_val),
                                      selected_shifts=shifts, no_reg_sampling=False)

    def render():
        with torch.no_grad():
            flat = tab_all.reshape(H * H, -1)
            return torch.cat([torch.sigmoid(net(None, flat[j:j + 20000])) for j in range(0, H * H, 20000)]).reshape(H, H, 3)

    def psnr(pred):
        return [float(-10 * torch.log10((((pred - img_t) ** 2) * m).sum() / (m.sum() * 3))) for m in (mask_t, 1 - mask_t)]

    traj, seq, lp_vals, patch_vals, sizes, global_step, t0 = [], [], [], [], [], 0, time.time()

    def save(name, final):
        extra = {"patch_loss": np.array(patch_vals, np.float64), "lpips_values": np.array(lp_vals, np.float64),
                 "latent_alpha": adaptive.latent_alpha.detach().numpy(), "latent_scale": adaptive.latent_scale.detach().numpy()}
        for kk, a_ in enumerate(percep.adaptive_perceps):
            extra[f"la{kk}"] = a_.latent_alpha.detach().numpy()
            extra[f"ls{kk}"] = a_.latent_scale.detach().numpy()
            extra[f"lin{kk}"] = percep.lins[kk].model[1].weight.detach().numpy().reshape(-1)
        if final is not None:
            extra["final_image_u8"] = np.round(255.0 * np.clip(final.numpy(), 0, 1)).astype(np.uint8)
        np.savez_compressed(os.path.join(OUT, name), traj=np.array(traj, np.float64), seq=np.array(seq, np.int64), freqs=freqs,
                            sizes=np.array(sizes, np.int64), H=np.int64(H), N_rand=np.int64(N_rand), global_step=np.int64(global_step),
                            K=np.int64(K), P=np.int64(P0), n_iters=np.int64(len(seq)), decay=np.int64(decay), **extra)

    for i in range(1, n_last + 1):
        if i % decay == 0 and i != 1 and patch_size > 31:                        # train.py:137-141
            patch_size, patch_num = patch_size // 2, patch_num * 2
            S.reset_patchsize(img=masked_t[None], mask=mask_t[None], N_samples=patch_num, patch_size=patch_size)
            S.reset_pool(torch.Tensor(i_train), torch.Tensor(i_val))
        n_p, P = patch_num, patch_size
        real, rmask, fake, fmask, coords, source, k, weight = S.sample_patches(topk=topk, invalid_ratio=0.3)   # train.py:152-157
        seq.append(({"val": 0, "train": 1, "same": 2, None: -1}[source], k))
        sizes.append((P, n_p))
        if k == 0:
            continue                                                              # :160-161
        coords = coords.reshape(-1, 2)
        emb_patch = tab_all[coords[:, 0], coords[:, 1], :]                        # :166-167
        sel = np.random.choice(i_train.shape[0], size=[N_rand], replace=False)    # :172
        c = i_train[sel]
        gt = masked_t[c[:, 0], c[:, 1], :]
        pred = torch.sigmoid(net(None, torch.cat([tab_train[sel], emb_patch])))  # :181,189
        opt.zero_grad()
        loss = msec.img2mse(pred[:N_rand], gt, "robust_loss_adaptive", adaptive, torch.ones_like(gt[:, :1]))   # :195
        pp = pred[N_rand:].reshape(n_p, 1, P, P, 3).permute(0, 1, 4, 2, 3).tile((1, k, 1, 1, 1))                 # :201-203
        real_p = real.reshape(-1, k, 3).reshape(n_p, k, P, P, 3).permute(0, 1, 4, 2, 3)                           # :206-208
        rm = rmask.permute(0, 1, 4, 2, 3).reshape(-1, 1, P, P)                                                    # :213-214
        pp, real_p = pp.reshape(-1, 3, P, P), real_p.reshape(-1, 3, P, P)
        fk, fm = fake.reshape(-1, 3, P, P), fmask.reshape(-1, 1, P, P)
        x_in = (fk * fm + pp * (1 - fm)) * rm if source == "val" else pp * rm                                    # :228-236 (use_comp)
        y_in = real_p * rm
        fx = vgg((x_in - mean) / std)[0]                                                                          # contextual.py:56-64
        with torch.no_grad():
            fy = vgg((y_in - mean) / std)[0]
        cx = cxf.contextual_loss(fx, fy, 0.5, None)
        patch_loss = cx * 0.001                                                                                   # :238-239
        if source == "same":                                                                                      # :241-251
            perc = torch.mean(percep(pp * rm, fk * rm, use_robust=True, normalize=True))
            lp_vals.append([i, float(perc)])
            patch_loss = patch_loss + perc * 0.001
        patch_vals.append([i, float(patch_loss), float(cx)])
        loss = loss + patch_loss
        loss.backward()
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))                                                      # :253-263
        for g in opt.param_groups:
            g["lr"] = new_lr
        global_step += 1
        if i in checkpoints:
            traj.append([i] + psnr(render()))
            print(traj[-1], seq[-1], (P, n_p), f"{time.time() - t0:.0f}s", flush=True)
        elif i % 50 == 0:
            print(i, seq[-1], f"{time.time() - t0:.0f}s", flush=True)
        if i % 250 == 0 and i != n_last:
            save(out_name.replace(".npz", ".partial.npz"), None)
    final = render()
    if n_last not in checkpoints:
        traj.append([n_last] + psnr(final))
    save(out_name, final)
    part = os.path.join(OUT, out_name.replace(".npz", ".partial.npz"))
    if os.path.exists(part):
        os.remove(part)
    print("wrote", out_name, "final PSNR (known, unknown)", psnr(final), f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    a = sys.argv[1:]
    if "--threads" in a:
        torch.set_num_threads(int(a[a.index("--threads") + 1]))
    if "--iters" in a:                                   # a shortened rehearsal (other file name)
        n = int(a[a.index("--iters") + 1])
        main(n_last=n, out_name=f"g8c2_full_{n}.npz", decay=n, checkpoints=(n // 2, n))
    else:
        main()
