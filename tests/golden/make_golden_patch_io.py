#!/usr/bin/env python3
"""Golden G8p (VERDICT r5 "Missing #3" / SURVEY.md 8 row a10): the TENSORS of the reference's patch plumbing,
NPP_completion/train.py:200-236 (+ the LPIPS inputs of :241-250), for the first 'val' (use_comp), 'train' and 'same' iteration
of the reference's loop on the synthetic 256^2 image (patch size 32 to keep the fixture small; top-1 network, PyTorch CPU fp32,
NumPy seed 0, tie order defined as in make_golden_fit_tasks.py).  Per source:

  inputs   what models/sampler.py GridPatchSampler.sample_patches returned (real, rmask, fake, fmask, k, as returned) and the
           network's prediction on the patch rows (pred[N_rand:])
  forward  x_in, y_in handed to contextual_loss's trunk (:228-236), and for 'same' the two tensors handed to percepLoss (:241-247)
  backward x_in.grad (what the contextual branch sends back), in0.grad of the LPIPS branch ('same'), and pred.grad on the patch
           rows -- the sum the plumbing forms from them (autograd through :200-236)

The trunks behind the two losses are the build's fixed-seed stand-ins (tests/comparators.py TorchTrunk) -- they only supply SOME
upstream gradient: the fixture pins the plumbing between `pred` and the loss inputs, whatever the gradient is.

    python tests/golden/make_golden_patch_io.py        (~1 min)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS, _net  # noqa: E402
from make_golden_fit_patch import reference_lpips  # noqa: E402
from make_golden_fit_tasks import stable_topk  # noqa: E402
import oracle  # noqa: E402


def main(H=256, P=32, N_rand=2048, n_p=2, topk=3, max_iters=80):
    R = import_reference()
    stable_topk()
    emb, msec, cxf = R["emb"], R["msec"], R["cxf"]
    from npp_amd.losses import _VGG19
    from comparators import TorchTrunk
    percep = reference_lpips(R)
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, 1)
    masked = img * mask
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_val = np.stack(np.nonzero(1 - mask[..., 0]), 1)
    i_all = np.stack(np.meshgrid(np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 2)
    torch.manual_seed(0)
    embedder, freq_nerf = emb.get_embedder(10, 0, (H, H))
    ep, _ = emb.get_embedder(10, 0, (H, H), selected_angles=torch.Tensor(angles[0]), selected_periods=torch.Tensor(periods[0]),
                             freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS)
    torch.manual_seed(0)
    net = _net(R, 1, 256, int(freq_nerf))
    adaptive = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
    grad_vars = list(net.parameters()) + list(adaptive.parameters())
    for a_ in percep.adaptive_perceps:
        grad_vars += list(a_.parameters())
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
    vgg = TorchTrunk(_VGG19, taps=(17,))
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    with torch.no_grad():
        tab_train = embedder.embed(ep.embed(torch.Tensor(i_train)))
        tab_all = embedder.embed(ep.embed(torch.Tensor(i_all))).reshape(H, H, -1)
    masked_t, mask_t = torch.Tensor(masked), torch.Tensor(mask)
    np.random.seed(0)
    S = R["sampler"].GridPatchSampler(img=masked_t[None], mask=mask_t[None], N_samples=n_p, patch_size=P, height=H, width=H,
                                      pool_train=torch.Tensor(i_train), pool_val=torch.Tensor(i_val), selected_shifts=shifts,
                                      no_reg_sampling=False)
    out, seen, global_step = {}, set(), 0
    for i in range(1, max_iters + 1):
        real, rmask, fake, fmask, coords, source, k, weight = S.sample_patches(topk=topk, invalid_ratio=0.3)      # train.py:152-157
        if k == 0:
            continue
        coords = coords.reshape(-1, 2)
        emb_patch = tab_all[coords[:, 0], coords[:, 1], :]
        sel = np.random.choice(i_train.shape[0], size=[N_rand], replace=False)
        c = i_train[sel]
        gt = masked_t[c[:, 0], c[:, 1], :]
        pred = torch.sigmoid(net(None, torch.cat([tab_train[sel], emb_patch])))
        pred.retain_grad()
        opt.zero_grad()
        loss = msec.img2mse(pred[:N_rand], gt, "robust_loss_adaptive", adaptive, torch.ones_like(gt[:, :1]))
        # ---- train.py:200-236, line by line
        pp = pred[N_rand:].reshape(n_p, 1, P, P, 3).permute(0, 1, 4, 2, 3).tile((1, k, 1, 1, 1))                  # :201-203
        real_p = real.reshape(-1, k, 3).reshape(n_p, k, P, P, 3).permute(0, 1, 4, 2, 3)                            # :206-208
        rm = rmask.permute(0, 1, 4, 2, 3).reshape(-1, 1, P, P)                                                     # :213-214
        pp, real_p = pp.reshape(-1, 3, P, P), real_p.reshape(-1, 3, P, P)
        fk, fm = fake.reshape(-1, 3, P, P), fmask.reshape(-1, 1, P, P)
        x_in = (fk * fm + pp * (1 - fm)) * rm if source == "val" else pp * rm                                     # :228-236 (use_comp)
        y_in = real_p * rm
        x_in.retain_grad()
        fx = vgg((x_in - mean) / std)[0]
        with torch.no_grad():
            fy = vgg((y_in - mean) / std)[0]
        patch_loss = cxf.contextual_loss(fx, fy, 0.5, None) * 0.001
        lp0 = lp1 = None
        if source == "same":                                                                                       # :241-250
            lp0, lp1 = pp * rm, fk * rm
            lp0.retain_grad()
            patch_loss = patch_loss + torch.mean(percep(lp0, lp1, use_robust=True, normalize=True)) * 0.001
        (loss + patch_loss).backward()
        if source not in seen:
            seen.add(source)
            t = source
            out[f"{t}_iter"], out[f"{t}_k"] = np.int64(i), np.int64(k)
            for name, v in (("real", real), ("rmask", rmask), ("fake", fake), ("fmask", fmask)):
                out[f"{t}_{name}"] = v.detach().numpy().astype(np.float32)
            out[f"{t}_pred_rows"] = pred[N_rand:].detach().numpy()
            out[f"{t}_x_in"], out[f"{t}_y_in"] = x_in.detach().numpy(), y_in.detach().numpy()
            out[f"{t}_dx_in"] = x_in.grad.numpy()
            out[f"{t}_dpred_rows"] = pred.grad[N_rand:].numpy().copy()
            if lp0 is not None:
                out[f"{t}_lp0"], out[f"{t}_lp1"], out[f"{t}_dlp0"] = lp0.detach().numpy(), lp1.detach().numpy(), lp0.grad.numpy()
            print(i, source, k, {n_: tuple(out[f"{t}_{n_}"].shape) for n_ in ("real", "rmask", "fake", "fmask", "x_in")}, flush=True)
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))
        for g in opt.param_groups:
            g["lr"] = new_lr
        global_step += 1
        if len(seen) == 3:
            break
    assert seen == {"val", "train", "same"}, seen
    np.savez_compressed(os.path.join(OUT, "g8p_patch_io.npz"), P=np.int64(P), n_p=np.int64(n_p), N_rand=np.int64(N_rand), **out)
    print("wrote g8p_patch_io.npz", os.path.getsize(os.path.join(OUT, "g8p_patch_io.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
