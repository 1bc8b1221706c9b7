#!/usr/bin/env python3
"""Goldens G8r / G8s / G8d (VERDICT r2 "Missing #2", "Weak #1"): the reference's loop bodies of the two task VARIANTS and of
the completion loop with LPIPS, driven from the reference's own modules (PyTorch CPU fp32, NumPy seed 0) on the synthetic 256^2
image, like make_golden_fit_patch.py does for G8b / G8c.

  g8r  NPP_remapping/train.py:158-300 -- the whole valid image is trained on, the CLEAR region is the 'val' pool and the sampler
       mask, the pixel loss weighs blurry pixels 0.3 through gt_mask = clear_mask (:203; models/mse_calculator.py:17), and
       models/style_loss.py VGG16FeatureExtractor.style_loss (adaptive form, 64^2 + 128^2 + 256^2 latent pairs in the same Adam,
       models/helpers.py:153-159) joins contextual_loss on comp / pred patches (:253-273; style_weight 1, contextual_weight 0.01).
  g8s  NPP_segmentation/train.py:148-290 -- the initial PERIODIC region is the known mask, the image trained and sampled on is
       the (blurred) input, contextual weight 0.005, no LPIPS, and the learning rate never decays (`global_step += 1` sits
       outside the loop, :408).
  g8d  = G8c (completion + the reference's LPIPS.forward) regenerated with a DEFINED tie order (below).

Tie order.  sampler.py:203 picks the k nearest lattice candidates with torch.topk(distance, largest=False); the distances are
small integers with many ties and which of the tied candidates win is backend-defined.  These goldens run the
reference with torch.topk replaced, for that call shape only, by its STABLE realisation (first candidates in the reference's
own enumeration order win) -- one of the orders the reference may produce, and the one the build's sampler implements -- so
that the patch losses of 'val' / 'train' iterations are comparable value by value instead of within a tie-induced 35 %.

Absent offline: pretrained VGG weights.  The trunks are the build's fixed-seed stand-ins of identical shape
(losses._Trunk: VGG19[0:18] seed 1234, VGG16 seed 4321, VGG16[:17] seed 777) behind the reference's own forward code.

    python tests/golden/make_golden_fit_tasks.py [--remap] [--remap1024] [--seg] [--lpips]        (~3 min each)
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
import oracle  # noqa: E402


def stable_topk():
    """torch.topk -> stable for 1-D smallest-k queries (the sampler's call)."""
    orig = torch.topk

    def topk(x, k, dim=-1, largest=True, sorted=True):
        if x.dim() == 1 and not largest:
            idx = torch.argsort(x, stable=True)[:k]
            return x[idx], idx
        return orig(x, k, dim=dim, largest=largest, sorted=sorted)
    torch.topk = topk
    return orig


def reference_style(R):
    """models/style_loss.py VGG16FeatureExtractor (use_adaptive=True) without its constructor (torchvision + pretrained
    weights): enc_1..3 are slices [:5], [5:10], [10:17] of the build's fixed-seed VGG16[:17]-shaped stack; forward() and
    style_loss() are the reference's code, the adaptives its AdaptiveLossFunction(num_dims = chn ** 2)."""
    import models.style_loss as SL
    from npp_amd.losses import _VGG16_STYLE
    from comparators import TorchTrunk as _Trunk
    trunk = _Trunk(_VGG16_STYLE, taps=(4, 9, 16), seed=777)
    obj = SL.VGG16FeatureExtractor.__new__(SL.VGG16FeatureExtractor)
    torch.nn.Module.__init__(obj)
    f = trunk.features
    obj.enc_1, obj.enc_2, obj.enc_3 = torch.nn.Sequential(*f[:5]), torch.nn.Sequential(*f[5:10]), torch.nn.Sequential(*f[10:17])
    obj.use_adaptive = True
    obj.adaptives = [R["adaptive"].AdaptiveLossFunction(num_dims=c ** 2, float_dtype=np.float32, device="cpu") for c in (64, 128, 256)]
    return obj


def task_inputs(task, H):
    """Synthetic inputs of a task: (image trained on, sampler / known mask, pixel-loss mask or None, i_train, i_val)."""
    img, cmask = oracle.synthetic_image(H)
    if task == "completion":
        mask = cmask
        return img * mask, img, mask, None, np.stack(np.nonzero(mask[..., 0]), 1), np.stack(np.nonzero(1 - mask[..., 0]), 1)
    if task == "remapping":
        clear = np.ones((H, H, 1), np.float32)
        clear[H // 3:H // 2] = 0.0                       # a blurry band (the reference finds it with blur_detection.py)
        i_train = np.stack(np.nonzero(np.ones((H, H))), 1)                           # loaders.py:279: every valid pixel
        i_val = np.stack(np.nonzero(clear[..., 0]), 1)                                # :280: the clear region
        return img, img, clear, clear, i_train, i_val
    # segmentation: the periodic region = everything but a disc; the image trained on is the input as given
    yy, xx = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
    disc = ((yy - 0.6 * H) ** 2 + (xx - 0.4 * H) ** 2) < (H / 8) ** 2
    period = (1.0 - disc.astype(np.float32))[..., None]
    return img, img, period, None, np.stack(np.nonzero(period[..., 0]), 1), np.stack(np.nonzero(1 - period[..., 0]), 1)


def main(task, with_lpips, out_name, n_iters=100, checkpoints=(10, 25, 50, 75, 100), H=256, P=64, K=1):
    """H = 1024, P = 160, K = 3 (--remap1024, g8r1024): BASELINE config c4's grid -- the remapping loop at its real size (1 048 576
    pixel rows in the 'train' pool, 2 patches of 160^2 against 3 real ones each, style Gram matrices of 6 x 160^2 patches)."""
    R = import_reference()
    stable_topk()
    emb, msec, cxf = R["emb"], R["msec"], R["cxf"]
    from npp_amd.losses import _VGG19
    from comparators import TorchTrunk as _Trunk
    percep = reference_lpips(R) if with_lpips else None
    style = reference_style(R) if task == "remapping" else None
    cx_w = {"completion": 1e-3, "remapping": 0.01, "segmentation": 0.005}[task]     # arg_config.py:90,281,196
    N_rand, n_p, topk = 8192, 2, 3
    train_img, clean, mask, pix_mask, i_train, i_val = task_inputs(task, H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    i_all = np.stack(np.meshgrid(np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 2)
    torch.manual_seed(0)
    embedder, freq_nerf = emb.get_embedder(10, 0, (H, H))
    freqs = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    eps = [emb.get_embedder(10, 0, (H, H), selected_angles=torch.Tensor(angles[k_]), selected_periods=torch.Tensor(periods[k_]),
                            freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS)[0] for k_ in range(K)]
    torch.manual_seed(0)
    net = _net(R, K, 256, int(freq_nerf))
    adaptive = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
    grad_vars = list(net.parameters()) + list(adaptive.parameters())                  # helpers.py:144
    if percep is not None:                                                            # :147-151
        for a_ in percep.adaptive_perceps:
            grad_vars += list(a_.parameters())
    if style is not None:                                                             # :153-159
        for a_ in style.adaptives:
            grad_vars += list(a_.parameters())
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
    vgg = _Trunk(_VGG19, taps=(17,))
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(3, 1, 1)                      # contextual.py:41-46
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    with torch.no_grad():
        tab_all = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_all))) for ep in eps], 1).reshape(H, H, -1)
        tab_train = tab_all[i_train[:, 0], i_train[:, 1], :]                         # (the same rows of the same table: train.py gathers them)
    train_t, clean_t, mask_t = torch.Tensor(train_img), torch.Tensor(clean), torch.Tensor(mask)
    pm_t = None if pix_mask is None else torch.Tensor(pix_mask)
    np.random.seed(0)
    S = R["sampler"].GridPatchSampler(img=train_t[None], mask=mask_t[None], N_samples=n_p, patch_size=P, height=H, width=H,
                                      pool_train=torch.Tensor(i_train), pool_val=torch.Tensor(i_val), selected_shifts=shifts,
                                      no_reg_sampling=False)

    def psnr():
        with torch.no_grad():
            flat = tab_all.reshape(H * H, -1)
            pred = torch.cat([torch.sigmoid(net(None, flat[j:j + 20000])) for j in range(0, H * H, 20000)]).reshape(H, H, 3)
        return [float(-10 * torch.log10((((pred - clean_t) ** 2) * m).sum() / (m.sum() * 3))) for m in (mask_t, 1 - mask_t)]
    traj, seq, lp_vals, patch_vals, global_step, t0 = [], [], [], [], 0, time.time()
    for i in range(1, n_iters + 1):
        real, rmask, fake, fmask, coords, source, k, weight = S.sample_patches(topk=topk, invalid_ratio=0.3)
        seq.append(({"val": 0, "train": 1, "same": 2, None: -1}[source], k))
        if k == 0:
            continue
        coords = coords.reshape(-1, 2)
        emb_patch = tab_all[coords[:, 0], coords[:, 1], :]
        sel = np.random.choice(i_train.shape[0], size=[N_rand], replace=False)
        c = i_train[sel]
        gt = train_t[c[:, 0], c[:, 1], :]
        gt_mask = torch.ones_like(gt[:, :1]) if pm_t is None else pm_t[c[:, 0], c[:, 1], :]         # remapping train.py:203
        pred = torch.sigmoid(net(None, torch.cat([tab_train[sel], emb_patch])))
        opt.zero_grad()
        loss = msec.img2mse(pred[:N_rand], gt, "robust_loss_adaptive", adaptive, gt_mask)
        pix_val = float(loss)
        pp = pred[N_rand:].reshape(n_p, 1, P, P, 3).permute(0, 1, 4, 2, 3).tile((1, k, 1, 1, 1))
        real_p = real.reshape(-1, k, 3).reshape(n_p, k, P, P, 3).permute(0, 1, 4, 2, 3)
        rm = rmask.permute(0, 1, 4, 2, 3).reshape(-1, 1, P, P)
        pp, real_p = pp.reshape(-1, 3, P, P), real_p.reshape(-1, 3, P, P)
        fk, fm = fake.reshape(-1, 3, P, P), fmask.reshape(-1, 1, P, P)
        x_in = (fk * fm + pp * (1 - fm)) * rm if source == "val" else pp * rm
        y_in = real_p * rm
        st_val = 0.0
        patch_loss = 0.0
        if style is not None:                                                                         # remapping train.py:253-261
            st = style.style_loss(x_in, y_in, None)
            st_val = float(st)
            patch_loss = patch_loss + st * 1.0
        fx = vgg((x_in - mean) / std)[0]
        with torch.no_grad():
            fy = vgg((y_in - mean) / std)[0]
        cx = cxf.contextual_loss(fx, fy, 0.5, None)
        patch_loss = patch_loss + cx * cx_w
        if percep is not None and source == "same":
            perc = torch.mean(percep(pp * rm, fk * rm, use_robust=True, normalize=True))
            lp_vals.append([i, float(perc)])
            patch_loss = patch_loss + perc * 0.001
        patch_vals.append([i, float(patch_loss), float(cx), st_val, pix_val])
        loss = loss + patch_loss
        loss.backward()
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))
        for g in opt.param_groups:
            g["lr"] = new_lr
        if task != "segmentation":                      # NPP_segmentation/train.py:408: the increment is outside the loop there
            global_step += 1
        if i in checkpoints:
            traj.append([i] + psnr())
            print(task, traj[-1], seq[-1], f"{time.time() - t0:.0f}s", flush=True)
    extra = {"patch_loss": np.array(patch_vals, np.float64)}   # (iteration, weighted patch loss, raw CX, raw style, pixel loss)
    extra["latent_alpha"] = adaptive.latent_alpha.detach().numpy()
    extra["latent_scale"] = adaptive.latent_scale.detach().numpy()
    if percep is not None:
        extra["lpips_values"] = np.array(lp_vals, np.float64)
        for kk, a_ in enumerate(percep.adaptive_perceps):
            extra[f"la{kk}"] = a_.latent_alpha.detach().numpy()
            extra[f"ls{kk}"] = a_.latent_scale.detach().numpy()
            extra[f"lin{kk}"] = percep.lins[kk].model[1].weight.detach().numpy().reshape(-1)
    if style is not None:
        # level 0 in full (64^2 pairs); of the two big levels (128^2, 256^2) a fixed sample of 4096 pairs each
        rs = np.random.RandomState(5)
        for kk, a_ in enumerate(style.adaptives):
            la, ls = a_.latent_alpha.detach().numpy().reshape(-1), a_.latent_scale.detach().numpy().reshape(-1)
            idx = np.arange(la.size) if la.size <= 4096 else np.sort(rs.choice(la.size, 4096, replace=False))
            extra[f"sidx{kk}"], extra[f"sla{kk}"], extra[f"sls{kk}"] = idx.astype(np.int64), la[idx], ls[idx]
    np.savez_compressed(os.path.join(OUT, out_name), traj=np.array(traj, np.float64), seq=np.array(seq, np.int64), freqs=freqs,
                        H=np.int64(H), N_rand=np.int64(N_rand), global_step=np.int64(global_step), K=np.int64(K), P=np.int64(P), **extra)


if __name__ == "__main__":
    if "--remap" in sys.argv:
        main("remapping", False, "g8r_fit_remap.npz")
    if "--remap1024" in sys.argv:                        # g8r1024: config c4's grid, 16 iterations (~10 s each)
        main("remapping", False, "g8r1024_fit_remap.npz", n_iters=16, checkpoints=(1, 4, 8, 12, 16), H=1024, P=160, K=3)
    if "--seg" in sys.argv:
        main("segmentation", False, "g8s_fit_segment.npz")
    if "--lpips" in sys.argv:
        main("completion", True, "g8d_fit_lpips_stable.npz")
