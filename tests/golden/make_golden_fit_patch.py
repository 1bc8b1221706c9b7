#!/usr/bin/env python3
"""Golden G8b: the reference's COMPLETE loop body -- models.sampler.GridPatchSampler, the table gathers,
NPP_Net_top1, img2mse(robust_loss_adaptive), the patch plumbing of NPP_completion/train.py:200-236 and
contextual_loss.functional.contextual_loss on VGG19[0:18]-shaped features, Adam + LR rule -- driven on the synthetic 256^2
image (PyTorch CPU fp32, NumPy seed 0).  Differences to a stock run, both forced by what is absent offline: the trunk is a
VGG19[0:18]-shaped torch stack with a fixed-seed init (the build's losses._Trunk: torchvision's pretrained weights are not
available) and the LPIPS term of 'same' iterations is left out (its trunk needs them too).
Stores PSNR checkpoints + the patch-source / k sequence.      python tests/golden/make_golden_fit_patch.py   (~2 min)
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
import oracle  # noqa: E402


def reference_lpips(R):
    """The reference's LPIPS(net='vgg') object (externel_lib/lpips/lpips.py:27-133) without its constructor (which needs
    torchvision + pretrained VGG16): every attribute the forward uses is set by hand, the lin layers carry the vendored
    weights/v0.1/vgg.pth, the per-layer AdaptiveLossFunction objects are the reference's, and the trunk is the build's
    fixed-seed VGG16-shaped stack (losses._Trunk, seed 4321) behind the `net.forward(x) -> 5 taps` interface."""
    import lpips.lpips as LL
    from npp_amd.losses import _VGG16
    from comparators import TorchTrunk as _Trunk          # tests/comparators.py: losses._Trunk + the torch forward
    chns = [64, 128, 256, 512, 512]
    obj = LL.LPIPS.__new__(LL.LPIPS)
    torch.nn.Module.__init__(obj)
    obj.pnet_type, obj.pnet_tune, obj.pnet_rand, obj.spatial, obj.lpips, obj.version = "vgg", False, False, False, True, "0.1"
    obj.scaling_layer = LL.ScalingLayer()
    obj.chns, obj.L = chns, 5
    obj.adaptive_perceps = [R["adaptive"].AdaptiveLossFunction(num_dims=c, float_dtype=np.float32, device="cpu") for c in chns]
    obj.lins = torch.nn.ModuleList([LL.NetLinLayer(c, use_dropout=True) for c in chns])
    for i, l in enumerate(obj.lins):
        setattr(obj, f"lin{i}", l)
    obj.load_state_dict(torch.load("/root/reference/externel_lib/lpips/weights/v0.1/vgg.pth", map_location="cpu"), strict=False)
    obj.eval()
    trunk = _Trunk(_VGG16, taps=(3, 8, 15, 22, 29), seed=4321)

    class Net:
        def forward(self, x):
            return trunk(x)
    obj.net = Net()
    return obj


def main(with_lpips=False, n_iters=100, out_name="g8b_fit_patch.npz", checkpoints=(10, 25, 50, 75, 100), H=256, K=1):
    """H = 512, K = 3 (--c2): BASELINE config c2's COMPLETE iteration at its real size -- NPP_Net with top-3 proposals, 8192 pixel rows
    + 2 patches of 96^2 against 3 real patches each, contextual loss every iteration, LPIPS on 'same' ones (g8c2_loop.npz)."""
    R = import_reference()
    emb, msec, cxf = R["emb"], R["msec"], R["cxf"]
    from npp_amd.losses import _VGG19
    from comparators import TorchTrunk as _Trunk                                   # the VGG19[0:18]-shaped stand-in, seed 1234
    percep = reference_lpips(R) if with_lpips else None
    N_rand, n_p, topk = 8192, 2, 3
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    P = 64 if H == 256 else int(oracle.patch_size_from_period(periods[0]))
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
    if percep is not None:                                                        # helpers.py:147-151
        for a_ in percep.adaptive_perceps:
            grad_vars += list(a_.parameters())
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
    vgg = _Trunk(_VGG19, taps=(17,))
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(3, 1, 1)                  # contextual.py:41-46
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    with torch.no_grad():
        tab_train = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_train))) for ep in eps], 1)            # train.py:93-105
        tab_all = torch.cat([embedder.embed(ep.embed(torch.Tensor(i_all))) for ep in eps], 1).reshape(H, H, -1)   # :103-105 i_all table
    masked_t, img_t, mask_t = torch.Tensor(masked), torch.Tensor(img), torch.Tensor(mask)
    np.random.seed(0)
    S = R["sampler"].GridPatchSampler(img=masked_t[None], mask=mask_t[None], N_samples=n_p, patch_size=P, height=H, width=H,
                                      pool_train=torch.Tensor(i_train), pool_val=torch.Tensor(i_val), selected_shifts=shifts,
                                      no_reg_sampling=False)

    def psnr():
        with torch.no_grad():
            flat = tab_all.reshape(H * H, -1)
            pred = torch.cat([torch.sigmoid(net(None, flat[j:j + 20000])) for j in range(0, H * H, 20000)]).reshape(H, H, 3)
        return [float(-10 * torch.log10((((pred - img_t) ** 2) * m).sum() / (m.sum() * 3))) for m in (mask_t, 1 - mask_t)]
    traj, seq, lp_vals, patch_vals, global_step, t0 = [], [], [], [], 0, time.time()
    for i in range(1, n_iters + 1):
        real, rmask, fake, fmask, coords, source, k, weight = S.sample_patches(topk=topk, invalid_ratio=0.3)   # train.py:152-157
        seq.append(({"val": 0, "train": 1, "same": 2, None: -1}[source], k))
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
        patch_loss = cx * 0.001                                                                                   # :238-239, weight 1e-3
        if percep is not None and source == "same":                                                               # :241-251
            perc = torch.mean(percep(pp * rm, fk * rm, use_robust=True, normalize=True))
            lp_vals.append([i, float(perc)])
            patch_loss = patch_loss + perc * 0.001
        patch_vals.append([i, float(patch_loss), float(cx)])
        loss = loss + patch_loss
        loss.backward()
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))
        for g in opt.param_groups:
            g["lr"] = new_lr
        global_step += 1
        if i in checkpoints:
            traj.append([i] + psnr())
            print(traj[-1], seq[-1], f"{time.time() - t0:.0f}s", flush=True)
    extra = {"patch_loss": np.array(patch_vals, np.float64)}        # (iteration, weighted patch loss of the iteration, raw CX value)
    if percep is not None:
        extra["lpips_values"] = np.array(lp_vals, np.float64)                     # (iteration, mean LPIPS-robust value) of 'same' iterations
        for kk, a_ in enumerate(percep.adaptive_perceps):
            extra[f"la{kk}"] = a_.latent_alpha.detach().numpy()
            extra[f"ls{kk}"] = a_.latent_scale.detach().numpy()
            extra[f"lin{kk}"] = percep.lins[kk].model[1].weight.detach().numpy().reshape(-1)
    np.savez_compressed(os.path.join(OUT, out_name), traj=np.array(traj, np.float64), seq=np.array(seq, np.int64), freqs=freqs,
                        H=np.int64(H), N_rand=np.int64(N_rand), global_step=np.int64(global_step), K=np.int64(K), P=np.int64(P), **extra)


if __name__ == "__main__":
    if "--c2" in sys.argv:                       # g8c2_loop.npz: config c2's complete iteration at 512^2, K = 3 (~1.5 s per iteration)
        main(with_lpips=True, n_iters=60, out_name="g8c2_loop.npz", checkpoints=(10, 20, 40, 60), H=512, K=3)
    elif "--lpips" in sys.argv:
        main(with_lpips=True, n_iters=100, out_name="g8c_fit_lpips.npz")
    else:
        main()
