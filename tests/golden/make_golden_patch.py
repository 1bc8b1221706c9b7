"""Golden vectors G5 (GridPatchSampler), G6 (contextual loss core), G7 (LPIPS head) from the
reference's own code.  Called by make_golden.py (same import shims).  The VGG trunks cannot
be built here (torchvision + pretrained weights absent): G6/G7 start from
feature tensors; G7 drives the reference's LPIPS.forward with a stand-in `net` that returns
prepared features, so everything after the trunk is the reference's code, with the vendored
lin weights (externel_lib/lpips/weights/v0.1/vgg.pth)."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def synthetic(H):
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import oracle
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, 1)
    return img, mask, shifts


def g5_sampler(R):
    """models/sampler.py GridPatchSampler.sample_patches, 24 calls, NumPy seed 0."""
    H = 256
    img, mask, shifts = synthetic(H)
    masked = img * mask
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_val = np.stack(np.nonzero(1 - mask[..., 0]), 1)
    out = {"H": np.int64(H), "shifts": np.array(shifts[0], np.float64)}
    for tag, P, nsamp in (("p64", 64, 2), ("p32", 32, 4)):
        np.random.seed(0)
        S = R["sampler"].GridPatchSampler(img=torch.Tensor(masked[None]), mask=torch.Tensor(mask[None]),
                                          N_samples=nsamp, patch_size=P, height=H, width=H,
                                          pool_train=torch.Tensor(i_train), pool_val=torch.Tensor(i_val),
                                          selected_shifts=shifts, no_reg_sampling=False)
        out[f"{tag}_pool_train_n"] = np.int64(S.pool_train.shape[0])
        out[f"{tag}_pool_val_n"] = np.int64(S.pool_val.shape[0])
        modes, ks, cents, dists, fake_sum, real_sum, rmask_sum, fmask_sum = [], [], [], [], [], [], [], []
        for it in range(24):
            r = S.sample_patches(topk=3, invalid_ratio=0.3)
            real, rmask, fake, fmask, coords, mode, k, w = r
            modes.append({"val": 0, "train": 1, "same": 2, None: -1}[mode])
            ks.append(k)
            if k == 0:
                cents.append(np.full((nsamp, 2), -1)); dists.append(np.full((nsamp, 3), -1.0))
                fake_sum.append(0.0); real_sum.append(0.0); rmask_sum.append(0.0); fmask_sum.append(0.0)
                continue
            c = coords[:, P // 2, P // 2, :].numpy()
            cents.append(c)
            # the per-patch weights (1/d normalised) identify the chosen lattice distances up to
            # a common factor; store them sorted (which tie wins is backend-defined)
            wv = w.numpy().reshape(nsamp, -1)
            d = np.full((nsamp, 3), -1.0)
            d[:, :wv.shape[1]] = np.sort(wv, 1)
            dists.append(d)
            fake_sum.append(float(fake.double().sum())); fmask_sum.append(float(fmask.double().sum()))
            real_sum.append(float(real.double().sum())); rmask_sum.append(float(rmask.double().sum()))
            if it < 3:
                out[f"{tag}_fake_{it}"] = fake[:, 0].numpy()          # (n,3,P,P)
                out[f"{tag}_fmask_{it}"] = fmask[:, 0].numpy()
                out[f"{tag}_real_{it}"] = real.numpy()                # (n,k,P,P,3)
                out[f"{tag}_rmask_{it}"] = rmask.numpy()
        out[f"{tag}_modes"] = np.array(modes); out[f"{tag}_k"] = np.array(ks)
        out[f"{tag}_centres"] = np.stack(cents); out[f"{tag}_weights_sorted"] = np.stack(dists)
        out[f"{tag}_fake_sum"] = np.array(fake_sum); out[f"{tag}_real_sum"] = np.array(real_sum)
        out[f"{tag}_rmask_sum"] = np.array(rmask_sum); out[f"{tag}_fmask_sum"] = np.array(fmask_sum)
        out[f"{tag}_rng_after"] = np.array(np.random.uniform(0, 1, 4))   # RNG consumption check
    # extract_glimpse alone (utils/extract_glimpse.py:7-79) incl. windows that leave the image
    g = torch.Generator().manual_seed(0)
    im = torch.rand(1, 3, 40, 56, generator=g)
    offs = torch.tensor([[10., 12.], [0., 0.], [55., 39.], [28., 20.], [3., 38.]])   # (x, y)
    gl = R["glimpse"].extract_glimpse(im.tile([5, 1, 1, 1]), size=(16, 16), offsets=offs, padding_mode="zeros",
                                      mode="nearest", normalized=False, centered=False)
    out["glimpse_img"] = im.numpy(); out["glimpse_offs_xy"] = offs.numpy(); out["glimpse_out"] = gl.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_sampler.npz"), **out)


def g6_cx(R):
    """contextual_loss/functional.py:9-63 on feature tensors, forward + d/dx."""
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, shape, same in (("a", (6, 256, 8, 8), False), ("b", (3, 64, 12, 12), False), ("same", (2, 32, 8, 8), True),
                             ("w", (6, 64, 8, 8), False)):
        y = torch.relu(torch.randn(*shape, generator=g))            # relu features are >= 0
        x = y.clone() if same else torch.relu(0.6 * y + 0.8 * torch.randn(*shape, generator=g))
        if same:
            x = x + 0.01 * torch.randn(*shape, generator=g)
        x.requires_grad_(True)
        w = torch.rand(shape[0], generator=g) if tag == "w" else None
        loss = R["cxf"].contextual_loss(x, y, 0.5, w)
        loss.backward()
        out[f"{tag}_x"] = x.detach().numpy(); out[f"{tag}_y"] = y.numpy()
        out[f"{tag}_loss"] = loss.detach().numpy(); out[f"{tag}_dx"] = x.grad.numpy()
        if w is not None:
            out[f"{tag}_w"] = w.numpy()
    np.savez_compressed(os.path.join(OUT, "g6_cx.npz"), **out)


def g7_lpips(R):
    """externel_lib/lpips/lpips.py:92-133 LPIPS.forward(use_robust=True, normalize=True) with
    a stand-in trunk; lin weights = the vendored v0.1/vgg.pth; robust latents perturbed."""
    import lpips as L
    import lpips.lpips as LL
    chns = [64, 128, 256, 512, 512]
    sizes = [16, 8, 4, 2, 1]
    g = torch.Generator().manual_seed(5)
    feats = {}

    class FakeNet:
        def forward(self, x):
            return feats[int(x[0, 0, 0, 0].item() * 0 + self.which)]

    obj = LL.LPIPS.__new__(LL.LPIPS)
    torch.nn.Module.__init__(obj)
    obj.pnet_type, obj.pnet_tune, obj.pnet_rand, obj.spatial, obj.lpips, obj.version = "vgg", False, False, False, True, "0.1"
    obj.scaling_layer = LL.ScalingLayer()
    obj.chns, obj.L = chns, 5
    obj.adaptive_perceps = [R["adaptive"].AdaptiveLossFunction(num_dims=c, float_dtype=np.float32, device="cpu") for c in chns]
    obj.lins = torch.nn.ModuleList([LL.NetLinLayer(c, use_dropout=True) for c in chns])
    for i, l in enumerate(obj.lins):
        setattr(obj, f"lin{i}", l)
    sd = torch.load(os.path.join(REF, "externel_lib/lpips/weights/v0.1/vgg.pth"), map_location="cpu")
    obj.load_state_dict(sd, strict=False)
    obj.eval()
    out = {}
    for kk, c in enumerate(chns):
        out[f"lin{kk}"] = obj.lins[kk].model[1].weight.detach().numpy().reshape(-1)
        with torch.no_grad():
            obj.adaptive_perceps[kk].latent_alpha.add_(0.5 * torch.randn(1, c, generator=g))
            obj.adaptive_perceps[kk].latent_scale.add_(0.5 * torch.randn(1, c, generator=g))
        out[f"la{kk}"] = obj.adaptive_perceps[kk].latent_alpha.detach().numpy()
        out[f"ls{kk}"] = obj.adaptive_perceps[kk].latent_scale.detach().numpy()
    N = 2
    f0 = [torch.relu(torch.randn(N, c, s, s, generator=g)).requires_grad_(True) for c, s in zip(chns, sizes)]
    f1 = [torch.relu(torch.randn(N, c, s, s, generator=g)) for c, s in zip(chns, sizes)]
    calls = {"n": 0}

    class Net:
        def forward(self, x):
            calls["n"] += 1
            return f0 if calls["n"] == 1 else f1
    obj.net = Net()
    in0 = torch.rand(N, 3, 16, 16, generator=g)
    in1 = torch.rand(N, 3, 16, 16, generator=g)
    val = obj.forward(in0, in1, use_robust=True, normalize=True)
    loss = torch.mean(val)                                     # train.py:249
    loss.backward()
    out["val"] = val.detach().numpy(); out["loss"] = loss.detach().numpy()
    for kk in range(5):
        out[f"f0_{kk}"] = f0[kk].detach().numpy(); out[f"f1_{kk}"] = f1[kk].numpy()
        out[f"df0_{kk}"] = f0[kk].grad.numpy()
        out[f"dla{kk}"] = obj.adaptive_perceps[kk].latent_alpha.grad.numpy()
        out[f"dls{kk}"] = obj.adaptive_perceps[kk].latent_scale.grad.numpy()
    # scaling layer (lpips.py:136-143) on the normalised input
    out["in0"] = in0.numpy()
    out["scaled0"] = obj.scaling_layer(2 * in0 - 1).numpy()
    np.savez_compressed(os.path.join(OUT, "g7_lpips.npz"), **out)


def main(R):
    g5_sampler(R)
    g6_cx(R)
    try:
        g7_lpips(R)
    except Exception as e:  # noqa
        import traceback
        traceback.print_exc()
        print("G7 skipped:", e)
