#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE's own
Python modules from /root/reference (read-only) in the build container.

Only runs where /root/reference exists; the committed .npz files are what the tests
use everywhere else (the GPU box never sees the reference).  The import recipe is
SURVEY.md Appendix B: three stub modules for packages the image lacks
(torchvision is only imported, never called, by models/activations.py:3; torch_dct
only by robust_loss_pytorch/util.py:25), and the vendored libs put on sys.path under
their own top-level names.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ("torchvision", "torchvision.models", "torch_dct"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    for p in (REF, os.path.join(REF, "externel_lib")):
        if p not in sys.path:
            sys.path.insert(0, p)
    pkg = types.ModuleType("contextual_loss")
    pkg.__path__ = [os.path.join(REF, "externel_lib", "contextual_loss")]
    sys.modules.setdefault("contextual_loss", pkg)
    torch.Tensor.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    import models.embedder as emb
    import models.networks as nets
    import models.activations as acts
    import models.mse_calculator as msec
    import models.sampler as sampler
    import utils.extract_glimpse as glimpse
    import robust_loss_pytorch
    import robust_loss_pytorch.adaptive as adaptive
    import contextual_loss.functional as cxf
    os.chdir(cwd)
    torch.autograd.set_detect_anomaly(False)
    return dict(emb=emb, nets=nets, acts=acts, msec=msec, sampler=sampler,
                glimpse=glimpse, adaptive=adaptive, cxf=cxf)


FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS = [1], [0, -1, 1, 0.5, -0.5], [0]


def g1_embed(R):
    """models/embedder.py: Embedder_periodic.embed then Embedder.embed, K in {1,3}."""
    out = {}
    torch.manual_seed(0)
    embedder, out_dim = R["emb"].get_embedder(10, 0, (256, 256))
    freqs = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    out["freqs"] = freqs
    out["out_dim"] = np.int64(out_dim)
    rng = np.random.RandomState(1)
    for tag, res in (("sq", (256, 256)), ("rect", (211, 325))):
        H, W = res
        coords = np.stack([rng.randint(0, H, 64), rng.randint(0, W, 64)], 1)
        coords = np.concatenate([coords, [[0, 0], [0, W - 1], [H - 1, 0], [H - 1, W - 1]]], 0)
        angles = np.array([[80.54, 168.69], [80.54, 168.69], [33.3, 121.0]], np.float32)
        periods = np.array([[40.77, 36.48], [81.54, 72.96], [17.25, 23.5]], np.float32)
        embedder.res = res
        v_all, e_all = [], []
        for k in range(3):
            ep, d22 = R["emb"].get_embedder(10, 0, res, selected_angles=torch.Tensor(angles[k]),
                                            selected_periods=torch.Tensor(periods[k]),
                                            freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS,
                                            angle_offsets=ANGLE_OFFSETS)
            assert d22 == 22
            v = ep.embed(torch.Tensor(coords.astype(np.float32)).clone())
            e = embedder.embed(v)
            v_all.append(v.numpy())
            e_all.append(e.numpy())
        out[f"{tag}_res"] = np.array(res, np.int64)
        out[f"{tag}_coords"] = coords.astype(np.int32)
        out[f"{tag}_angles"] = angles
        out[f"{tag}_periods"] = periods
        out[f"{tag}_warp"] = np.stack(v_all, 0)            # (3, N, 22)
        out[f"{tag}_emb"] = np.concatenate(e_all, 1)       # (N, 3*462)
    np.savez_compressed(os.path.join(OUT, "g1_embed.npz"), **out)
    return freqs


def _net(R, K, W, freq_nerf):
    if K > 1:
        return R["nets"].NPP_Net(input_ch_periodic=22, input_ch_periodic_aux=22 * (K - 1),
                                 freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS,
                                 angle_offsets=ANGLE_OFFSETS, D=8, W=W, freq_nerf=freq_nerf,
                                 output_ch=3, skips=[4], activation="snake")
    return R["nets"].NPP_Net_top1(input_ch_periodic=22, freq_scales=FREQ_SCALES,
                                  freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, D=8, W=W,
                                  freq_nerf=freq_nerf, output_ch=3, skips=[4], activation="snake")


def g2_mlp(R):
    """models/networks.py NPP_Net / NPP_Net_top1 forward + autograd parameter grads."""
    out = {}
    for tag, K, W, fn, B in (("small_k3", 3, 32, 5, 32), ("small_k1", 1, 32, 5, 32),
                             ("full_k3", 3, 256, 21, 16), ("full_k1", 1, 256, 21, 16)):
        torch.manual_seed(0)
        net = _net(R, K, W, fn)
        E = 22 * fn
        g = torch.Generator().manual_seed(7)
        emb = torch.rand(B, K * E, generator=g) * 2 - 1
        draw = torch.randn(B, 3, generator=g)
        raw = net(None, emb)
        (raw * draw).sum().backward()
        out[f"{tag}_emb"] = emb.numpy()
        out[f"{tag}_draw"] = draw.numpy()
        out[f"{tag}_raw"] = raw.detach().numpy()
        out[f"{tag}_pred"] = torch.sigmoid(raw).detach().numpy()
        for name, p in net.named_parameters():
            if p.grad is None:
                continue
            if tag.startswith("small"):
                out[f"{tag}_P_{name}"] = p.detach().numpy()
                out[f"{tag}_G_{name}"] = p.grad.numpy()
            else:
                # full size: weights are re-created in the tests with torch.manual_seed(0)
                # and the reference's nn.Linear construction order (tests/refinit.py);
                # store only checksums, norms and a corner of every gradient.
                pd, gd = p.detach().double(), p.grad.double()
                out[f"{tag}_Psum_{name}"] = np.array([pd.sum().item(), pd.abs().sum().item()])
                out[f"{tag}_Gnorm_{name}"] = np.array([gd.norm().item(), gd.sum().item()])
                g2 = p.grad.reshape(p.shape[0], -1)
                out[f"{tag}_Gcorner_{name}"] = g2[:8, :8].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g2_mlp.npz"), **out)


def g3_snake(R):
    z = torch.linspace(-12, 12, 257, requires_grad=True)
    a = R["acts"].SnakeActivation()(z)
    a.sum().backward()
    np.savez_compressed(os.path.join(OUT, "g3_snake.npz"), z=z.detach().numpy(),
                        a=a.detach().numpy(), da=z.grad.numpy())


def g4_robust(R):
    """models/mse_calculator.img2mse + AdaptiveLossFunction (num_dims=3, cpu)."""
    out = {}
    msec = R["msec"]
    g = torch.Generator().manual_seed(3)
    for tag, la, ls in (("init", None, None), ("pert", [0.7, -1.3, 2.2], [-0.8, 0.4, 1.5])):
        ad = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
        if la is not None:
            with torch.no_grad():
                ad.latent_alpha.copy_(torch.tensor([la]))
                ad.latent_scale.copy_(torch.tensor([ls]))
        pred = torch.rand(257, 3, generator=g)
        pred.requires_grad_(True)
        gt = torch.rand(257, 3, generator=g)
        mask = (torch.rand(257, 1, generator=g) > 0.3).float()
        for mtag, m in (("nomask", None), ("mask", mask)):
            for p in (pred, ad.latent_alpha, ad.latent_scale):
                p.grad = None
            loss = msec.img2mse(pred, gt, "robust_loss_adaptive", ad, m)
            loss.backward()
            k = f"{tag}_{mtag}"
            out[f"{k}_loss"] = loss.detach().numpy()
            out[f"{k}_dpred"] = pred.grad.numpy().copy()
            out[f"{k}_dla"] = ad.latent_alpha.grad.numpy().copy()
            out[f"{k}_dls"] = ad.latent_scale.grad.numpy().copy()
        out[f"{tag}_pred"] = pred.detach().numpy()
        out[f"{tag}_gt"] = gt.numpy()
        out[f"{tag}_mask"] = mask.numpy()
        out[f"{tag}_latent_alpha"] = ad.latent_alpha.detach().numpy()
        out[f"{tag}_latent_scale"] = ad.latent_scale.detach().numpy()
        out[f"{tag}_alpha"] = ad.alpha().detach().numpy()
        out[f"{tag}_scale"] = ad.scale().detach().numpy()
        # element-wise nll on a residual grid, for the spline / power-law pieces
        x = torch.linspace(-3, 3, 61)[:, None].repeat(1, 3)
        out[f"{tag}_grid_x"] = x.numpy()
        out[f"{tag}_grid_nll"] = ad.lossfun(x).detach().numpy()
    # log-partition spot values over the reachable alpha range (distribution.py:143-169)
    ad = R["adaptive"].AdaptiveLossFunction(3, np.float32, "cpu")
    alphas = torch.linspace(0.001, 1.999, 400)
    out["logz_alpha"] = alphas.numpy()
    out["logz"] = ad.distribution.log_base_partition_function(alphas).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_robust.npz"), **out)


def g9_adam(R):
    """torch.optim.Adam(lr=5e-4, betas=(0.9,0.999)) + the train.py:253-263 LR rule."""
    torch.manual_seed(0)
    p = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    opt = torch.optim.Adam(p, lr=5e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(5)
    out = {"p0_init": p[0].detach().numpy().copy(), "p1_init": p[1].detach().numpy().copy()}
    global_step = 0
    lrs = []
    for it in range(5):
        grads = [torch.randn(7, 5, generator=g) * (10.0 ** (it - 2)), torch.randn(11, generator=g)]
        for q, gr in zip(p, grads):
            q.grad = gr.clone()
        out[f"g0_{it}"] = grads[0].numpy()
        out[f"g1_{it}"] = grads[1].numpy()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        new_lr = 5e-4 * (0.1 ** (global_step / (500 * 100)))
        for pg in opt.param_groups:
            pg["lr"] = new_lr
        global_step += 1
        out[f"p0_{it}"] = p[0].detach().numpy().copy()
        out[f"p1_{it}"] = p[1].detach().numpy().copy()
    out["lrs_used"] = np.array(lrs, np.float64)
    np.savez_compressed(os.path.join(OUT, "g9_adam.npz"), **out)


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here; golden vectors are already committed")
        return 0
    R = import_reference()
    g1_embed(R)
    g2_mlp(R)
    g3_snake(R)
    g4_robust(R)
    g9_adam(R)
    try:
        import make_golden_patch as mgp  # G5-G8 live in a second file
        mgp.main(R)
    except ImportError:
        pass
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
    return 0


if __name__ == "__main__":
    sys.path.insert(0, OUT)
    sys.exit(main())
