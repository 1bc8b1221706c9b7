"""Patch losses of the loop: ContextualLoss (externel_lib/contextual_loss/modules/contextual.py:
9-68) and LPIPS(net='vgg') (externel_lib/lpips/lpips.py:27-133), same call signatures.

From the feature tensors onward everything runs in libnpp_hip.so (npp_cx_fwd_bwd,
npp_lpips_layer); they are wired into torch.autograd with two small Functions so that the
modules remain drop-ins for the reference's (`loss.backward()` keeps working).

The VGG19[0:18] / VGG16 trunks are frozen third-party convolution stacks whose pretrained
torchvision weights are not available offline: they are built here layer by
layer (no torchvision import), run through PyTorch/MIOpen as glue, and take either a
torchvision-format state_dict supplied by the user or a fixed-seed random init.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

_VGG19 = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256]                       # features[0:18] -> relu3_4
_VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def _make_features(cfg):
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
            cin = v
    return nn.Sequential(*layers)


class _Trunk(nn.Module):
    def __init__(self, cfg, taps, state_dict=None, seed=1234):
        super().__init__()
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.features = _make_features(cfg)
        torch.random.set_rng_state(g)
        if state_dict is not None:      # torchvision naming: features.<idx>.weight / .bias
            self.load_state_dict({k: v for k, v in state_dict.items() if k.startswith("features.")}, strict=False)
        self.taps = taps
        for p in self.parameters():
            p.requires_grad = False     # vgg.py:26-28, pretrained_networks.py:116-118

    def forward(self, x):
        x = x.contiguous()          # strided views make MIOpen fall back to its naive "nonpacked" kernels
        outs = []
        for i, m in enumerate(self.features):
            x = m(x)
            if i in self.taps:
                outs.append(x)
        return outs


class _CXFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fx, fy, band_width, weight):
        loss, dfx = ops.cx_fwd_bwd(fx.contiguous(), fy.contiguous(), band_width, weight, 1.0, None, fx.requires_grad)
        ctx.save_for_backward(dfx if dfx is not None else torch.empty(0))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dfx,) = ctx.saved_tensors
        return dfx * g, None, None, None


def contextual_loss(x, y, band_width=0.5, weight=None, loss_type="cosine"):
    """contextual_loss/functional.py:9-63 on feature tensors."""
    assert x.size() == y.size(), "input tensor must have the same size."
    assert loss_type == "cosine", "only the cosine distance is on the built path"
    return _CXFunction.apply(x, y, float(band_width), weight)


class ContextualLoss(nn.Module):
    def __init__(self, band_width=0.5, loss_type="cosine", use_vgg=False, vgg_layer="relu3_4", vgg_state_dict=None):
        super().__init__()
        assert band_width > 0, "band_width parameter must be positive."
        assert loss_type == "cosine" and vgg_layer == "relu3_4"
        self.band_width = band_width
        if use_vgg:
            self.vgg_model = _Trunk(_VGG19, taps=(17,), state_dict=vgg_state_dict)
            self.register_buffer("vgg_mean", torch.tensor([[[0.485]], [[0.456]], [[0.406]]]))
            self.register_buffer("vgg_std", torch.tensor([[[0.229]], [[0.224]], [[0.225]]]))

    def forward(self, x, y, weight=None):
        if hasattr(self, "vgg_model"):
            assert x.shape[1] == 3 and y.shape[1] == 3, "VGG model takes 3 chennel images."
            x = x.sub(self.vgg_mean).div(self.vgg_std)
            y = y.sub(self.vgg_mean).div(self.vgg_std)
            x = self.vgg_model(x)[0]
            with torch.no_grad():
                y = self.vgg_model(y)[0]
        return contextual_loss(x, y, self.band_width, weight)


class _LPIPSLayerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f0, f1, owner, kk):
        loss = torch.zeros(1, dtype=torch.float32, device=f0.device)
        need = f0.requires_grad
        df0 = torch.empty_like(f0) if need else None
        dlat = torch.zeros_like(owner.latents[kk]) if need else None
        ops.lpips_layer(f0.contiguous(), f1.contiguous(), owner.lins[kk], owner.latents[kk], owner.spline, owner.n_knots,
                        owner.x_scale, 1.0, loss, df0, dlat)
        ctx.owner, ctx.kk = owner, kk
        ctx.save_for_backward(df0 if need else torch.empty(0), dlat if need else torch.empty(0))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        df0, dlat = ctx.saved_tensors
        ctx.owner.dlatents[ctx.kk].add_(dlat * g)           # the reference's list of AdaptiveLossFunction params
        ctx.owner.touched = True
        return df0 * g, None, None, None


class LPIPS(nn.Module):
    """LPIPS(net='vgg') with the reference's adaptive-robust head.  forward(in0, in1, use_robust,
    normalize) returns the batch MEAN as a scalar (the caller's torch.mean, train.py:249, is folded in)."""
    chns = [64, 128, 256, 512, 512]

    def __init__(self, net="vgg", lin_weights=None, vgg_state_dict=None, device="cuda"):
        super().__init__()
        assert net in ("vgg", "vgg16")
        self.net = _Trunk(_VGG16, taps=(3, 8, 15, 22, 29), state_dict=vgg_state_dict, seed=4321)
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        dev = torch.device(device)
        if lin_weights is None:          # weights/v0.1/vgg.pth is not redistributed here: fixed-seed non-negative stand-ins
            rng = np.random.RandomState(7)
            lin_weights = [np.abs(rng.randn(c)).astype(np.float32) * 0.05 for c in self.chns]
        self.lins = [torch.as_tensor(np.asarray(w, np.float32).reshape(-1)).to(dev) for w in lin_weights]
        # AdaptiveLossFunction(num_dims=chn) per tap (lpips.py:57-61): [latent_alpha(C) | latent_scale(C)]
        self.latents = [torch.cat([torch.full((c,), 2.3841858e-07), torch.zeros(c)]).to(dev) for c in self.chns]
        self.dlatents = [torch.zeros_like(l) for l in self.latents]
        self.lat_m = [torch.zeros_like(l) for l in self.latents]
        self.lat_v = [torch.zeros_like(l) for l in self.latents]
        self.lat_step = 0
        self.touched = False
        self.spline, self.n_knots, self.x_scale = ops.load_spline(dev)
        self.to(dev)

    def forward(self, in0, in1, use_robust=True, retPerLayer=False, normalize=False):
        assert use_robust and not retPerLayer, "only the use_robust=True path of the loop is built"
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        in0, in1 = (in0 - self.shift) / self.scale, (in1 - self.shift) / self.scale
        outs0 = self.net(in0)
        with torch.no_grad():
            outs1 = self.net(in1)
        val = 0
        for kk in range(5):
            val = val + _LPIPSLayerFunction.apply(outs0[kk], outs1[kk], self, kk)
        return val

    def zero_latent_grads(self):
        for d in self.dlatents:
            d.zero_()
        self.touched = False

    def adam_step(self, lr):
        """Adam over the robust latents; only called when they received a gradient this iteration
        (torch skips parameters whose grad is None: their step count does not advance)."""
        self.lat_step += 1
        for kk in range(5):
            ops.adam_step(self.latents[kk], self.lat_m[kk], self.lat_v[kk], self.dlatents[kk], 1, self.latents[kk].numel(),
                          lr, self.lat_step)
