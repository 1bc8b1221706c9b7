"""Comparator code for the tests and the golden generators -- NOT part of the product (VERDICT r5 "What's weak" #12): the torch
forward of a trunk and the reference-style (torch.autograd) form of one loop iteration.  The product package runs neither."""
import torch

from npp_amd.losses import _Trunk


class TorchTrunk(_Trunk):
    """losses._Trunk (the layers + weights HipTrunk packs) with the plain torch fp32 forward over them: list of tap outputs, like
    the reference's `vgg.features` walks (externel_lib/contextual_loss/modules/vgg.py:30-36, lpips/pretrained_networks.py:119-134)."""

    def forward(self, x):
        x = x.contiguous()          # strided views make MIOpen fall back to its naive "nonpacked" kernels
        outs = []
        for i, m in enumerate(self.features):
            x = m(x)
            if i in self.taps:
                outs.append(x)
        return outs


def step_from_autograd(fit, b):
    """The same iteration written like the reference's loop body, through the torch.autograd wrappers of the loss
    modules (train.py:200-251 line by line).  The comparator of CompletionFit.step_from() (fit: a CompletionFit)."""
    fit.last_source = source = b["source"]
    net, P, n_p, k, n_pix, n, bp = fit.net, b["P"], b["n_p"], b["k"], b["n_pix"], b["n"], b["bp"]
    ws = net.workspace(bp)
    net.zero_grad()
    fit.percepLoss.zero_latent_grads()
    pred = net.forward_train(b["coords"])
    ws["dpred"][n:].zero_()
    ws["n_rows"] = n
    net.pixel_loss(bp, n_pix, b["gt"], mask=b.get("pmask"), weight=fit.pix_w)
    pp_leaf = pred[n_pix:n].detach().clone().requires_grad_(True)
    pp = pp_leaf.reshape(n_p, 1, P, P, 3).permute(0, 1, 4, 2, 3).tile((1, k, 1, 1, 1)).reshape(-1, 3, P, P)
    raw = b["raw"]                                    # contiguous crops: real (n_p k,3,P,P), rmask (n_p k,1,P,P), fake / fmask (n_p,..)
    real_p, rm = raw["real"].reshape(-1, 3, P, P), raw["rmask"].reshape(-1, 1, P, P)
    fk = raw["fake"][:, None].tile([1, k, 1, 1, 1]).reshape(-1, 3, P, P)             # train.py:219-226 tiling
    fm = raw["fmask"][:, None].tile([1, k, 1, 1, 1]).reshape(-1, 1, P, P)
    x_in = (fk * fm + pp * (1 - fm)) * rm if (fit.use_comp and source == "val") else pp * rm
    nk_ = n_p * k
    weight = b.get("weight")
    loss_patch = fit.contextualLoss(x_in, real_p * rm, weight) * fit.cx_w if fit.use_contextual_loss else pp_leaf.sum() * 0.0
    if source == "same" and fit.use_perceptual_loss:
        lp = fit.percepLoss(pp * rm, fk * rm, use_robust=fit.lp_robust, normalize=True)       # mean over the nk samples
        loss_patch = loss_patch + lp * (nk_ if weight is not None else 1) * fit.lp_w
    loss_patch.backward()
    ws["dpred"][n_pix:n].copy_(pp_leaf.grad)
    fit.last_patch_loss = loss_patch.detach().reshape(1)
    lr_used = net.lr
    net.backward(bp)
    net.optimizer_step(bp)
    if fit.percepLoss.touched:
        fit.percepLoss.adam_step(lr_used)
