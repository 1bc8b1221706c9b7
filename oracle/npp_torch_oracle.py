"""PyTorch-CPU fp32 restatement of the MLP half of the optimisation step -- the "reference on the host CPU cores" figure
SURVEY.md 8(d) specifies for the bench (`torch.set_num_threads(all)`, whole batch at once, autograd like the reference).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and by bench.py's cpu_baseline leg, never by the
product.  Pinned against the NumPy oracle (which is pinned against the reference's own modules through the golden
vectors): tests/test_oracle_torch.py compares forward values and every parameter gradient.

Restates: models/embedder.py:102-148 + :11-56 (embed), models/networks.py:56-95 / :145-173 (NPP_Net / NPP_Net_top1 with
snake, models/activations.py:29-35), models/helpers.py:55-56 (sigmoid), torch.optim.Adam as set up in helpers.py:144-164.
The robust pixel loss on the (B, 3) output is evaluated by the NumPy oracle (`img2mse_grads`): a few dozen elementwise
operations on 3 columns, no measurable share of the step."""
import math

import numpy as np
import torch

from . import npp_oracle as O


def embed_t(coords_yx, angles_deg, periods, freqs, res, offsets=O.FREQ_OFFSETS):
    """(N,2) int (row=y, col=x) -> (N, K*462) fp32: warp (embedder.py:110-133) then Fourier features (:41-44,:56)."""
    y = coords_yx[:, 0:1].to(torch.float32)
    x = coords_yx[:, 1:2].to(torch.float32)
    a = torch.as_tensor(np.asarray(angles_deg, np.float32).reshape(-1, 2))
    p = torch.as_tensor(np.asarray(periods, np.float32).reshape(-1, 2))
    f = torch.as_tensor(np.asarray(freqs, np.float32))
    outs = []
    for k in range(a.shape[0]):
        cols = []
        for ori in range(2):
            cols.append(((x / res[1] - 0.5) * 2) if ori == 0 else ((y / res[0] - 0.5) * 2))
            th = torch.deg2rad(a[k, ori])
            t = y * torch.cos(th) + x * torch.sin(th)
            for o in offsets:
                per = p[k, ori] + o
                ph = torch.remainder(t, per) / per * 2 * math.pi
                cols += [torch.sin(ph), torch.cos(ph)]
        v = torch.cat(cols, 1)                                       # (N, 22)
        blocks = [v]
        for fj in f:
            blocks += [torch.sin(v * fj), torch.cos(v * fj)]
        outs.append(torch.cat(blocks, 1))                            # (N, 462)
    return torch.cat(outs, 1)


def snake_t(z):
    return z + torch.sin(z) ** 2


def mlp_forward_t(P, emb, K, E=O.E_PER_PROPOSAL, D=8, skips=(4,)):
    """networks.py:56-95 (K>1) / :145-173 (K==1) on a dict of torch parameters in the reference's state_dict names."""
    lin = torch.nn.functional.linear
    x0, aux = emb[:, :E], emb[:, E:]
    h = x0
    for i in range(D):
        h = snake_t(lin(h, P[f"periodic_linears.{i}.weight"], P[f"periodic_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([x0, h], 1)
    f1 = lin(h, P["feature_linear1.weight"], P["feature_linear1.bias"])
    if K > 1:
        a_s = snake_t(lin(torch.cat([f1, aux], 1), P["scale_linears.0.weight"], P["scale_linears.0.bias"]))
        f2 = lin(a_s, P["feature_linear2.weight"], P["feature_linear2.bias"])
        p_in = torch.cat([f1, f2], 1)
    else:
        p_in = f1
    a_p = snake_t(lin(p_in, P["pos_linears.0.weight"], P["pos_linears.0.bias"]))
    return lin(a_p, P["rgb_linear.weight"], P["rgb_linear.bias"])


def params_t(P_np):
    return {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in P_np.items()}


def train_step_t(P, opt, coords, gt, angles, periods, freqs, res, K, latent_alpha, latent_scale):
    """One MLP-half step: embed -> forward -> sigmoid -> adaptive robust pixel loss -> backward -> Adam."""
    emb = embed_t(coords, angles, periods, freqs, res)
    pred = torch.sigmoid(mlp_forward_t(P, emb, K))
    loss, dpred, _, _ = O.img2mse_grads(pred.detach().numpy(), gt, latent_alpha, latent_scale)
    opt.zero_grad(set_to_none=True)
    pred.backward(torch.from_numpy(np.ascontiguousarray(dpred, dtype=np.float32)))
    opt.step()
    return float(loss)


# ----------------------------------------------------------------------------------------------------------------------
# Patch-loss half on PyTorch-CPU (the reference runs these as torch modules with autograd, F.conv2d on the host's MKL-DNN):
# contextual_loss/modules/vgg.py:16-36 + lpips/pretrained_networks.py:96-134 (trunks), contextual_loss/functional.py:9-63,
# 127-163 (core).  Pinned against the NumPy oracle (trunk_forward / trunk_backward / cx_backward) by tests/test_oracle_torch.py.
# ----------------------------------------------------------------------------------------------------------------------
def trunk_weights_t(weights):
    """[(w (Co,Ci,3,3), b (Co,))] NumPy -> frozen torch tensors (vgg.py:26-28: requires_grad False)."""
    return [(torch.from_numpy(np.ascontiguousarray(w, np.float32)), torch.from_numpy(np.ascontiguousarray(b, np.float32)))
            for w, b in weights]


def trunk_forward_t(x, cfg, weights_t, taps):
    """x (N,3,H,W) torch, already normalised -> list of tap outputs (indices of torchvision's `features`, like the NumPy
    oracle's trunk_forward); differentiable w.r.t. x."""
    F = torch.nn.functional
    outs, idx, wi = [], 0, 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, 2)
            idx += 1
        else:
            w, b = weights_t[wi]
            wi += 1
            x = F.relu(F.conv2d(x, w, b, padding=1))
            idx += 2
            if idx - 1 in taps:
                outs.append(x)
    return outs


def cx_loss_t(x, y, band_width=0.5, weight=None):
    """functional.py:9-63 with loss_type 'cosine' (:139-163), relative distance (:133-136) and compute_cx (:127-130)."""
    F = torch.nn.functional
    N, C = x.shape[:2]
    mu = y.mean(dim=(0, 2, 3), keepdim=True)
    xn = F.normalize(x - mu, p=2, dim=1).reshape(N, C, -1)
    yn = F.normalize(y - mu, p=2, dim=1).reshape(N, C, -1)
    dist = 1 - torch.clamp(torch.bmm(xn.transpose(1, 2), yn), min=0, max=1)
    dmin, _ = torch.min(dist, dim=2, keepdim=True)
    w = torch.exp((1 - dist / (dmin + 1e-5)) / band_width)
    cx = w / torch.sum(w, dim=2, keepdim=True)
    cx = torch.mean(torch.max(cx, dim=1)[0], dim=1)
    if weight is not None:
        return torch.sum(-torch.log(cx * weight + 1e-5))
    return torch.mean(-torch.log(cx + 1e-5))


def contextual_step_t(xy, nk, cfg, weights_t, taps, band_width=0.5):
    """One contextual-loss evaluation with its gradient w.r.t. the prediction half (contextual.py:53-68 + backward): trunk on
    all 2*nk patches ([x | y], the real half without a graph), core, autograd back to the first nk images."""
    x = xy[:nk].clone().requires_grad_(True)
    fx = trunk_forward_t(x, cfg, weights_t, taps)[0]
    with torch.no_grad():
        fy = trunk_forward_t(xy[nk:], cfg, weights_t, taps)[0]
    loss = cx_loss_t(fx, fy, band_width)
    loss.backward()
    return loss.detach(), x.grad


def lpips_step_t(xy, n, cfg, weights_t, taps, head_grads):
    """The LPIPS branch of a 'same' iteration (lpips.py:92-133): VGG16 taps of both halves on torch, the per-layer adaptive head
    and its tap gradients from `head_grads(feats0, feats1)` (the NumPy oracle's lpips_head_grads: elementwise work on the taps),
    then autograd through the trunk back to the first n images."""
    x = xy[:n].clone().requires_grad_(True)
    f0 = trunk_forward_t(x, cfg, weights_t, taps)
    with torch.no_grad():
        f1 = trunk_forward_t(xy[n:], cfg, weights_t, taps)
    loss, dfs = head_grads([t.detach().numpy() for t in f0], [t.numpy() for t in f1])
    torch.autograd.backward(f0, [torch.from_numpy(np.ascontiguousarray(d, np.float32)) for d in dfs])
    return loss, x.grad
