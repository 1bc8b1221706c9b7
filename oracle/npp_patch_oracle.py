"""NumPy restatement of the patch-loss half of the path: integer glimpse crops, the
periodicity-guided GridPatchSampler, the contextual-loss core and the LPIPS head.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/g5-g7.

VGG16/VGG19 trunks: torchvision's pretrained weights are not part of /root/reference, so parity
for CX / LPIPS is stated from the feature tensors onward ("parity unpinned for the
VGG trunks").  The STRUCTURE of the trunks (3x3/pad-1 convolutions + ReLU, 2x2 max-pools, which
torchvision `features` indices are tapped) is restated at the bottom of this file from
contextual_loss/modules/vgg.py:16-21,30-36 and lpips/pretrained_networks.py:96-134, and pinned
against torch's own F.conv2d / F.max_pool2d (the third-party ops those modules call) in
tests/test_oracle_trunk.py with arbitrary weights.
"""
from __future__ import annotations

import numpy as np

from .npp_oracle import F32, adaptive_params, robust_nll, robust_nll_grads, load_partition_spline

__all__ = ["patch_compose", "patch_compose_bwd", "sampler_returns_to_crops", "extract_glimpse_int", "GridPatchSamplerOracle", "cx_forward", "cx_backward", "normalize_tensor",
           "lpips_head", "lpips_head_grads", "scaling_layer", "style_loss_grads", "VGG19_CX_CFG", "VGG16_LPIPS_CFG", "VGG19_CX_TAPS",
           "VGG16_LPIPS_TAPS", "conv3x3", "conv3x3_dgrad", "maxpool2", "maxpool2_bwd", "trunk_forward", "trunk_backward", "conv3x3_gemm", "conv3x3_dgrad_gemm"]


# --------------------------------------------------------------------------
# a9: utils/extract_glimpse.py:7-79 with mode='nearest', padding 'zeros',
# normalized=False, centered=False, as models/sampler.py:171-178,284-291 calls it
# --------------------------------------------------------------------------
def extract_glimpse_int(img_chw, centres_yx, P):
    """img (C,H,W); centres (M,2) (row, col) -> (M,C,P,P).  The reference passes offsets as
    (x, y) and samples pixel index  c - P/2 + j  (grid_sample align_corners=False turns the
    half-pixel offset of xs = arange(w) - (w-1)/2 into an exact integer for even P);
    out-of-image samples are zero."""
    img = np.asarray(img_chw, dtype=F32)
    C, H, W = img.shape
    cen = np.rint(np.asarray(centres_yx, dtype=np.float64)).astype(np.int64)
    out = np.zeros((cen.shape[0], C, P, P), dtype=F32)
    for m, (cy, cx) in enumerate(cen):
        y0, x0 = cy - P // 2, cx - P // 2
        ys, xs = np.arange(y0, y0 + P), np.arange(x0, x0 + P)
        oy, ox = np.nonzero((ys >= 0) & (ys < H))[0], np.nonzero((xs >= 0) & (xs < W))[0]
        if oy.size and ox.size:
            out[m][:, oy[:, None], ox[None, :]] = img[:, ys[oy][:, None], xs[ox][None, :]]
    return out


class GridPatchSamplerOracle:
    """models/sampler.py:8-354, periodicity-guided mode (no_reg_sampling=False), literal
    restatement: every candidate is cropped and its unknown pixels counted, like the
    reference does.  RNG: the NumPy RandomState passed in (the reference uses the global
    np.random; same MT19937 stream for the same seed), same call order."""

    def __init__(self, img_hw3, mask_hw1, N_samples, patch_size, pool_train, pool_val, selected_shifts, rng):
        self.img = np.transpose(np.asarray(img_hw3, F32), (2, 0, 1))        # sampler.py:29 permute
        self.mask = np.transpose(np.asarray(mask_hw1, F32), (2, 0, 1))
        self.H, self.W = self.img.shape[1:]
        s = selected_shifts[0]                                               # top-1 only (:32)
        self.shifts = [np.array([sh[1], sh[0]], np.float64) for sh in s]     # (dy, dx) (:35)
        self.rng = rng
        self.reset_patchsize(patch_size, N_samples)
        self.reset_pool(pool_train, pool_val)

    def reset_patchsize(self, patch_size, N_samples):
        self.N = int(N_samples)
        self.P = int(patch_size)
        self.half = self.P // 2
        a, b = np.meshgrid(np.arange(-10, 10), np.arange(-10, 10), indexing="ij")   # :90-93
        self.perm_a, self.perm_b = a.reshape(-1), b.reshape(-1)
        self.perm_dist = (np.abs(a) + np.abs(b)).reshape(-1)

    def reset_pool(self, pool_train, pool_val):
        def valid(pool):                                                     # :110-121
            pool = np.asarray(pool)
            h = self.half
            ok = (pool[:, 0] > h) & (pool[:, 0] < self.H - (h + 1)) & (pool[:, 1] > h) & (pool[:, 1] < self.W - (h + 1))
            return pool[ok]
        self.pool_train, self.pool_val = valid(pool_train), valid(pool_val)

    def sample_patch_fake(self, mode):                                       # :242-293
        pool = self.pool_train if mode == "train" else self.pool_val
        sel = self.rng.choice(pool.shape[0], size=[self.N], replace=False)
        cen = pool[sel].astype(np.int64)
        h = self.half
        grids = np.stack([np.stack(np.meshgrid(np.arange(c[0] - h, c[0] + h), np.arange(c[1] - h, c[1] + h),
                                               indexing="ij"), -1) for c in cen])
        return (extract_glimpse_int(self.img, cen, 2 * h), extract_glimpse_int(self.mask, cen, 2 * h), grids, cen)

    def sample_patch_real(self, centres, topk, invalid_ratio):               # :127-237
        P2 = 2 * self.half
        imgs, masks, weights, chosen_d = [], [], [], []
        topk_min = topk
        for i in range(self.N):
            cand = centres[i][None].astype(np.float64) + self.perm_a[:, None] * self.shifts[0][None] \
                + self.perm_b[:, None] * self.shifts[1][None]
            ok = (cand[:, 0] > 0) & (cand[:, 0] < self.H - 1) & (cand[:, 1] > 0) & (cand[:, 1] < self.W - 1)
            cand, dist = cand[ok], self.perm_dist[ok].astype(np.float64)
            m = extract_glimpse_int(self.mask, cand, P2)
            good = ~((m < 0.5).sum(axis=(1, 2, 3)) > P2 * P2 * invalid_ratio)     # :181
            cand, dist = cand[good], dist[good]
            dist = dist.copy()
            dist[dist == 0] = 10000                                           # exclude itself (:197)
            if min(len(dist) - 1, topk) < topk_min:
                topk_min = min(len(dist) - 1, topk)
                if topk_min <= 0:
                    return None, None, None, 0, None
            order = np.argsort(dist, kind="stable")[:topk_min]                # torch.topk(largest=False)
            d = dist[order]
            inv = 1.0 / d
            weights.append((inv / inv.sum()).astype(F32))
            imgs.append(extract_glimpse_int(self.img, cand[order], P2))
            masks.append(extract_glimpse_int(self.mask, cand[order], P2))
            chosen_d.append(d)
        if topk_min < topk:                                                   # :213-217
            weights = [w[:topk_min] for w in weights]
            imgs = [x[:topk_min] for x in imgs]
            masks = [x[:topk_min] for x in masks]
            chosen_d = [d[:topk_min] for d in chosen_d]
        real = np.transpose(np.stack(imgs), (0, 1, 3, 4, 2))                  # (N,k,P,P,3) (:234)
        rmask = np.transpose(np.stack(masks), (0, 1, 3, 4, 2))
        return real, rmask, np.concatenate(weights), topk_min, np.stack(chosen_d)

    def sample_patches(self, topk, invalid_ratio):                            # :297-354
        prob = self.rng.uniform(0, 1)
        dists = None
        if prob < 0.5:
            mode = "val"
        elif 0.5 < prob < 0.8:
            mode = "train"
        else:
            mode = "same"
        fake, fmask, grids, cen = self.sample_patch_fake("val" if mode == "val" else "train")
        if mode == "same":
            real = np.transpose(fake, (0, 2, 3, 1))[:, None]
            rmask = np.transpose(fmask, (0, 2, 3, 1))[:, None]
            k, w = 1, np.ones(self.N, F32)
        else:
            real, rmask, w, k, dists = self.sample_patch_real(cen, topk, invalid_ratio)
        if k == 0:
            return dict(k=0, mode=mode)
        fake = np.tile(fake[:, None], (1, k, 1, 1, 1))
        fmask = np.tile(fmask[:, None], (1, k, 1, 1, 1))
        return dict(real=real, real_mask=rmask, fake=fake, fake_mask=fmask, coords=grids, mode=mode, k=k, weight=w,
                    centres=cen, dists=dists)


# --------------------------------------------------------------------------
# a12: contextual_loss/functional.py:9-63,127-163 (loss_type 'cosine')
# --------------------------------------------------------------------------
def _cx_common(x, y, band_width):
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    N, C = x.shape[:2]
    mu = y.mean(axis=(0, 2, 3), keepdims=True, dtype=np.float64).astype(F32)      # :141
    xc = (x - mu).reshape(N, C, -1)
    yc = (y - mu).reshape(N, C, -1)
    nx = np.maximum(np.sqrt((xc * xc).sum(1, keepdims=True)), F32(1e-12))          # F.normalize eps
    ny = np.maximum(np.sqrt((yc * yc).sum(1, keepdims=True)), F32(1e-12))
    xh, yh = xc / nx, yc / ny
    raw = np.einsum("nci,ncj->nij", xh, yh).astype(F32)                            # bmm (:154)
    S = np.clip(raw, 0, 1)
    D = F32(1) - S
    dmin = D.min(axis=2, keepdims=True)                                            # :134
    jmin = D.argmin(axis=2)
    Dt = D / (dmin + F32(1e-5))
    w = np.exp((F32(1) - Dt) / F32(band_width))                                    # :128
    s = w.sum(axis=2, keepdims=True)
    cx = w / s
    return dict(N=N, C=C, xh=xh, yh=yh, nx=nx, raw=raw, D=D, dmin=dmin, jmin=jmin, Dt=Dt, w=w, s=s, cx=cx)


def cx_forward(x, y, band_width=0.5, weight=None):
    t = _cx_common(x, y, band_width)
    cxn = t["cx"].max(axis=1).mean(axis=1)                                         # max over i, mean over j (:53)
    if weight is not None:
        return F32(np.sum(-np.log(cxn * np.asarray(weight, F32) + F32(1e-5))))
    return F32(np.mean(-np.log(cxn + F32(1e-5))))


def cx_backward(x, y, band_width=0.5, weight=None):
    """(loss, dL/dx) by the closed form derived in DESIGN.md section 7: the loss only sees
    cx at the per-column arg-max rows, so dL/dcx is one entry per column."""
    t = _cx_common(x, y, band_width)
    N, C = t["N"], t["C"]
    cx, w, s, D, dmin, jmin, Dt = t["cx"], t["w"], t["s"], t["D"], t["dmin"], t["jmin"], t["Dt"]
    I, J = cx.shape[1:]
    istar = cx.argmax(axis=1)                          # (N,J)
    cmax = np.take_along_axis(cx, istar[:, None, :], 1)[:, 0, :]
    cxn = cmax.mean(axis=1)
    if weight is not None:
        wt = np.asarray(weight, F32)
        loss = F32(np.sum(-np.log(cxn * wt + F32(1e-5))))
        dcxn = -wt / (cxn * wt + F32(1e-5))
    else:
        loss = F32(np.mean(-np.log(cxn + F32(1e-5))))
        dcxn = -(F32(1) / N) / (cxn + F32(1e-5))
    G = np.zeros_like(cx)
    gj = (dcxn / J)[:, None] * np.ones((N, J), F32)
    np.put_along_axis(G, istar[:, None, :], gj[:, None, :], 1)
    A = (G * cx).sum(axis=2, keepdims=True)
    dw = (G - A) / s
    dDt = dw * (-w / F32(band_width))
    dD = dDt / (dmin + F32(1e-5))
    corr = -(dDt * D).sum(axis=2) / ((dmin[..., 0] + F32(1e-5)) ** 2)             # flows to arg-min of each row
    np.add.at(dD, (np.arange(N)[:, None], np.arange(I)[None, :], jmin), corr)
    raw = t["raw"]
    draw = np.where((raw >= 0) & (raw <= 1), -dD, F32(0))                          # clamp + (1 - s)
    dxh = np.einsum("nij,ncj->nci", draw, t["yh"]).astype(F32)
    xh, nx = t["xh"], t["nx"]
    dxc = (dxh - xh * (xh * dxh).sum(1, keepdims=True)) / nx
    return loss, dxc.reshape(np.asarray(x).shape).astype(F32)


# --------------------------------------------------------------------------
# a13: externel_lib/lpips/lpips.py:92-133 (+ :136-154, __init__.py:42-44)
# --------------------------------------------------------------------------
def scaling_layer(x):
    """lpips.py:136-143 on an input already mapped to [-1,1] (normalize=True, :93-95)."""
    shift = np.array([-.030, -.088, -.188], F32)[None, :, None, None]
    scale = np.array([.458, .448, .450], F32)[None, :, None, None]
    return ((np.asarray(x, F32) - shift) / scale).astype(F32)


def normalize_tensor(f, eps=1e-10):
    f = np.asarray(f, F32)
    nrm = np.sqrt((f * f).sum(axis=1, keepdims=True))
    return f / (nrm + F32(eps)), nrm


def lpips_head(feats0, feats1, lins, latents_alpha, latents_scale):
    """use_robust=True head: per layer channel-unit-normalise, per-channel robust NLL of the
    difference, 1x1 lin conv, spatial mean; sum over layers -> (N,).  mean() of it is the loss
    (NPP_completion/train.py:249)."""
    val = 0
    for f0, f1, lin, la, ls in zip(feats0, feats1, lins, latents_alpha, latents_scale):
        h0, _ = normalize_tensor(f0)
        h1, _ = normalize_tensor(f1)
        N, C, H, W = h0.shape
        d = (h0 - h1).transpose(0, 2, 3, 1).reshape(-1, C)
        alpha, scale, _, _ = adaptive_params(la, ls)
        nll = robust_nll(d, alpha, scale).reshape(N, H, W, C)
        val = val + (nll * np.asarray(lin, F32)[None, None, None, :]).sum(-1).mean(axis=(1, 2))
    return val.astype(F32)


def lpips_head_grads(feats0, feats1, lins, latents_alpha, latents_scale):
    """loss = mean_n val_n; returns (loss, [dL/df0_k], [dL/dlatent_alpha_k], [dL/dlatent_scale_k])."""
    val = lpips_head(feats0, feats1, lins, latents_alpha, latents_scale)
    loss = F32(val.mean())
    dfs, dlas, dlss = [], [], []
    for f0, f1, lin, la, ls in zip(feats0, feats1, lins, latents_alpha, latents_scale):
        h0, n0 = normalize_tensor(f0)
        h1, _ = normalize_tensor(f1)
        N, C, H, W = h0.shape
        d = (h0 - h1).transpose(0, 2, 3, 1).reshape(-1, C)
        alpha, scale, dalpha, dscale = adaptive_params(la, ls)
        dx, da, dc = robust_nll_grads(d, alpha, scale)
        coef = np.asarray(lin, F32)[None, :] / F32(N * H * W)
        dd = (dx * coef).reshape(N, H, W, C).transpose(0, 3, 1, 2)                # dL/dh0
        f0 = np.asarray(f0, F32)
        se = n0 + F32(1e-10)
        df = dd / se - f0 * (dd * f0).sum(1, keepdims=True) / (np.maximum(n0, F32(1e-30)) * se * se)
        dfs.append(df.astype(F32))
        dlas.append(((da * coef).sum(0, keepdims=True, dtype=np.float64) * dalpha).astype(F32))
        dlss.append(((dc * coef).sum(0, keepdims=True, dtype=np.float64) * dscale).astype(F32))
    return loss, dfs, dlas, dlss


# --------------------------------------------------------------------------
# remapping variant: models/style_loss.py:37-74 (use_adaptive=True) from the feature tensors on
# --------------------------------------------------------------------------
def style_loss_grads(A_feats, B_feats, latents_alpha, latents_scale, weight=None):
    """Per level: Gram matrices A A^T, B B^T (:55-58), per-element adaptive robust NLL of their difference with
    num_dims = C^2 latents (:60-65), divided by c*w*h; mean over (N, C^2) -- or, with `weight`, mean over C^2 per sample,
    times weight, summed (:66-69).  Returns (loss, [dL/dA_i], [dL/dlatent_alpha_i], [dL/dlatent_scale_i])."""
    loss = F32(0)
    dAs, dlas, dlss = [], [], []
    for A, B, la, ls in zip(A_feats, B_feats, latents_alpha, latents_scale):
        A = np.asarray(A, F32)
        B = np.asarray(B, F32)
        N, C, H, W = A.shape
        a2, b2 = A.reshape(N, C, H * W), B.reshape(N, C, H * W)
        d = (np.einsum("nik,njk->nij", a2, a2) - np.einsum("nik,njk->nij", b2, b2)).astype(F32).reshape(N, C * C)
        alpha, scale, dalpha, dscale = adaptive_params(la, ls)
        nll = robust_nll(d, alpha, scale)
        dx, da, dc = robust_nll_grads(d, alpha, scale)
        denom = F32(C * W * H)
        if weight is None:
            loss = loss + F32(np.mean(nll / denom))
            coef = np.full((N, 1), 1.0 / (N * C * C * denom), F32)
        else:
            wv = np.asarray(weight, F32).reshape(N)
            loss = loss + F32(np.sum(np.mean(nll / denom, axis=1) * wv))
            coef = (wv / (C * C * denom)).reshape(N, 1).astype(F32)
        G = (dx * coef).reshape(N, C, C)
        dAs.append(np.einsum("nij,njk->nik", G + G.transpose(0, 2, 1), a2).reshape(A.shape).astype(F32))
        dlas.append(((da * coef).sum(0, keepdims=True, dtype=np.float64) * dalpha).astype(F32))
        dlss.append(((dc * coef).sum(0, keepdims=True, dtype=np.float64) * dscale).astype(F32))
    return loss, dAs, dlas, dlss


# --------------------------------------------------------------------------
# a11 / a13 trunks: structure of the frozen convolution stacks
#   contextual_loss/modules/vgg.py:16-21: vgg19.features[0:4 | 4:9 | 9:18] -> relu1_2, relu2_2, relu3_4
#     (ContextualLoss uses vgg_layer='relu3_4' = features index 17, contextual.py:27,64)
#   lpips/pretrained_networks.py:104-115: vgg16.features[0:4 | 4:9 | 9:16 | 16:23 | 23:30]
#     -> relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 = features indices 3, 8, 15, 22, 29
# torchvision's `features` = Conv2d(cin, v, 3, padding=1) + ReLU per number, MaxPool2d(2, 2) per 'M'.
# --------------------------------------------------------------------------
VGG19_CX_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256]
VGG16_LPIPS_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
VGG19_CX_TAPS = (17,)
VGG16_LPIPS_TAPS = (3, 8, 15, 22, 29)


def conv3x3(x, w, b):
    """F.conv2d(x, w, b, padding=1): x (N,Cin,H,W), w (Cout,Cin,3,3) cross-correlation."""
    x = np.asarray(x, F32)
    N, _, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((N, w.shape[0], H, W), np.float64)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("nchw,oc->nohw", xp[:, :, ky:ky + H, kx:kx + W].astype(np.float64), w[:, :, ky, kx].astype(np.float64))
    return (out + np.asarray(b, np.float64)[None, :, None, None]).astype(F32)


def conv3x3_dgrad(dz, w):
    """dL/dx of conv3x3 given dL/dz: correlation with the flipped, transposed kernel."""
    dz = np.asarray(dz, F32)
    N, _, H, W = dz.shape
    dp = np.pad(dz, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((N, w.shape[1], H, W), np.float64)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("nohw,oc->nchw", dp[:, :, ky:ky + H, kx:kx + W].astype(np.float64), w[:, :, 2 - ky, 2 - kx].astype(np.float64))
    return out.astype(F32)


def _im2col(x):
    """(C,H,W) -> (H*W, 9*C) patches of the zero-padded image, column order (ky, kx, c)."""
    C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    cols = np.empty((H, W, 9, C), F32)
    for ky in range(3):
        for kx in range(3):
            cols[:, :, ky * 3 + kx, :] = xp[:, ky:ky + H, kx:kx + W].transpose(1, 2, 0)
    return cols.reshape(H * W, 9 * C)


def conv3x3_gemm(x, w, b):
    """conv3x3 as im2col + fp32 SGEMM per image (same maths as conv3x3, BLAS speed): the form the CPU
    baseline of bench.py times."""
    x = np.asarray(x, F32)
    N, _, H, W = x.shape
    wm = np.ascontiguousarray(np.asarray(w, F32).transpose(2, 3, 1, 0).reshape(-1, w.shape[0]))     # (9*Cin, Cout)
    out = np.empty((N, w.shape[0], H, W), F32)
    for n in range(N):
        out[n] = (_im2col(x[n]) @ wm + np.asarray(b, F32)[None, :]).T.reshape(w.shape[0], H, W)
    return out


def conv3x3_dgrad_gemm(dz, w):
    dz = np.asarray(dz, F32)
    N, _, H, W = dz.shape
    wm = np.ascontiguousarray(np.asarray(w, F32)[:, :, ::-1, ::-1].transpose(2, 3, 0, 1).reshape(-1, w.shape[1]))   # (9*Cout, Cin)
    out = np.empty((N, w.shape[1], H, W), F32)
    for n in range(N):
        out[n] = (_im2col(dz[n]) @ wm).T.reshape(w.shape[1], H, W)
    return out


def maxpool2(x):
    """nn.MaxPool2d(2, 2) (floor mode) -> (pooled, argmax slot 0..3 in scan order, first maximum wins)."""
    N, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    win = np.stack([x[:, :, dy:2 * Ho:2, dx:2 * Wo:2] for dy in (0, 1) for dx in (0, 1)], -1)
    return win.max(-1), win.argmax(-1)          # np.argmax returns the first maximum


def maxpool2_bwd(dy, arg, shape):
    N, C, H, W = shape
    Ho, Wo = H // 2, W // 2
    dx = np.zeros(shape, F32)
    for k, (oy, ox) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        dx[:, :, oy:2 * Ho:2, ox:2 * Wo:2] = np.where(arg == k, dy, 0)
    return dx


def trunk_forward(x, cfg, weights, taps, gemm=False):
    """x (N,3,H,W) already normalised; weights: [(w, b)] per conv.  Returns (tap outputs, cache)."""
    conv = conv3x3_gemm if gemm else conv3x3
    outs, cache, idx, wi = [], [], 0, 0
    for v in cfg:
        if v == "M":
            y, arg = maxpool2(x)
            cache.append(("pool", x.shape, arg))
            x = y
            idx += 1
        else:
            w, b = weights[wi]
            wi += 1
            z = conv(x, w, b)
            x = np.maximum(z, 0)
            cache.append(("conv", w, z > 0))
            idx += 2
            if idx - 1 in taps:
                outs.append(x)
    return outs, cache


def trunk_backward(cfg, cache, taps, tap_grads, gemm=False):
    """dL/dx (the normalised input) from dL/dtap for every tap (None = no gradient)."""
    dgrad = conv3x3_dgrad_gemm if gemm else conv3x3_dgrad
    tap_at, idx, k = {}, 0, 0
    for li, v in enumerate(cfg):
        idx += 1 if v == "M" else 2
        if v != "M" and idx - 1 in taps:
            tap_at[li] = tap_grads[k]
            k += 1
    g = None
    for li in range(len(cfg) - 1, -1, -1):
        if li in tap_at and tap_at[li] is not None:
            g = tap_at[li] if g is None else g + tap_at[li]
        if g is None:
            continue
        c = cache[li]
        if c[0] == "pool":
            g = maxpool2_bwd(g, c[2], c[1])
        else:
            g = dgrad(np.where(c[2], g, 0).astype(F32), c[1])
    return g


# --------------------------------------------------------------------------
# a10: the patch plumbing between the prediction and the patch losses
#   NPP_completion/train.py:200-236 (contextual inputs), :241-250 (LPIPS inputs); its backward is what autograd forms
#   Pinned by tests/golden/g8p_patch_io.npz (the reference's own tensors for a 'val', a 'train' and a 'same' iteration).
# --------------------------------------------------------------------------
def patch_compose(pred_rows, real, rmask, fake, fmask, n_p, k, P, source, use_comp=True):
    """Inputs as models/sampler.py GridPatchSampler.sample_patches returns them: real (n_p, k, P, P, 3), rmask (n_p, k, P, P, 1),
    fake (n_p, k, 3, P, P) (the fake patch tiled k times, :219-226), fmask (n_p, k, 1, P, P); pred_rows (n_p P^2, 3) = the
    network's prediction on the patch rows.  -> (x_in, y_in, lp0, lp1): the (n_p k, 3, P, P) tensors handed to the contextual
    loss's trunk and, for source 'same', to percepLoss (else None)."""
    pred_rows = np.asarray(pred_rows, F32)
    pp = pred_rows.reshape(n_p, 1, P, P, 3).transpose(0, 1, 4, 2, 3)                                   # :201
    pp = np.tile(pp, (1, k, 1, 1, 1)).reshape(-1, 3, P, P)                                             # :203
    real_p = np.asarray(real, F32).reshape(-1, k, 3).reshape(n_p, k, P, P, 3).transpose(0, 1, 4, 2, 3).reshape(-1, 3, P, P)   # :206-208
    rm = np.asarray(rmask, F32).transpose(0, 1, 4, 2, 3).reshape(-1, 1, P, P)                          # :213-214
    fk, fm = np.asarray(fake, F32).reshape(-1, 3, P, P), np.asarray(fmask, F32).reshape(-1, 1, P, P)
    if source == "val" and use_comp:                                                                   # :228-232: known pixels of the fake patch, prediction elsewhere
        x_in = ((fk * fm + pp * (F32(1) - fm)) * rm).astype(F32)
    else:
        x_in = (pp * rm).astype(F32)                                                                   # :234-236
    y_in = (real_p * rm).astype(F32)
    lp0 = lp1 = None
    if source == "same":                                                                               # :241-247
        lp0, lp1 = (pp * rm).astype(F32), (fk * rm).astype(F32)
    return x_in, y_in, lp0, lp1


def patch_compose_bwd(dx_in, dlp0, rmask, fmask, n_p, k, P, source, use_comp=True):
    """dL/dpred on the patch rows (n_p P^2, 3) from dL/dx_in (n_p k, 3, P, P) of the contextual branch and, for 'same', dL/dlp0 of
    the LPIPS branch: the adjoint of patch_compose -- mask products back, then the k tiles of a patch summed (the `tile` of :203)."""
    rm = np.asarray(rmask, F32).transpose(0, 1, 4, 2, 3).reshape(-1, 1, P, P)
    fm = np.asarray(fmask, F32).reshape(-1, 1, P, P)
    g = np.asarray(dx_in, np.float64) * rm
    if source == "val" and use_comp:
        g = g * (1.0 - fm)
    if dlp0 is not None:
        g = g + np.asarray(dlp0, np.float64) * rm
    g = g.reshape(n_p, k, 3, P, P).sum(axis=1)                       # the k copies of a fake patch share one prediction
    return g.transpose(0, 2, 3, 1).reshape(n_p * P * P, 3).astype(F32)


def sampler_returns_to_crops(real, rmask, fake, fmask):
    """The reference-shaped sampler returns -> the contiguous crops the C ABI takes (include/npp_hip.h npp_patch_compose_fwd): real
    (n_p k, 3, P, P), rmask (n_p k, 1, P, P), fake (n_p, 3, P, P), fmask (n_p, 1, P, P) -- the fake patch once, not tiled."""
    real, rmask, fake, fmask = (np.asarray(v, F32) for v in (real, rmask, fake, fmask))
    n_p, k, P = real.shape[0], real.shape[1], real.shape[2]
    return (np.ascontiguousarray(real.transpose(0, 1, 4, 2, 3).reshape(n_p * k, 3, P, P)),
            np.ascontiguousarray(rmask.transpose(0, 1, 4, 2, 3).reshape(n_p * k, 1, P, P)),
            np.ascontiguousarray(fake[:, 0]), np.ascontiguousarray(fmask[:, 0]))
