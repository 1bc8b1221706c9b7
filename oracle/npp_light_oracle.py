"""NumPy restatement of the proposal-ranking fit: the is_search embedders, NPP_Net_light and the plain
LPIPS head.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/g10_light.npz, generated from the
reference's own modules (tests/golden/make_golden_light.py).
"""
from __future__ import annotations

import numpy as np

from .npp_oracle import F32, periodic_warp, fourier_features, snake, snake_grad, sigmoid
from .npp_patch_oracle import normalize_tensor

__all__ = ["search_pos_embed", "search_periodic_embed", "light_param_shapes", "light_forward", "light_backward", "lpips_plain"]


def search_pos_embed(coords_yx, freqs, res):
    """get_embedder(is_search=True) without periodicity (models/embedder.py:76-80): Embedder with input_dims = 2; embed()
    first normalises IN PLACE col 0 by res[0] and col 1 by res[1] to [-1, 1] (:52-54), then cat[x, sin(f0 x), cos(f0 x), ...]
    -> (N, 2 * (1 + 2 * n_freq)) = 42."""
    c = np.asarray(coords_yx, dtype=F32).copy()
    c[:, 0] = (c[:, 0] / F32(res[0]) - F32(0.5)) * F32(2)
    c[:, 1] = (c[:, 1] / F32(res[1]) - F32(0.5)) * F32(2)
    return fourier_features(c, freqs)


def search_periodic_embed(coords_yx, angles_deg, periods, res):
    """get_embedder(is_search=True) with periodicity (embedder.py:84-88): include_input = False drops the two normalised raw
    coordinates of the 22-vector (:110-116), leaving the 20 sin / cos values."""
    v = periodic_warp(coords_yx, angles_deg, periods, res)
    return np.concatenate([v[:, 1:11], v[:, 12:22]], axis=1)


def light_param_shapes(W=256, D=4, in_pos=42, in_per=20):
    """The tensors of NPP_Net_light that the forward uses when len(freq_scales) == 1 (models/networks.py:199-214,216-262):
    periodic_linears[0..D-1] (skip index 4 is never reached for D = 4), feature_linear1, pos_linears[0] with input
    [feature1 (W), input_pos (in_pos)] (:247), rgb_linear.  scale_linears / feature_linear2 / alpha_linear exist in the module
    and the optimiser but never receive a gradient."""
    shapes = {}
    for i in range(D):
        shapes[f"periodic_linears.{i}"] = (W, in_per if i == 0 else W)
    shapes["feature_linear1"] = (W, W)
    shapes["pos_linears.0"] = (W // 2, W + in_pos)
    shapes["rgb_linear"] = (3, W // 2)
    return shapes


def _lin(x, P, name):
    return (x @ P[name + ".weight"].T + P[name + ".bias"]).astype(F32)


def light_forward(P, x_pos, x_per, D=4):
    """NPP_Net_light.forward(x, x_periodic) (networks.py:216-262) -> raw rgb (sigmoid is render's, helpers.py:55-56)."""
    cache = {"x_pos": np.asarray(x_pos, F32), "h_in": [], "z": []}
    h = np.asarray(x_per, F32)
    for i in range(D):
        cache["h_in"].append(h)
        z = _lin(h, P, f"periodic_linears.{i}")
        cache["z"].append(z)
        h = snake(z)
    cache["h_last"] = h
    f1 = _lin(h, P, "feature_linear1")
    hp = np.concatenate([f1, cache["x_pos"]], axis=1)
    cache["hp"] = hp
    zp = _lin(hp, P, "pos_linears.0")
    cache["zp"] = zp
    ap = snake(zp)
    cache["ap"] = ap
    return _lin(ap, P, "rgb_linear"), cache


def light_backward(P, cache, draw, D=4):
    """Gradients of every used tensor given dL/d(raw rgb)."""
    G = {}
    W = P["feature_linear1.weight"].shape[0]

    def wg(name, dz, inp):
        G[name + ".weight"] = (dz.T @ inp).astype(F32)
        G[name + ".bias"] = dz.sum(0).astype(F32)
    wg("rgb_linear", draw, cache["ap"])
    dzp = (draw @ P["rgb_linear.weight"]) * snake_grad(cache["zp"])
    wg("pos_linears.0", dzp, cache["hp"])
    df1 = (dzp @ P["pos_linears.0.weight"])[:, :W]
    wg("feature_linear1", df1, cache["h_last"])
    dh = df1 @ P["feature_linear1.weight"]
    for i in range(D - 1, -1, -1):
        dz = dh * snake_grad(cache["z"][i])
        wg(f"periodic_linears.{i}", dz, cache["h_in"][i])
        dh = dz @ P[f"periodic_linears.{i}.weight"]
    return G


def lpips_plain(feats0, feats1, lins):
    """LPIPS.forward(use_robust=False) from the feature tensors (lpips.py:99-133): channel-unit-normalise, squared difference,
    1x1 lin conv, spatial mean, sum over taps -> (N,)."""
    val = 0
    for f0, f1, lin in zip(feats0, feats1, lins):
        h0, _ = normalize_tensor(f0)
        h1, _ = normalize_tensor(f1)
        d = (h0 - h1) ** 2
        val = val + (d * np.asarray(lin, F32)[None, :, None, None]).sum(1).mean(axis=(1, 2))
    return np.asarray(val, F32)
