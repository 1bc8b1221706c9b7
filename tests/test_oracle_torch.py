"""The PyTorch-CPU restatement used for the bench's CPU baseline (oracle/npp_torch_oracle.py) against the NumPy oracle:
embedding, forward values and every parameter gradient, K = 1 and K = 3."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle import npp_torch_oracle as T


@pytest.mark.parametrize("K", [1, 3])
def test_torch_restatement_matches_numpy_oracle(K):
    H = 64
    rng = np.random.RandomState(3)
    angles, periods, _ = oracle.synthetic_periodicity(256, K)
    c = np.stack([rng.randint(0, H, 48), rng.randint(0, H, 48)], 1)
    emb_np = oracle.embed(c, angles, periods, oracle.SEED0_FREQS, (H, H))
    emb_t = T.embed_t(torch.from_numpy(c), angles, periods, oracle.SEED0_FREQS, (H, H))
    assert np.abs(emb_t.numpy() - emb_np).max() < 2e-4          # fp32 sin/cos of arguments up to ~25 rad, two libms
    P = oracle.init_params(K, seed=1)
    Pt = T.params_t(P)
    raw_np, cache = oracle.mlp_forward(P, emb_np, K)
    raw_t = T.mlp_forward_t(Pt, torch.from_numpy(emb_np), K)
    assert np.abs(raw_t.detach().numpy() - raw_np).max() < 2e-4
    draw = rng.randn(48, 3).astype(np.float32)
    raw_t.backward(torch.from_numpy(draw))
    G = oracle.mlp_backward(P, cache, draw)
    for k, g in G.items():
        rel = np.linalg.norm(Pt[k].grad.numpy() - g) / max(np.linalg.norm(g), 1e-20)
        assert rel < 2e-4, (k, rel)


def test_torch_patch_half_matches_numpy_oracle():
    """bench.py's cpu_baseline patch half (F.conv2d trunks + autograd, torch contextual core) against the NumPy oracle's trunk and
    closed-form contextual backward: features, loss and dL/d(prediction patches)."""
    rng = np.random.RandomState(5)
    cfg, taps = oracle.VGG19_CX_CFG, oracle.VGG19_CX_TAPS
    ws, cin = [], 3
    for v in cfg:
        if v != "M":
            ws.append(((rng.randn(v, cin, 3, 3) * np.sqrt(2.0 / (9 * cin))).astype(np.float32), (rng.randn(v) * 0.05).astype(np.float32)))
            cin = v
    nk, P = 2, 16
    xy = rng.rand(2 * nk, 3, P, P).astype(np.float32)
    f, cache = oracle.trunk_forward(xy, cfg, ws, taps, gemm=True)
    loss_np, dfx = oracle.cx_backward(f[0][:nk], f[0][nk:])
    cache_x = [(c[0], (nk,) + tuple(c[1][1:]), c[2][:nk]) if c[0] == "pool" else (c[0], c[1], c[2][:nk]) for c in cache]
    dx_np = oracle.trunk_backward(cfg, cache_x, taps, [dfx], gemm=True)
    wt = T.trunk_weights_t(ws)
    ft = T.trunk_forward_t(torch.from_numpy(xy), cfg, wt, taps)[0]
    assert np.abs(ft.numpy() - f[0]).max() < 1e-4 * max(1.0, np.abs(f[0]).max())
    loss_t, dx_t = T.contextual_step_t(torch.from_numpy(xy), nk, cfg, wt, taps)
    assert abs(float(loss_t) - float(loss_np)) < 1e-4 * max(1.0, abs(float(loss_np)))
    rel = np.linalg.norm(dx_t.numpy() - dx_np) / max(np.linalg.norm(dx_np), 1e-20)
    assert rel < 5e-3, rel
