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
