"""The product's synthetic-workload definition (npp_amd.synthetic, used by bench.py) equals the oracle's (used by the tests)."""
import numpy as np

import oracle


def test_product_and_oracle_workloads_agree():
    from npp_amd import synthetic as syn
    assert tuple(syn.SEED0_FREQS) == tuple(oracle.SEED0_FREQS)
    for H in (64, 256):
        for seed in (0, 3):
            a, b = syn.synthetic_image(H, seed=seed), oracle.synthetic_image(H, seed=seed)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for K in (1, 3, 5):
            pa, pb = syn.synthetic_periodicity(H, K), oracle.synthetic_periodicity(H, K)
            assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1]) and pa[2] == pb[2]
    for K in (1, 3):
        A, B = syn.init_params(K, seed=2), oracle.init_params(K, seed=2)
        assert set(A) == set(B)
        for k in A:
            assert np.array_equal(A[k].reshape(-1), B[k].reshape(-1)), k
        assert syn.mlp_macs_per_pixel(K) == oracle.mlp_macs_per_pixel(K)
