"""The displacement-search oracle (oracle/npp_search_oracle.py) against golden vectors from the reference's own
generate_possible_shifts / compute_loss / generate_periodicity / feature_search (tests/golden/make_golden_search.py)."""
import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("tag", ["a", "b"])
def test_search_oracle_vs_reference(golden, tag):
    g = golden("g11_search.npz")
    act, mask, rr = g[f"{tag}_act"], g[f"{tag}_mask"], [int(v) for v in g[f"{tag}_rr"]]
    for i in range(rr[0], rr[1], rr[2]):
        r = (i, i + rr[2])
        sh = oracle.possible_shifts(act.shape[1:], r, r)
        assert np.array_equal(sh, g[f"{tag}_shifts_{i}"])
        for edge in (1, 0):
            L = oracle.shift_losses(act, mask, sh, bool(edge))
            np.testing.assert_allclose(L, g[f"{tag}_loss_{i}_{edge}"], rtol=2e-5, atol=2e-4)
            key = f"{tag}_angles_{i}_{edge}"
            a, p, s = oracle.periodicity_from_losses(g[f"{tag}_loss_{i}_{edge}"], sh)
            if key in g.files:
                np.testing.assert_allclose(a, g[key], atol=1e-3)
                np.testing.assert_allclose(p, g[f"{tag}_periods_{i}_{edge}"], rtol=1e-5)
                np.testing.assert_array_equal(np.stack(s), g[f"{tag}_sel_{i}_{edge}"])
            else:
                assert a is None
    A, P, S = oracle.feature_search_oracle(act, mask, rr, True)
    np.testing.assert_allclose(np.array(A), g[f"{tag}_fs_angles"], atol=1e-3)
    np.testing.assert_allclose(np.array(P), g[f"{tag}_fs_periods"], rtol=1e-5)
    np.testing.assert_array_equal(np.array(S), g[f"{tag}_fs_shifts"])
