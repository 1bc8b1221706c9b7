"""The light-path oracle (oracle/npp_light_oracle.py) against golden vectors produced by the reference's own is_search
embedders, NPP_Net_light and LPIPS.forward(use_robust=False) (tests/golden/make_golden_light.py)."""
import numpy as np

import oracle


def _P(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("sd.")}


def test_search_embedders(golden):
    g = golden("g10_light.npz")
    res = tuple(int(v) for v in g["res"])
    assert int(g["freq_nerf"]) == 42 and int(g["input_ch_periodic"]) == 20
    pos = oracle.search_pos_embed(g["coords"], g["freqs"], res)
    assert pos.shape == g["pos_emb"].shape
    np.testing.assert_allclose(pos, g["pos_emb"], atol=3e-6)
    per = oracle.search_periodic_embed(g["coords"], g["angles"], g["periods"], res)
    assert per.shape == g["per_emb"].shape
    np.testing.assert_allclose(per, g["per_emb"], atol=2e-5)


def test_light_forward_backward(golden):
    g = golden("g10_light.npz")
    P = _P(g)
    shapes = oracle.light_param_shapes(W=64)
    for name, shp in shapes.items():
        assert P[name + ".weight"].shape == shp                     # the reference module's own tensor shapes
    raw, cache = oracle.light_forward(P, g["pos_emb"], g["per_emb"])
    np.testing.assert_allclose(raw, g["raw"], rtol=2e-4, atol=2e-5)
    pred = oracle.sigmoid(raw)
    np.testing.assert_allclose(pred, g["pred"], atol=1e-5)
    dpred = 2.0 * (pred - g["tgt"]) / pred.size
    G = oracle.light_backward(P, cache, dpred * pred * (1 - pred))
    for name in shapes:
        for part in ("weight", "bias"):
            ref = g[f"grad.{name}.{part}"]
            got = G[f"{name}.{part}"]
            assert np.linalg.norm(got - ref) <= 2e-4 * np.linalg.norm(ref) + 1e-9, (name, part)
    for dead in ("scale_linears.0", "feature_linear2", "alpha_linear"):    # in the optimiser, never used (no gradient)
        assert g[f"grad.{dead}.weight"].size == 0


def test_lpips_plain(golden):
    g = golden("g10_light.npz")
    f0 = [g[f"lp_f0_{k}"] for k in range(5)]
    f1 = [g[f"lp_f1_{k}"] for k in range(5)]
    lins = [g[f"lp_lin{k}"] for k in range(5)]
    np.testing.assert_allclose(oracle.lpips_plain(f0, f1, lins), g["lp_val"].reshape(-1), rtol=2e-5)


def test_default_light_init_is_the_references_generator_stream(golden):
    """The weights a candidate fit starts from: torch.manual_seed(0), then the position embedder's 10 Gaussian frequency draws, then
    NPP_Net_light's modules in construction order -- against the reference's own construction (g10c_light_init.npz,
    tests/golden/make_golden_light_init.py): every tensor's shape, first 16 values (bit for bit), sum and sum of squares."""
    from npp_amd.light import default_light_init
    g = golden("g10c_light_init.npz")
    sd = default_light_init(256, 4)
    names = sorted(k[5:] for k in g.files if k.startswith("head."))
    assert sorted(sd) == names
    for k in names:
        a = sd[k].astype(np.float64).reshape(-1)
        assert tuple(g["shape." + k]) == sd[k].shape
        np.testing.assert_array_equal(sd[k].reshape(-1)[:16], g["head." + k])
        assert abs(a.sum() - float(g["sum." + k])) < 1e-9 and abs((a * a).sum() - float(g["sq." + k])) < 1e-9
