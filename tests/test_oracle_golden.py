"""Pin the NumPy oracle to golden vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from refinit import reference_init


def test_g1_freqs_match_seed0(golden):
    g = golden("g1_embed.npz")
    np.testing.assert_allclose(g["freqs"], np.array(oracle.SEED0_FREQS, np.float32), rtol=1e-6)
    assert int(g["out_dim"]) == 21  # embedder.py:44 with input_dims=1


@pytest.mark.parametrize("tag", ["sq", "rect"])
def test_g1_embed(golden, tag):
    g = golden("g1_embed.npz")
    res = tuple(int(v) for v in g[f"{tag}_res"])
    coords = g[f"{tag}_coords"]
    for k in range(3):
        v = oracle.periodic_warp(coords, g[f"{tag}_angles"][k], g[f"{tag}_periods"][k], res)
        # sin/cos of a phase that went through an f32 mod: a few ulp of the phase
        np.testing.assert_allclose(v, g[f"{tag}_warp"][k], atol=2e-5)
    e = oracle.embed(coords, g[f"{tag}_angles"], g[f"{tag}_periods"], g["freqs"], res)
    assert e.shape == (coords.shape[0], 3 * 462)
    # second-stage arguments reach |f v| ~ 22 rad: errors of the warp are amplified by f
    np.testing.assert_allclose(e, g[f"{tag}_emb"], atol=5e-4)
    # with the golden warp as input the Fourier stage itself is tight
    for k in range(3):
        e_k = oracle.fourier_features(g[f"{tag}_warp"][k], g["freqs"])
        np.testing.assert_allclose(e_k, g[f"{tag}_emb"][:, 462 * k:462 * (k + 1)], atol=3e-6)


def test_g3_snake(golden):
    g = golden("g3_snake.npz")
    np.testing.assert_allclose(oracle.snake(g["z"]), g["a"], atol=2e-6)
    np.testing.assert_allclose(oracle.snake_grad(g["z"]), g["da"], atol=3e-6)


def _small_params(g, tag):
    return {k[len(tag) + 3:]: g[k] for k in g.files if k.startswith(f"{tag}_P_")}


@pytest.mark.parametrize("tag,K", [("small_k3", 3), ("small_k1", 1)])
def test_g2_mlp_small(golden, tag, K):
    g = golden("g2_mlp.npz")
    P = _small_params(g, tag)
    assert set(P) == set(oracle.param_shapes(K, W=32, E=110))
    raw, cache = oracle.mlp_forward(P, g[f"{tag}_emb"], K, E=110)
    np.testing.assert_allclose(raw, g[f"{tag}_raw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(oracle.sigmoid(raw), g[f"{tag}_pred"], atol=1e-5)
    G = oracle.mlp_backward(P, cache, g[f"{tag}_draw"], E=110)
    assert set(G) == set(P)
    for name in P:
        ref = g[f"{tag}_G_{name}"]
        np.testing.assert_allclose(G[name], ref, rtol=2e-3, atol=1e-4 * max(1.0, np.abs(ref).max()),
                                   err_msg=name)


@pytest.mark.parametrize("tag,K", [("full_k3", 3), ("full_k1", 1)])
def test_g2_mlp_full(golden, tag, K):
    g = golden("g2_mlp.npz")
    P = reference_init(K)
    for name, p in P.items():
        s = g[f"{tag}_Psum_{name}"]
        assert abs(p.astype(np.float64).sum() - s[0]) < 1e-6 * max(1.0, s[1]), name
    raw, cache = oracle.mlp_forward(P, g[f"{tag}_emb"], K)
    np.testing.assert_allclose(raw, g[f"{tag}_raw"], rtol=1e-4, atol=2e-5)
    G = oracle.mlp_backward(P, cache, g[f"{tag}_draw"])
    for name in P:
        nrm = g[f"{tag}_Gnorm_{name}"]
        assert abs(np.linalg.norm(G[name].astype(np.float64)) - nrm[0]) <= 1e-3 * nrm[0] + 1e-6, name
        c = g[f"{tag}_Gcorner_{name}"]
        got = G[name].reshape(G[name].shape[0], -1)[:8, :8]
        np.testing.assert_allclose(got, c, rtol=5e-3, atol=1e-4 * max(1.0, np.abs(c).max()), err_msg=name)


def test_macs_per_pixel():
    fwd, train = oracle.mlp_macs_per_pixel(3)
    assert fwd == 1194368 and train == 3110016
    fwd1, train1 = oracle.mlp_macs_per_pixel(1)
    assert fwd1 == 793984 and train1 == 2145408
    # and they agree with the parameter shapes
    sh = oracle.param_shapes(3)
    w = sum(a * b for (a, b) in (s for n, s in sh.items() if n.endswith("weight")))
    assert w == fwd


def test_spline_matches_reference_table(golden):
    g = golden("g4_robust.npz")
    alpha = g["logz_alpha"].astype(np.float32)[None, :]
    from oracle.npp_oracle import _log_partition
    val, _ = _log_partition(alpha, oracle.load_partition_spline())
    np.testing.assert_allclose(val[0], g["logz"], atol=2e-6)


@pytest.mark.parametrize("tag", ["init", "pert"])
@pytest.mark.parametrize("mtag", ["nomask", "mask"])
def test_g4_robust(golden, tag, mtag):
    g = golden("g4_robust.npz")
    la, ls = g[f"{tag}_latent_alpha"], g[f"{tag}_latent_scale"]
    alpha, scale, _, _ = oracle.adaptive_params(la, ls)
    np.testing.assert_allclose(alpha, g[f"{tag}_alpha"], rtol=1e-6)
    np.testing.assert_allclose(scale, g[f"{tag}_scale"], rtol=1e-6)
    np.testing.assert_allclose(oracle.robust_nll(g[f"{tag}_grid_x"], alpha, scale),
                               g[f"{tag}_grid_nll"], rtol=1e-5, atol=2e-6)
    mask = g[f"{tag}_mask"] if mtag == "mask" else None
    loss, dpred, dla, dls = oracle.img2mse_grads(g[f"{tag}_pred"], g[f"{tag}_gt"], la, ls, mask)
    k = f"{tag}_{mtag}"
    np.testing.assert_allclose(loss, g[f"{k}_loss"], rtol=1e-5)
    assert abs(oracle.img2mse(g[f"{tag}_pred"], g[f"{tag}_gt"], la, ls, mask) - loss) < 1e-6
    np.testing.assert_allclose(dpred, g[f"{k}_dpred"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dla, g[f"{k}_dla"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(dls, g[f"{k}_dls"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("lt", ["l2", "robust_loss"])
@pytest.mark.parametrize("mtag", ["nomask", "mask"])
def test_g4b_quadratic_loss_switches(golden, lt, mtag):
    """--loss_type l2 / robust_loss (models/mse_calculator.py:19-23) by the reference's own autograd (make_golden_quad.py)."""
    g = golden("g4b_quad.npz")
    loss, dpred = oracle.img2mse_quad_grads(g["pred"], g["gt"], lt, g["mask"] if mtag == "mask" else None)
    np.testing.assert_allclose(loss, g[f"{lt}_{mtag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(dpred, g[f"{lt}_{mtag}_dpred"], rtol=1e-5, atol=1e-9)


def test_g9_adam(golden):
    g = golden("g9_adam.npz")
    P = {"p0": g["p0_init"].copy(), "p1": g["p1_init"].copy()}
    st = oracle.adam_init(P)
    lr = 5e-4
    gs = 0
    for it in range(5):
        assert abs(lr - g["lrs_used"][it]) < 1e-12
        oracle.adam_step(P, {"p0": g[f"g0_{it}"], "p1": g[f"g1_{it}"]}, st, lr)
        lr = oracle.lr_schedule(gs)
        gs += 1
        np.testing.assert_allclose(P["p0"], g[f"p0_{it}"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(P["p1"], g[f"p1_{it}"], rtol=2e-6, atol=1e-7)
    assert g["lrs_used"][0] == g["lrs_used"][1] == 5e-4  # first two steps at lrate exactly


def test_synthetic_lattice_is_periodic_under_the_embedding():
    H = 256
    angles, periods, shifts = oracle.synthetic_periodicity(H, 3)
    np.testing.assert_allclose(angles[0], [80.54, 168.69], atol=0.01)
    np.testing.assert_allclose(periods[0], [40.77, 36.48], atol=0.01)
    c = np.array([[100, 60]], np.float32)
    d1 = np.array([[8.0, 40.0]], np.float32)  # (dy, dx)
    v0 = oracle.periodic_warp(c, angles[0], periods[0], (H, H), freq_offsets=(0.0,))
    v1 = oracle.periodic_warp(c + d1, angles[0], periods[0], (H, H), freq_offsets=(0.0,))
    np.testing.assert_allclose(v0[:, [1, 2, 4, 5]], v1[:, [1, 2, 4, 5]], atol=2e-3)
    assert oracle.patch_size_from_period(periods[0]) == 64
    assert oracle.patch_size_from_period(oracle.synthetic_periodicity(512, 1)[1][0]) == 96


def test_bf16_round():
    x = np.array([1.0, 1.0078125, 1.00390625, 1.01171875, 3.1415927, -2.7182817], np.float32)
    r = oracle.bf16_round(x)
    assert r[0] == 1.0 and r[1] == 1.0078125
    assert r[2] == 1.0 and r[3] == 1.015625  # ties -> even mantissa
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)
