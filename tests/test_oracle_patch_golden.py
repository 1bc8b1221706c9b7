"""Pin the patch-loss half of the oracle (glimpse crops, sampler, CX core, LPIPS head) to
golden vectors produced by the reference's own code (tests/golden/make_golden_patch.py)."""
import numpy as np
import pytest

import oracle


def test_extract_glimpse_matches_grid_sample(golden):
    g = golden("g5_sampler.npz")
    img = g["glimpse_img"][0]
    offs = g["glimpse_offs_xy"]
    out = oracle.extract_glimpse_int(img, offs[:, ::-1], 16)       # reference offsets are (x, y)
    assert np.array_equal(out, g["glimpse_out"])


def _sampler(H, P, nsamp, g):
    img, mask = oracle.synthetic_image(H)
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_val = np.stack(np.nonzero(1 - mask[..., 0]), 1)
    shifts = [g["shifts"].tolist()]
    rng = np.random.RandomState(0)
    return oracle.GridPatchSamplerOracle(img * mask, mask, nsamp, P, i_train, i_val, shifts, rng), rng


@pytest.mark.parametrize("tag,P,nsamp", [("p64", 64, 2), ("p32", 32, 4)])
def test_sampler_sequence(golden, tag, P, nsamp):
    g = golden("g5_sampler.npz")
    S, rng = _sampler(int(g["H"]), P, nsamp, g)
    assert S.pool_train.shape[0] == int(g[f"{tag}_pool_train_n"])
    assert S.pool_val.shape[0] == int(g[f"{tag}_pool_val_n"])
    code = {"val": 0, "train": 1, "same": 2}
    for it in range(24):
        r = S.sample_patches(topk=3, invalid_ratio=0.3)
        assert code[r["mode"]] == g[f"{tag}_modes"][it], it
        assert r["k"] == g[f"{tag}_k"][it], it
        if r["k"] == 0:
            continue
        assert np.array_equal(r["centres"], g[f"{tag}_centres"][it]), it
        np.testing.assert_allclose(r["fake"].astype(np.float64).sum(), g[f"{tag}_fake_sum"][it], rtol=1e-6)
        np.testing.assert_allclose(r["fake_mask"].astype(np.float64).sum(), g[f"{tag}_fmask_sum"][it], rtol=1e-6)
        if r["mode"] != "same":
            # which patches win among equal lattice distances is backend-defined:
            # compare the sorted 1/d weights
            wv = np.sort(r["weight"].reshape(nsamp, -1), 1)
            np.testing.assert_allclose(wv, g[f"{tag}_weights_sorted"][it][:, :r["k"]], atol=1e-6, err_msg=str(it))
        if it < 3:
            assert np.array_equal(r["fake"][:, 0], g[f"{tag}_fake_{it}"])
            assert np.array_equal(r["fake_mask"][:, 0], g[f"{tag}_fmask_{it}"])
            assert r["real"].shape == g[f"{tag}_real_{it}"].shape
            if r["mode"] == "same":
                assert np.array_equal(r["real"], g[f"{tag}_real_{it}"])
    # the oracle consumed exactly as many random numbers as the reference
    np.testing.assert_array_equal(rng.uniform(0, 1, 4), g[f"{tag}_rng_after"])


@pytest.mark.parametrize("tag", ["a", "b", "same", "w"])
def test_cx_core(golden, tag):
    g = golden("g6_cx.npz")
    w = g[f"{tag}_w"] if f"{tag}_w" in g.files else None
    loss = oracle.cx_forward(g[f"{tag}_x"], g[f"{tag}_y"], 0.5, w)
    np.testing.assert_allclose(loss, g[f"{tag}_loss"], rtol=2e-4, atol=2e-5)
    loss2, dx = oracle.cx_backward(g[f"{tag}_x"], g[f"{tag}_y"], 0.5, w)
    assert abs(loss2 - loss) < 1e-6
    ref = g[f"{tag}_dx"]
    if np.linalg.norm(ref) < 1e-6:       # 'same' patches: cx saturates at 1, the gradient underflows
        assert np.abs(dx - ref).max() < 1e-7
    else:
        err = np.linalg.norm(dx - ref) / np.linalg.norm(ref)
        assert err < 2e-3, err


def test_lpips_head(golden):
    g = golden("g7_lpips.npz")
    f0 = [g[f"f0_{k}"] for k in range(5)]
    f1 = [g[f"f1_{k}"] for k in range(5)]
    lins = [g[f"lin{k}"] for k in range(5)]
    la = [g[f"la{k}"] for k in range(5)]
    ls = [g[f"ls{k}"] for k in range(5)]
    val = oracle.lpips_head(f0, f1, lins, la, ls)
    np.testing.assert_allclose(val, g["val"].ravel(), rtol=2e-5)
    loss, dfs, dlas, dlss = oracle.lpips_head_grads(f0, f1, lins, la, ls)
    np.testing.assert_allclose(loss, g["loss"], rtol=2e-5)
    for k in range(5):
        ref = g[f"df0_{k}"]
        assert np.linalg.norm(dfs[k] - ref) / np.linalg.norm(ref) < 1e-3, k
        np.testing.assert_allclose(dlas[k], g[f"dla{k}"], rtol=5e-3, atol=1e-6)
        np.testing.assert_allclose(dlss[k], g[f"dls{k}"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(oracle.scaling_layer(2 * g["in0"] - 1), g["scaled0"], rtol=1e-6, atol=1e-6)
    assert all((l >= 0).all() for l in lins)          # vendored lin weights are non-negative


def test_lpips_plain_oracle_vs_reference(golden):
    """oracle.lpips_plain (the ranking score's and the non-adaptive loop's head) against the reference's LPIPS.forward(use_robust=False)
    on g7's features (g7b_lpips_plain.npz)."""
    g, gp = golden("g7_lpips.npz"), golden("g7b_lpips_plain.npz")
    val = oracle.lpips_plain([g[f"f0_{k}"] for k in range(5)], [g[f"f1_{k}"] for k in range(5)], [g[f"lin{k}"] for k in range(5)])
    np.testing.assert_allclose(val, gp["val"].ravel(), rtol=2e-5)
    np.testing.assert_allclose(val.mean(), gp["loss"], rtol=2e-5)


def test_style_loss_oracle_vs_reference(golden):
    """oracle.style_loss_grads against models/style_loss.py:37-74 (g12_style.npz: the reference class fed prepared features)."""
    import oracle
    g = golden("g12_style.npz")
    A = [g[f"A{i}"] for i in range(3)]
    B = [g[f"B{i}"] for i in range(3)]
    la = [g[f"la{i}"] for i in range(3)]
    ls = [g[f"ls{i}"] for i in range(3)]
    for tag, w in (("mean", None), ("weighted", g["weighted_w"])):
        loss, dA, dla, dls = oracle.style_loss_grads(A, B, la, ls, w)
        np.testing.assert_allclose(loss, g[f"{tag}_loss"], rtol=2e-5)
        for i in range(3):
            assert np.linalg.norm(dA[i] - g[f"{tag}_dA{i}"]) < 2e-4 * np.linalg.norm(g[f"{tag}_dA{i}"])
            assert np.linalg.norm(dla[i] - g[f"{tag}_dla{i}"]) < 2e-3 * np.linalg.norm(g[f"{tag}_dla{i}"]) + 1e-9
            assert np.linalg.norm(dls[i] - g[f"{tag}_dls{i}"]) < 2e-3 * np.linalg.norm(g[f"{tag}_dls{i}"]) + 1e-9


@pytest.mark.parametrize("source", ["val", "train", "same"])
def test_patch_plumbing_vs_reference_tensors_g8p(golden, source):
    """SURVEY 8 row a10, directly: oracle.patch_compose / patch_compose_bwd against the tensors the reference's own loop produced
    (NPP_completion/train.py:200-236, :241-250 and autograd through them; tests/golden/make_golden_patch_io.py) for the first
    'val' (use_comp), 'train' and 'same' iteration: the inputs of both losses bit for bit, dL/dpred on the patch rows to fp32
    summation order."""
    g = golden("g8p_patch_io.npz")
    P, n_p, k = int(g["P"]), int(g["n_p"]), int(g[f"{source}_k"])
    x_in, y_in, lp0, lp1 = oracle.patch_compose(g[f"{source}_pred_rows"], g[f"{source}_real"], g[f"{source}_rmask"], g[f"{source}_fake"],
                                                 g[f"{source}_fmask"], n_p, k, P, source)
    np.testing.assert_array_equal(x_in, g[f"{source}_x_in"])
    np.testing.assert_array_equal(y_in, g[f"{source}_y_in"])
    dlp0 = None
    if source == "same":
        np.testing.assert_array_equal(lp0, g["same_lp0"])
        np.testing.assert_array_equal(lp1, g["same_lp1"])
        np.testing.assert_array_equal(lp0, x_in)                  # 'same': the two losses see the same tensors (what the
        np.testing.assert_array_equal(lp1, y_in)                  # build exploits: one [x | y] batch feeds both trunks)
        dlp0 = g["same_dlp0"]
    else:
        assert lp0 is None and lp1 is None
    d = oracle.patch_compose_bwd(g[f"{source}_dx_in"], dlp0, g[f"{source}_rmask"], g[f"{source}_fmask"], n_p, k, P, source)
    want = g[f"{source}_dpred_rows"]
    assert np.abs(want).max() > 0
    np.testing.assert_allclose(d, want, rtol=2e-6, atol=1e-12)
    # 'val' with use_comp: the prediction only shows through where the fake patch is unknown
    if source == "val":
        fm = np.broadcast_to(g["val_fmask"][:, 0].transpose(0, 2, 3, 1).reshape(-1, 1), want.shape)
        assert np.all(want[fm == 1] == 0)
