"""Segmentation variant (NPP_segmentation/train.py): loader + masked blur + the evaluation criteria.
CPU part: host logic.  GPU part: LPIPS(alex, spatial) maps against the reference's own LPIPS.forward (g14_segment.npz) and the
task end to end on a synthetic image with a planted non-periodic region."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle  # noqa: E402


def test_blur_with_mask_properties():
    from npp_amd import io as nio
    rng = np.random.RandomState(0)
    img = rng.rand(40, 50, 3) * 255
    ones = np.ones((40, 50, 1))
    from scipy.ndimage import gaussian_filter
    want = np.stack([gaussian_filter(img[..., c], 3, mode="nearest", truncate=4.0) for c in range(3)], -1)
    np.testing.assert_allclose(nio.blur_with_mask(img, ones), want / (1 + 1e-6), rtol=1e-9)    # full mask: plain Gaussian blur
    mask = ones.copy()
    mask[10:20, 15:30] = 0
    b = nio.blur_with_mask(img, mask)
    assert np.all(b[10:20, 15:30] == 0)                                     # invalid pixels stay zero (utils/ops.py:74)
    const = np.full((40, 50, 3), 77.0)
    np.testing.assert_allclose(nio.blur_with_mask(const, mask)[mask[..., 0] > 0], 77.0, rtol=1e-5)   # normalised: a constant survives


def test_remove_small_objects_and_loader_errors(tmp_path):
    from npp_amd import io as nio, segment
    m = np.zeros((30, 30, 1), bool)
    m[2:4, 2:4] = True                     # 4 pixels
    m[10:25, 10:25] = True                 # 225 pixels
    m[5, 20] = True                        # touches nothing (diagonal neighbours do not connect: connectivity 1)
    out = segment.remove_small_objects(m, min_size=10)
    assert out[10:25, 10:25].all() and not out[2:4, 2:4].any() and not out[5, 20]
    img, mask = oracle.synthetic_image(64)
    a, p, s = oracle.synthetic_periodicity(64, 1)
    d = nio.write_detected_dir(str(tmp_path / "seg"), img, mask, np.ones_like(mask), a, p, s)
    with pytest.raises(FileNotFoundError, match="period_mask.png"):
        nio.load_npp_segmentation(d, 1)
    pm = np.ones((64, 64), np.float32)
    pm[20:40, 20:44] = 0
    out = nio.load_npp_segmentation(d, 1, period_mask=pm, non_period_mask=1 - pm)
    assert out["blur_img"].shape == (64, 64, 3) and out["period_mask"].shape == (64, 64, 1)
    assert out["period_mask"].sum() == pm.sum() and 64 <= out["patch_size"] <= 160


@pytest.fixture(scope="module")
def dev():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import npp_amd
    npp_amd.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_lpips_alex_spatial_vs_reference(dev):
    """segment.lpips_alex_spatial (AlexNet convolutions as im2col + npp_linear_fwd) against the reference's LPIPS.forward
    (spatial=True, retPerLayer=True) on grayscale inputs: all five per-layer maps and their sum."""
    import torch
    from make_golden_segment import alex_weights
    from npp_amd import segment
    g = np.load(os.path.join(ROOT, "tests", "golden", "g14_segment.npz"))
    sd = {}
    for (idx, *_), (w, b) in zip(segment._ALEX, alex_weights(int(g["seed"]))):
        sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"] = w, b
    alex = segment.AlexFeatures(sd, device=dev)
    lins = [g[f"lin{k}"] for k in range(5)]
    val, res = segment.lpips_alex_spatial(torch.from_numpy(g["in0"]).to(dev), torch.from_numpy(g["in1"]).to(dev), alex, lins)
    for k in range(5):
        want = g[f"map{k}"]
        assert np.abs(res[k].cpu().numpy() - want).max() < 2e-4 * max(1.0, np.abs(want).max()), k
    np.testing.assert_allclose(val.cpu().numpy(), g["val"], rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_segmentation_task_end_to_end(dev):
    """CompletionFit(task='segmentation') + segment.segmentation_eval on a lattice image with a planted non-periodic blob:
    the fit (on the blurred image, periodic region = everything outside a candidate rectangle, constant LR) explains the
    periodic half of the candidate rectangle and not the blob."""
    import torch
    from npp_amd import io as nio, segment
    from npp_amd.fit import CompletionFit
    H, K = 256, 1
    img, _ = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    blob = np.zeros((H, H, 1), np.float32)
    yy, xx = np.mgrid[:H, :H]
    blob[((yy - 128) ** 2 + (xx - 100) ** 2) < 30 ** 2] = 1
    img = (img * (1 - blob) + blob * 1.0).astype(np.float32)               # a flat white disc (the lattice's gray stays below ~0.85)
    cand = np.zeros((H, H, 1), np.float32)
    cand[80:176, 50:210] = 1                                                # non-periodic CANDIDATES: the disc + clean lattice
    valid = np.ones((H, H, 1), np.float32)
    blur = nio.blur_with_mask(img * 255.0, valid).astype(np.float32) / 255.0
    fit = CompletionFit(img, 1 - cand, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), device=dev, N_rand=8192,
                        seed=0, shifts=shifts, task="segmentation", masked_img=blur, contextual_weight=0.005,
                        use_perceptual_loss=False, rng_mode="fast", patch_size=64)
    assert fit.net.lr_clock is False
    for _ in range(250):
        fit.step_full()
    assert fit.net.global_step == 0 and fit.net.lr == 5e-4                  # NPP_segmentation/train.py:408: the clock never runs
    pred = fit.render_image().cpu().numpy()
    alex = segment.AlexFeatures(None, device=dev)
    lins = [np.full(c, 1.0 / c, np.float32) for c in (64, 192, 384, 256, 256)]
    r = segment.segmentation_eval(pred, blur, valid, cand, alex, lins, l1_thresh=0.15, lpips_thresh=1e9, lpips_layers=1)
    got = r["non_period_mask_final"][..., 0] > 0
    inter, union = (got & (blob[..., 0] > 0)).sum(), (got | (blob[..., 0] > 0)).sum()
    assert inter / union > 0.6, inter / union                               # the disc is found ...
    clean = (cand[..., 0] > 0) & ~(blob[..., 0] > 0)
    assert (got & clean).sum() < 0.25 * clean.sum()                         # ... and most of the periodic candidates are released
