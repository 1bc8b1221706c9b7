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
def test_alexnet_front_end_pieces_vs_torch(dev):
    """The four launches around the AlexNet products against the torch calls they replace (the reference's own modules are
    nn.Conv2d / nn.MaxPool2d of torchvision's alexnet, pretrained_networks.py:60-96, and F.interpolate / normalize_tensor of
    lpips.py:20-22, lpips/__init__.py:42-44): the rows of every convolution geometry in the net from both source layouts and the
    pooling BIT-EXACT (they move values), the per-pixel head and the upsampling to float rounding -- incl. ragged sizes,
    a non-integer upsampling ratio, a zero feature vector (the 1e-10 in the norm) and the bad-argument paths."""
    import torch
    import torch.nn.functional as F
    from npp_amd import ops
    g = torch.Generator().manual_seed(5)
    for (N, C, H, W, k, st, pd) in [(2, 3, 67, 45, 11, 4, 2), (1, 3, 64, 96, 11, 4, 5), (2, 64, 15, 10, 5, 1, 2), (1, 192, 7, 4, 3, 1, 1),
                                    (1, 5, 3, 3, 3, 1, 0)]:
        x = torch.randn(N, C, H, W, generator=g).to(dev)
        want = F.unfold(x, k, padding=pd, stride=st).transpose(1, 2).reshape(-1, C * k * k)
        ho, wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
        for nhwc in (False, True):
            cols, h_, w_ = ops.im2col(x.permute(0, 2, 3, 1).contiguous() if nhwc else x, k, st, pd, nhwc=nhwc)
            assert (h_, w_) == (ho, wo) and torch.equal(cols, want), (N, C, H, W, k, st, pd, nhwc)
    for (N, C, H, W) in [(2, 64, 15, 10), (1, 192, 7, 7), (1, 3, 3, 3), (1, 7, 4, 9)]:
        x = torch.randn(N, C, H, W, generator=g).to(dev)
        y = ops.maxpool_nhwc(x.permute(0, 2, 3, 1).contiguous(), 3, 2)
        assert torch.equal(y.permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2))
    x = torch.full((1, 3, 3, 1), 1.0, device=dev)
    x[0, 1, 1, 0] = float("nan")
    assert bool(torch.isnan(ops.maxpool_nhwc(x, 3, 2)).all())
    for (N, h, w, H, W) in [(2, 15, 15, 64, 64), (1, 3, 7, 50, 33), (1, 1, 1, 5, 4), (1, 8, 8, 8, 8), (1, 9, 5, 4, 3)]:
        x = torch.randn(N, h, w, generator=g).to(dev)
        want = F.interpolate(x[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0]
        got = ops.resize_bilinear(x, H, W)
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), (N, h, w, H, W)
        twice = ops.resize_bilinear(x, H, W, out=got.clone(), accumulate=True)
        assert float((twice - 2 * want).abs().max()) <= 4e-6 * float(want.abs().max())
    for (N, h, w, C) in [(2, 15, 15, 64), (1, 3, 3, 384), (1, 1, 5, 100)]:
        a, b = torch.randn(N, h, w, C, generator=g).to(dev), torch.randn(N, h, w, C, generator=g).to(dev)
        a[0, 0, 0] = 0.0                                                    # |a| = 0: a / (0 + 1e-10) = 0
        lin = torch.rand(C, generator=g).to(dev)
        na = a / (torch.sqrt(torch.sum(a ** 2, dim=-1, keepdim=True)) + 1e-10)
        nb = b / (torch.sqrt(torch.sum(b ** 2, dim=-1, keepdim=True)) + 1e-10)
        want = ((na - nb) ** 2 * lin).sum(-1)
        got = ops.lpips_spatial_layer(a, b, lin)
        assert bool(torch.isfinite(got).all()) and float((got - want).abs().max()) <= 5e-6 * float(want.abs().max())
    x = torch.zeros(1, 3, 4, 4, device=dev)
    with pytest.raises(RuntimeError):
        ops.im2col(x, 11, 4, 0)                                              # window larger than the padded image
    with pytest.raises(RuntimeError):
        ops.maxpool_nhwc(torch.zeros(1, 2, 2, 3, device=dev), 3, 2)
    with pytest.raises(ValueError):
        ops.im2col(x.permute(0, 2, 3, 1), 3, 1, 1, nhwc=True)               # not contiguous


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
