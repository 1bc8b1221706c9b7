"""config.odgt / PNG formats of the completion task (npp_amd.io; loaders/loaders.py:67-136, NPP_proposal/search.py:228-280)."""
import json
import os

import numpy as np
import pytest

import oracle


def _make(tmp_path, H=64, K=2):
    from npp_amd import io as nio
    img, mask = oracle.synthetic_image(H)
    a, p, s = oracle.synthetic_periodicity(H, K)
    valid = np.ones_like(mask)
    valid[:3] = 0                                            # an invalid band: excluded from both splits
    d = nio.write_detected_dir(str(tmp_path / "detected" / "img0"), img, mask, valid, a, p, s)
    return d, img, mask, valid, a, p, s


def test_odgt_round_trip_and_splits(tmp_path):
    from npp_amd import io as nio
    d, img, mask, valid, a, p, s = _make(tmp_path)
    info = json.loads(open(os.path.join(d, "config.odgt")).readline())
    for key in ("fpath_masked_img", "fpath_valid_mask", "fpath_mask", "fpath_gt_img", "selected_angles", "selected_periods",
                "selected_shifts", "distances"):
        assert key in info                                    # the keys search.py:231-242 writes and loaders.py reads
    r = nio.load_npp_completion(d, p_topk=1)
    assert r["angles"].shape == (1, 2) and len(r["shifts"]) == 1            # top-k truncation (loaders.py:125-127)
    r = nio.load_npp_completion(d, p_topk=3)
    assert r["angles"].shape == (2, 2)
    np.testing.assert_allclose(r["img"], np.uint8(img * 255) / 255.0, atol=1e-7)     # 8-bit PNG, like cv2.imread / 255
    m = (mask * valid)[..., 0]
    assert np.array_equal(r["mask"][..., 0], m)
    assert np.array_equal(r["i_train"], np.stack(np.nonzero(m), 1))                  # np.nonzero order (loaders.py:107)
    assert np.array_equal(r["i_val"], np.stack(np.nonzero((1 - m) * valid[..., 0]), 1))
    assert r["patch_size"] == oracle.patch_size_from_period(p[0])
    np.testing.assert_allclose(r["masked_img"], np.uint8(img * mask * 255) / 255.0, atol=1e-7)   # search.py:250: img * unknown mask
    r2 = nio.load_npp_completion(d, invalid_as_unknown=True)
    assert r2["valid_mask"].min() == 1.0 and r2["i_val"].shape[0] > r["i_val"].shape[0]


def test_load_data_reroots_paths(tmp_path):
    """search.py stores the paths it wrote to; loaders.py:70-77 keeps only the file name and joins it to --datadir."""
    from npp_amd import io as nio
    d, *_ = _make(tmp_path)
    moved = str(tmp_path / "elsewhere")
    os.rename(d, moved)
    info = nio.load_data(moved)
    assert info["fpath_gt_img"] == os.path.join(moved, "gt_img.png") and os.path.exists(info["fpath_gt_img"])
    nio.load_npp_completion(moved)


def test_dump_testset_files(tmp_path):
    from npp_amd import io as nio
    H = 32
    img, mask = oracle.synthetic_image(H)
    pred = np.clip(img + 0.01, 0, 1)
    out = str(tmp_path / "testset_000500")
    nio.dump_testset(out, pred, img, img * mask, mask, np.ones_like(mask))
    assert sorted(os.listdir(out)) == sorted(["pred_rgb_train_img.png", "pred_rgb_val_img.png", "gt_rgb_img.png", "input_rgb_img.png",
                                              "pred_rgb_img.png", "pred_rgb_img_comp.png"])          # train.py:319-328
    from PIL import Image
    comp = np.asarray(Image.open(os.path.join(out, "pred_rgb_img_comp.png")), np.float64) / 255
    known = mask[..., 0] > 0
    np.testing.assert_allclose(comp[known], img[known], atol=1 / 255 + 1e-6)          # known pixels come from the input


@pytest.mark.gpu
def test_train_driver_end_to_end(tmp_path):
    """python -m npp_amd.train on a detected/ directory: config.odgt + PNGs in, testset_* PNG dumps out."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from npp_amd import io as nio, train
    H, K = 256, 3
    img, mask = oracle.synthetic_image(H)
    a, p, s = oracle.synthetic_periodicity(H, K)
    d = nio.write_detected_dir(str(tmp_path / "detected" / "syn"), img, mask, np.ones_like(mask), a, p, s)
    fit = train.main(["--datadir", d, "--basedir", str(tmp_path / "results"), "--p_topk", "3", "--N_iters", "121",
                      "--i_testset", "60", "--i_print", "60", "--rng_mode", "fast", "--random-trunks"])
    out = tmp_path / "results" / "completion_top3" / "syn"
    assert sorted(os.listdir(out)) == ["testset_000060", "testset_000120"]
    assert len(os.listdir(out / "testset_000120")) == 6
    assert fit.psnr() > 26.0                               # torch-default init + freshly drawn Fourier frequencies
    again = train.main(["--datadir", d, "--basedir", str(tmp_path / "results"), "--p_topk", "3", "--N_iters", "121", "--random-trunks"])
    assert again is None                                   # train.py:42-44: an existing result directory is never overwritten


@pytest.mark.gpu
def test_train_driver_ablation_switches(tmp_path):
    """The switches of options/arg_config.py:78-92 the loop is built for, in the reference's store_false / store_true senses: random
    real patches (--no_reg_sampling), no known-region paste (--use_comp), LPIPS off (--use_perceptual_loss on the completion task),
    at the reference's default width (no --netwidth: 512)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from npp_amd import io as nio, train
    H, K = 256, 3
    img, mask = oracle.synthetic_image(H)
    a, p, s = oracle.synthetic_periodicity(H, K)
    d = nio.write_detected_dir(str(tmp_path / "detected" / "syn"), img, mask, np.ones_like(mask), a, p, s)
    fit = train.main(["--datadir", d, "--basedir", str(tmp_path / "results"), "--N_iters", "121", "--i_testset", "120", "--i_print", "60",
                      "--random-trunks", "--no_reg_sampling", "--use_comp", "--use_perceptual_loss", "--use_patch_weight"])
    assert fit.net.width == 512 and fit.patch_sampler.no_reg_sampling and not fit.use_comp and not fit.use_perceptual_loss
    assert fit.skipped < 60 and fit.psnr() > 24.0 and np.isfinite(float(fit.last_patch_loss[0]))


def test_train_driver_refuses_unbuilt_ablations(tmp_path):
    from npp_amd import train
    for flags in (["--netdepth", "4"], ["--activation", "relu"], ["--normalize_type", "3"], ["--loss_type", "mse"]):
        with pytest.raises(SystemExit, match="ablation"):
            train.main(["--datadir", str(tmp_path), "--random-trunks"] + flags)
    assert train.parse(["--datadir", "x", "--normalize_type", "2"]).normalize_type == 2                # accepted since round 5 (tanh output)
    a = train.parse(["--datadir", "x", "--loss_type", "l2", "--use_adaptive_perceptual_loss"])      # built in round 4
    assert a.loss_type == "l2" and a.use_adaptive_perceptual_loss is False
    a = train.parse(["--datadir", "x"])
    assert a.loss_type == "robust_loss_adaptive" and a.use_adaptive_perceptual_loss is True
    assert a.netwidth == 512 and a.use_comp is True and a.use_perceptual_loss is False and a.no_reg_sampling is False


def test_train_driver_requires_trunk_weights(tmp_path):
    """Without pretrained VGG / LPIPS weights the driver refuses to run unless --random-trunks is given."""
    from npp_amd import train
    with pytest.raises(SystemExit) as e:
        train.main(["--datadir", str(tmp_path), "--basedir", str(tmp_path / "r")])
    assert "--random-trunks" in str(e.value)


def test_trunk_state_dict_key_forms_and_strictness():
    """A torchvision state_dict is accepted in both key forms ('features.N.*' and 'N.*'); a missing or mis-shaped convolution
    raises instead of silently keeping its random init."""
    torch = pytest.importorskip("torch")
    from npp_amd import losses
    ref = losses._make_features(losses._VGG19)
    sd = {k: torch.randn_like(v) for k, v in ref.state_dict().items()}
    full = {"features." + k: v for k, v in sd.items()}
    full["classifier.0.weight"] = torch.zeros(3, 3)
    for form in (sd, full):
        t = losses._Trunk(losses._VGG19, taps=(17,), state_dict=form)
        for k, v in sd.items():
            assert torch.equal(t.features.state_dict()[k], v)
    short = dict(sd)
    short.pop("14.weight")
    with pytest.raises(KeyError):
        losses._Trunk(losses._VGG19, taps=(17,), state_dict=short)
    bad = dict(sd)
    bad["0.weight"] = torch.zeros(64, 3, 5, 5)
    with pytest.raises(ValueError):
        losses._Trunk(losses._VGG19, taps=(17,), state_dict=bad)
    with pytest.warns(UserWarning, match="RANDOM"):
        losses._warned_random.clear()
        losses._Trunk(losses._VGG19, taps=(17,))


def test_blur_map_vs_reference(golden):
    """io.get_blur_map (vectorised) against NPP_remapping/blur_detection.py:14-60 (g13_blur.npz)."""
    from npp_amd import io as nio
    g = golden("g13_blur.npz")
    bm, clear = nio.get_blur_map(g["img"], thresh=50)
    np.testing.assert_allclose(bm, g["blur_map"], atol=1e-9)
    assert np.array_equal(clear, g["clear"])


def test_pretrained_weight_discovery(tmp_path, monkeypatch):
    """npp_amd.weights: the torchvision checkpoints are looked up where the reference's `pretrained=True` / README leave them (working
    directory, torch hub cache); an explicit path wins; the LPIPS lin layers ship with the package and equal the ones the reference's
    own LPIPS module carried when the g7 golden was generated."""
    import argparse
    from npp_amd import weights
    hub = tmp_path / "th" / "hub" / "checkpoints"
    hub.mkdir(parents=True)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "th"))
    monkeypatch.chdir(tmp_path)
    assert weights.find_checkpoint("vgg19") is None
    (hub / "vgg19-dcbb9e9d.pth").write_bytes(b"x")
    (tmp_path / "alexnet-owt-4df8aa71.pth").write_bytes(b"x")
    assert weights.find_checkpoint("vgg19") == str(hub / "vgg19-dcbb9e9d.pth")
    assert weights.find_checkpoint("alexnet") == str(tmp_path / "alexnet-owt-4df8aa71.pth")
    other = tmp_path / "mine.pth"
    other.write_bytes(b"y")
    assert weights.find_checkpoint("vgg19", str(other)) == str(other)
    with pytest.raises(FileNotFoundError):
        weights.find_checkpoint("vgg16", str(tmp_path / "absent.pth"))
    args = argparse.Namespace(vgg19=None, vgg16=None)
    with pytest.raises(SystemExit, match="vgg16"):
        weights.resolve(args, ["vgg19", "vgg16"], random_ok=False)
    assert weights.resolve(args, ["vgg19", "vgg16"], random_ok=True) == ["vgg16"] and args.vgg19.endswith("vgg19-dcbb9e9d.pth")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_lpips.npz"))
    for k, lin in enumerate(weights.lpips_lin("vgg")):
        assert np.array_equal(lin, g[f"lin{k}"].reshape(-1))
    assert [v.shape[0] for v in weights.lpips_lin("alex")] == [64, 192, 384, 256, 256]


def test_lattice_drawing_and_ltrb():
    """io.draw_lattice (utils/periodicity_visualizer.py:30-66): both line families over the whole canvas, R = G = 255 with B untouched
    (the reference draws on all channels but the last); io.mask2ltrb (utils/miscs.py:17-20)."""
    from npp_amd import io as nio
    m = np.zeros((40, 60))
    m[5:30, 10:50] = 1
    assert nio.mask2ltrb(m) == (10, 5, 49, 29)
    img = np.full((64, 96, 3), 7, np.uint8)
    out = nio.draw_lattice(img, (3, 2), (16.0, 0.0), (0.0, 12.0), thickness=1)
    hit = (out[..., 0] == 255) & (out[..., 1] == 255)
    assert (out[..., 2] == 7).all() and (out[~hit] == 7).all()
    assert hit[:, 3].all() and hit[:, 19].all() and hit[:, 83].all() and hit[:, 10].sum() == len(range(2, 64, 12))   # column 10: crossings with the rows only
    assert hit[2, :].all() and hit[14, :].all() and hit[62, :].all()
    assert hit.sum() == 6 * 64 + 6 * 96 - 36                           # 6 vertical + 6 horizontal lines, 36 crossings counted once
    with pytest.raises(np.linalg.LinAlgError):
        nio.draw_lattice(img, (0, 0), (4.0, 2.0), (8.0, 4.0))          # collinear shifts span no lattice
