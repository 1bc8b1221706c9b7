"""The feature extraction in front of the periodicity search (NPP_proposal/feature_searching.py:14-75,158-204): the NumPy
restatement of the OpenCV calls (npp_amd.cvlite -- PARITY UNPINNED: cv2 is not in this image, so these are closed-form cases of
OpenCV's documented arithmetic, not comparisons with its output) and, on the GPU, the complete
image -> features -> edges -> displacement search -> (angles, periods) chain on a synthetic lattice."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from npp_amd import cvlite  # noqa: E402


def test_gray_fixed_point_known_values():
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [10, 20, 30]]], np.uint8)
    # OpenCV's documented results for the primaries (0.299 / 0.587 / 0.114 in 14-bit fixed point, round to nearest)
    assert cvlite.rgb_to_gray_u8(px).tolist() == [[255, 76, 150, 29, 0, 18]]


def test_resize_linear_exact_halving_is_the_rounded_block_mean():
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (12, 20)).astype(np.uint8)
    got = cvlite.resize_linear_u8(a, (10, 6))
    blk = a.astype(np.int64).reshape(6, 2, 10, 2).sum((1, 3))
    assert np.array_equal(got, ((blk + 2) >> 2).astype(np.uint8))
    assert np.array_equal(cvlite.resize_linear_u8(a, (20, 12)), a)                 # identity size: taps (1, 0)
    c = np.full((7, 9), 123, np.uint8)
    assert np.array_equal(cvlite.resize_linear_u8(c, (5, 4)), np.full((4, 5), 123, np.uint8))   # constants survive any ratio
    up = cvlite.resize_linear_u8(np.array([[0, 100]], np.uint8), (4, 1))           # sample points -0.25, .25, .75, 1.25
    assert up.tolist() == [[0, 25, 75, 100]]


def test_resize_nearest_index_rule():
    a = np.arange(35).reshape(5, 7)
    got = cvlite.resize_nearest(a, (3, 2))                                          # dsize = (w, h)
    assert got.tolist() == [[0, 2, 4], [14, 16, 18]]                               # floor(dst * src / dst_size)
    assert np.array_equal(cvlite.resize_nearest(a, (7, 5)), a)


def test_gaussian_blur3():
    assert np.array_equal(cvlite.gaussian_blur3_u8(np.full((6, 5), 200, np.uint8)), np.full((6, 5), 200, np.uint8))
    a = np.zeros((5, 5), np.uint8)
    a[2, 2] = 160
    want = np.zeros((5, 5), np.int64)
    want[1:4, 1:4] = (160 * np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]]) + 8) >> 4
    assert np.array_equal(cvlite.gaussian_blur3_u8(a), want.astype(np.uint8))
    b = np.zeros((4, 4), np.uint8)
    b[0, 0] = 64                                                                    # BORDER_REFLECT_101: the corner sees itself once
    assert cvlite.gaussian_blur3_u8(b)[0, 0] == (64 * 4 + 8) >> 4 and cvlite.gaussian_blur3_u8(b)[1, 1] == (64 + 8) >> 4


def test_canny_step_ramp_and_hysteresis():
    img = np.zeros((20, 24), np.uint8)
    img[:, 12:] = 200
    e = cvlite.canny_u8(img, 10, 100)
    assert set(np.unique(e)) == {0, 255}
    assert (e[:, 11] == 255).all() and e.sum() == 255 * 20                         # one-pixel line, on the dark side of the tie
    et = cvlite.canny_u8(img.T.copy(), 10, 100)
    assert (et[11, :] == 255).all() and et.sum() == 255 * 20                       # vertical sector: m > above, m >= below
    weak = np.zeros((20, 24), np.uint8)
    weak[:, 12:] = 20                                                               # |dx| = 80: above low, never above high
    assert cvlite.canny_u8(weak, 10, 100).sum() == 0
    mixed = weak.copy()
    mixed[:10, 12:] = 200                                                           # the strong half pulls in the connected weak half
    em = cvlite.canny_u8(mixed, 10, 100)
    assert (em[:7, 11] == 255).all() and (em[13:, 11] == 255).all()                # (the junction rows bend the gradient direction)
    assert cvlite.canny_u8(np.full((9, 9), 50, np.uint8), 10, 100).sum() == 0
    d = np.fromfunction(lambda y, x: (x + y >= 16) * 180, (16, 16)).astype(np.uint8)   # 45-degree edge: diagonal sector
    ed = cvlite.canny_u8(d, 10, 100)
    ys, xs = np.nonzero(ed)
    inner = (ys > 1) & (ys < 14) & (xs > 1) & (xs < 14)
    # the two anti-diagonals next to an exact binary 45-degree step carry EQUAL magnitudes and their suppression partners lie
    # two anti-diagonals away (along the gradient), so both survive -- the rule's documented behaviour on exact ties
    assert inner.sum() >= 10 and set((xs + ys)[inner]) <= {15, 16}


def test_canny_masked_removes_edges_near_the_mask_border():
    img = np.zeros((32, 32), np.uint8)
    img[:, 16:] = 200
    mask = np.zeros((32, 32))
    mask[4:28, 4:28] = 1
    e = cvlite.canny_masked(img, mask)
    assert e[8:24, 15].min() == 255 and e[:8].sum() == 0 and e[24:].sum() == 0     # erosion x4: rows 8..23 survive


def test_normalize_to_uint8():
    a = np.stack([np.linspace(-1, 3, 12).reshape(3, 4), np.full((3, 4), 5.0)])
    u = cvlite.normalize_to_uint8(a, channel_idx=(1, 2))
    assert u.dtype == np.uint8 and u[0].min() == 0 and u[0].max() == 255 and (u[1] == 0).all()


@pytest.mark.gpu
def test_image_to_periodicity_chain_on_a_lattice():
    """search_periodicity_by_feat on the synthetic lattice image (oracle.synthetic_image: known two displacement vectors): the
    gray-only feature map, the Canny-edge variant and the AlexNet-conv1 variant (fixed-seed random filters: the checkpoint is
    not in the reference tree) all recover a displacement pair whose periods match the lattice's to 1.5 feature-map pixels."""
    import torch
    from npp_amd import proposal
    dev = torch.device("cuda:0")
    H = 256
    img, mask = oracle.synthetic_image(H, noise=0.01)
    _, periods, _ = oracle.synthetic_periodicity(H, 1)
    true_p = sorted(float(v) for v in np.ravel(periods[0]))
    im8, m8 = np.uint8(np.clip(img * mask, 0, 1) * 255), np.uint8(mask[..., 0])
    with pytest.raises(ValueError, match="alexnet"):
        proposal.AlexConv1(None, device=dev)
    with pytest.warns(UserWarning, match="RANDOM"):
        conv1 = proposal.AlexConv1(None, device=dev, allow_random=True)
    act, m = proposal.im2act(im8, m8, conv1=conv1)
    assert act.shape == (66, 64, 64) and m.shape == (1, 64, 64) and float((act * (1 - m)).abs().max()) == 0.0
    # conv1 against torch's own convolution of the same normalised, padded image
    x = torch.from_numpy(im8).to(dev).permute(2, 0, 1).float().div(255)
    x = (x - torch.tensor([0.485, 0.456, 0.406], device=dev).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225], device=dev).view(3, 1, 1)
    ref = torch.nn.functional.conv2d(x[None], conv1.w.view(64, 3, 11, 11), conv1.b, stride=4, padding=5)[0]
    assert float((conv1(im8) - ref).abs().max()) < 1e-3
    edge = proposal.act2edge(act[:-1], m)
    assert edge.shape == (2, 64, 64) and float(edge[0].max()) >= 1.0 and float(edge[0].min()) == 0.0
    found = {}
    for name, kw in (("gray", dict(gray_only=True)), ("gray_edges", dict(gray_only=True, edge_searching=True)),
                     ("conv1_edges", dict(conv1=conv1, edge_searching=True))):
        ang, per, sh = proposal.search_periodicity_by_feat(im8, m8, repeat_range=(2, 12, 5), device=dev, **kw)
        assert len(ang) == len(per) == len(sh) >= 1, name
        best = min(per, key=lambda p: abs(sorted(p)[0] - true_p[0]) + abs(sorted(p)[1] - true_p[1]))
        found[name] = sorted(float(v) for v in best)
    for name, p in found.items():
        # the feature map is a quarter of the image and displacements are whole map pixels: one map pixel = 4 image pixels per
        # vector; the true pair must be among the candidates of some repeat-range group
        assert abs(p[0] - true_p[0]) <= 6.0 and abs(p[1] - true_p[1]) <= 6.0, (name, p, true_p)


def test_pseudo_mask_split_follows_the_loader():
    """loaders/loaders.py:34-54 + utils/miscs.py:53-96: centroids = points furthest from the unknown region, pairwise
    >= 0.3 min(H, W) apart; square holes of half-width dist / sqrt(2) / 1.2; known pixels inside them evaluate the fits."""
    from npp_amd import search
    H = 96
    mask = np.ones((H, H))
    mask[30:60, 34:70] = 0                                                          # unknown block
    cents, dist = search.find_mask_centroid(mask)
    assert len(cents) == 3 and all(mask[h, w] == 1 for h, w in cents)
    for i in range(3):
        for j in range(i):
            assert np.hypot(cents[i][0] - cents[j][0], cents[i][1] - cents[j][1]) >= 0.3 * H
    assert dist == sorted(dist, reverse=True)
    pseudo, i_train, i_val = search.pseudo_mask_split(mask, np.ones_like(mask))
    assert pseudo.shape == (H, H, 1) and len(i_val) > 0 and len(i_train) + len(i_val) == int(mask.sum())
    assert all(mask[y, x] == 1 and pseudo[y, x, 0] == 0 for y, x in i_val[::17])
    hw = int(dist[0] / np.sqrt(2) / 1.2)
    h, w = cents[0]
    assert (pseudo[max(h - hw, 0):h + hw, max(w - hw, 0):w + hw, 0] == 0).all()


@pytest.mark.gpu
def test_search_driver_writes_a_loadable_detected_dir(tmp_path):
    """python -m npp_amd.search on a synthetic image directory (the reference's default switches: gray features + Canny edges;
    short candidate fits, fixed-seed random ranking trunks): config.odgt + PNGs that the completion loader reads back, the best
    candidate's periods near the lattice's."""
    import warnings
    from npp_amd import io as nio, search
    H = 256                                     # (the feature map is a quarter of it: 64 x 64, as in the chain test above)
    img, mask = oracle.synthetic_image(H, noise=0.01)
    _, periods, _ = oracle.synthetic_periodicity(H, 1)
    src = tmp_path / "input" / "lattice"
    nio.write_detected_dir(str(src), img, mask, np.ones_like(mask), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]])   # only the four PNGs matter
    out = tmp_path / "detected"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rc = search.main(["--datadir", str(src), "--outdir", str(out), "--N_iters", "40", "--N_rand", "1024", "--search_range", "2", "12", "5",
                          "--topk_detection", "3", "--random-trunks", "--rng_mode", "fast"])
    assert rc == 0
    d = nio.load_npp_completion(str(out / "lattice"), p_topk=2)
    assert len(d["angles"]) == 2 and d["img"].shape == (H, H, 3)
    assert np.abs(d["img"] - img).max() <= 2.0 / 255 and np.array_equal(d["mask"][..., 0] > 0.5, np.asarray(mask).reshape(H, H) > 0.5)   # the images pass through unscaled
    import json
    odgt = json.loads(open(out / "lattice" / "config.odgt").readline())
    assert odgt["epoch"] == 40 and odgt["distances"] == sorted(odgt["distances"]) and len(odgt["selected_shifts"]) <= 3
    true_p = sorted(float(v) for v in np.ravel(periods[0]))
    cand = [sorted(p) for p in odgt["selected_periods"]]
    assert min(max(abs(c[0] - true_p[0]), abs(c[1] - true_p[1])) for c in cand) <= 6.0, (cand, true_p)     # one map pixel = 4 image pixels
    assert all(os.path.exists(p[0]) for k, p in odgt.items() if k.startswith("fpath_reg_img_")) and "fpath_reg_img_0" in odgt
    with pytest.raises(SystemExit, match="exists"):
        search.main(["--datadir", str(src), "--outdir", str(out), "--random-trunks"])


@pytest.mark.gpu
def test_images_searched_together_rank_like_the_serial_loop(tmp_path):
    """run.search_all, the three forms of searching a rank's images (what `python -m npp_amd.run --stack M` does before it fits them
    together): one after the other; side by side on host threads with a stream each (round 5); candidate k of EVERY image in one launch
    sequence (round 6 default: search.main_multi / light.rank_images).  Since the candidate fits and their scores are bit-reproducible
    (round 6), all three write the SAME config.odgt per image -- candidates, order and distances to the last bit."""
    import json
    import warnings
    from npp_amd import io as nio, run
    H, M = 256, 4
    srcs = []
    for i in range(M):
        img, mask = oracle.synthetic_image(H, noise=0.01, seed=i)
        srcs.append(nio.write_detected_dir(str(tmp_path / "input" / f"img{i}"), img, mask, np.ones_like(mask), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]]))
    flags = ["--device", "cuda:0", "--N_iters", "40", "--N_rand", "1024", "--search_range", "2", "12", "5", "--topk_detection", "3", "--random-trunks"]
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw in (("serial", dict(threads=1, together=False)), ("threads", dict(threads=4, together=False)), ("together", dict())):
            det = str(tmp_path / f"det_{tag}")
            assert run.search_all(srcs, det, flags, **kw) == [None] * M
            out[tag] = [json.loads(open(os.path.join(det, f"img{i}", "config.odgt")).readline()) for i in range(M)]
        # a second call finds the directories in place ("file exists"): not an error, nothing is searched again
        assert run.search_all(srcs, str(tmp_path / "det_together"), flags) == [None] * M
        assert run.search_all(srcs, str(tmp_path / "det_threads"), flags, threads=4, together=False) == [None] * M
    for tag in ("threads", "together"):
        for a, b in zip(out["serial"], out[tag]):
            assert a["selected_angles"] == b["selected_angles"] and a["selected_periods"] == b["selected_periods"]
            assert a["carry_adaptive_latents"] is True and b["carry_adaptive_latents"] is True      # the reference's chained candidates in every form
            assert a["distances"] == b["distances"], (tag, a["distances"], b["distances"])
    # a failing image is reported in its slot, the others are searched
    for kw in (dict(threads=2, together=False), dict()):
        errs = run.search_all([srcs[0], str(tmp_path / "input" / "missing")], str(tmp_path / f"det_err{len(kw)}"), flags, **kw)
        assert errs[0] is None and errs[1] is not None


def test_search_flags_keep_the_reference_store_false_semantics():
    """options/arg_config.py:122-126: --gray_only / --edge_searching are store_false switches, so the reference's default run is
    gray features + Canny edges and passing --gray_only turns the AlexNet features ON."""
    from npp_amd import search
    a = search.parse(["--datadir", "x"])
    assert a.gray_only is True and a.edge_searching is True and tuple(a.search_range) == (1, 10, 1)
    assert a.N_iters == 300 and a.N_rand == 2048 and a.netdepth == 4 and a.perceptual_weight == 30 and a.topk_detection == 10
    b = search.parse(["--datadir", "x", "--gray_only", "--edge_searching"])
    assert b.gray_only is False and b.edge_searching is False
