"""Stacked launches (M images of one shape per launch, npp_amd.stack.StackedFit) against each image's stand-alone fit: the
per-image arithmetic is the single-image path's, launch by launch, so after 20 iterations every image's parameters must equal
those of its own CompletionFit run alone -- exactly where the path has no order-dependent reduction left, to a small tolerance
where the trunk launches pick another tile shape for the larger batch (different summation order over channels)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import npp_amd
    npp_amd.lib()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _fits(dev, M, H, K, ksplit, **kw):
    from npp_amd.fit import CompletionFit
    out = []
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    for i in range(M):
        img, mask = oracle.synthetic_image(H, seed=i)                      # another noise field per image
        a = np.asarray(angles, np.float64) + 0.3 * i                       # ... and slightly different periodicity proposals
        out.append(CompletionFit(img, mask, a, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=i), device=dev, N_rand=4096,
                                 shifts=shifts, seed=10 + i, ksplit=ksplit, **kw))
    return out


@pytest.mark.parametrize("M,K", [(2, 1), (4, 3), (3, 1)])
def test_stacked_fit_equals_the_stand_alone_fits(dev, M, K):
    """20 iterations of M stacked images vs each image alone (same split-K count, so the weight gradient sums in the same order).
    M = 3 takes the launch order without XCD ownership (8 / M is not an integer)."""
    from npp_amd.stack import StackedFit
    H, iters = 256, 20
    probe = StackedFit(_fits(dev, M, H, K, None))
    ks = probe.ksplit
    del probe
    alone = _fits(dev, M, H, K, ks)
    sources = []
    for f in alone:
        src = []
        for _ in range(iters):
            f.step_full()
            src.append(f.last_draw["source"] if f.last_draw["k"] > 0 else None)
        sources.append(src)
    st = StackedFit(_fits(dev, M, H, K, ks), ksplit=ks)
    got = [[] for _ in range(M)]
    for _ in range(iters):
        st.step_full()
        for i in range(M):
            got[i].append(st.last_sources[i])
    torch.cuda.synchronize()
    assert got == sources                                                  # every image drew its own stream
    assert any("same" in s for s in sources), "the LPIPS branch was not exercised"
    for i in range(M):
        a, b = alone[i].net, st.fits[i].net
        assert (a.opt_step, a.global_step) == (b.opt_step, b.global_step)
        pa, pb = a.params.cpu().numpy(), b.params.cpu().numpy()
        e = rel_l2(pb, pa)
        print(f"M={M} image {i}: params rel-L2 {e:.2e}, max |d| {np.abs(pa - pb).max():.2e}, latents |d| "
              f"{float((a.latents - b.latents).abs().max()):.2e}")
        assert e < 1e-3, (i, e)
        np.testing.assert_allclose(b.latents.cpu().numpy(), a.latents.cpu().numpy(), atol=2e-4)
        la, lb = alone[i].percepLoss, st.fits[i].percepLoss
        assert la.lat_step == lb.lat_step
        for ta, tb in zip(la.latents, lb.latents):
            np.testing.assert_allclose(tb.cpu().numpy(), ta.cpu().numpy(), atol=2e-4)
        assert abs(alone[i].psnr("known") - st.fits[i].psnr("known")) < 0.05


def test_lpips_branch_of_several_same_images_in_one_trunk_pass(dev):
    """Iterations on which several images of a stack draw 'same' run ONE VGG16 pass for all of them (LPIPS.fused_groups: heads per
    image on its own latents) instead of one branch per image: same parameters and LPIPS latents as the per-image branches
    (batch_lpips = False) to the tolerance of another trunk batch size, and as the stand-alone fits."""
    from npp_amd.stack import StackedFit
    H, K, M, iters = 256, 1, 4, 24
    probe = StackedFit(_fits(dev, M, H, K, None))
    ks = probe.ksplit
    del probe
    runs = {}
    for together in (True, False):
        st = StackedFit(_fits(dev, M, H, K, ks), ksplit=ks)
        st.batch_lpips = together
        shared = 0
        for _ in range(iters):
            st.step_full()
            shared += sum(1 for s_ in st.last_sources if s_ == "same") >= 2
        torch.cuda.synchronize()
        assert shared >= 2, "no iteration with two 'same' images: the shared pass was not exercised"
        assert (st._lp_in is not None) == together
        runs[together] = st
    alone = _fits(dev, M, H, K, ks)
    for f in alone:
        for _ in range(iters):
            f.step_full()
    torch.cuda.synchronize()
    for i in range(M):
        a, b, c = runs[True].fits[i], runs[False].fits[i], alone[i]
        assert a.percepLoss.lat_step == b.percepLoss.lat_step == c.percepLoss.lat_step > 0
        for other in (b, c):
            assert rel_l2(a.net.params.cpu().numpy(), other.net.params.cpu().numpy()) < 1e-3
            for ta, tb in zip(a.percepLoss.latents, other.percepLoss.latents):
                np.testing.assert_allclose(ta.cpu().numpy(), tb.cpu().numpy(), atol=2e-4)
            assert abs(float(a.last_patch_loss) - float(other.last_patch_loss)) <= 5e-3 * abs(float(other.last_patch_loss)) + 1e-6


def test_stacked_mlp_half_is_bit_exact(dev):
    """With the patch-loss weights at zero the patch rows' gradient is an exact zero and what is left -- fused forward, adaptive
    pixel loss, backward chain, grouped weight gradient, Adam + re-pack -- is the same arithmetic in the same order: the stacked
    parameters equal the stand-alone ones bit for bit (the pixel loss sums its block partials by atomics: compared after ONE step
    exactly and after ten to round-off)."""
    from npp_amd.stack import StackedFit
    M, H, K = 2, 256, 3
    kw = dict(contextual_weight=0.0, perceptual_weight=0.0)
    probe = StackedFit(_fits(dev, M, H, K, None, **kw))
    ks = probe.ksplit
    del probe
    alone = _fits(dev, M, H, K, ks, **kw)
    st = StackedFit(_fits(dev, M, H, K, ks, **kw), ksplit=ks)
    for it in range(10):
        for f in alone:
            f.step_full()
        st.step_full()
        torch.cuda.synchronize()
        for i in range(M):
            pa, pb = alone[i].net.params, st.fits[i].net.params
            if it == 0:
                assert torch.equal(pa, pb), (i, float((pa - pb).abs().max()))
                assert torch.equal(alone[i].net.wf, st.fits[i].net.wf) and torch.equal(alone[i].net.wb, st.fits[i].net.wb)
    for i in range(M):
        e = rel_l2(st.fits[i].net.params.cpu().numpy(), alone[i].net.params.cpu().numpy())
        print(f"image {i}: rel-L2 after 10 steps {e:.2e}")
        assert e < 1e-6


def test_stack_with_an_image_sitting_an_iteration_out(dev):
    """An image whose sampler finds no valid real patch (train.py:160-161) skips the iteration: no launch touches its state, its
    Adam step count and LR clock stay, the other images step."""
    from npp_amd.stack import StackedFit
    M, H, K = 2, 256, 1
    st = StackedFit(_fits(dev, M, H, K, None))
    st.step_full()
    b = st.sample()
    p1 = st.fits[1].net.params.clone()
    steps = [f.net.opt_step for f in st.fits]
    b[1] = None
    assert st.step_from(b) == 1
    torch.cuda.synchronize()
    assert torch.equal(st.fits[1].net.params, p1) and st.fits[1].net.opt_step == steps[1]
    assert st.fits[0].net.opt_step == steps[0] + 1
    assert st.step_full() == 2 and bool(torch.isfinite(st.params).all())


def test_stack_re_forms_across_a_patch_size_decay(dev):
    """train.py:137-141 halves the patch size (and doubles the patch count) every patch_size_decay iterations: another batch shape.
    StackedFit.shape_change_due() / restacked() re-form the stack around it; every image still equals its stand-alone fit."""
    from npp_amd.stack import StackedFit
    H, K, M, iters = 256, 1, 2, 16
    probe = StackedFit(_fits(dev, M, H, K, None, patch_size_decay=10))
    ks = probe.ksplit
    del probe
    alone = _fits(dev, M, H, K, ks, patch_size_decay=10)
    for f in alone:
        for _ in range(iters):
            f.step_full()
    st = StackedFit(_fits(dev, M, H, K, ks, patch_size_decay=10), ksplit=ks)
    P0, n_restack = st.P, 0
    for _ in range(iters):
        if st.shape_change_due():
            st = st.restacked()
            n_restack += 1
        st.step_full()
    torch.cuda.synchronize()
    assert n_restack == 1 and st.P == P0 // 2 and st.n_p == 4
    for i in range(M):
        a, b = alone[i].net, st.fits[i].net
        assert (a.opt_step, alone[i].patch_size, alone[i].patch_num) == (b.opt_step, st.fits[i].patch_size, st.fits[i].patch_num)
        # (the re-formed stack picks the split-K of ITS batch shape: weight gradients sum in another order from there on)
        assert rel_l2(b.params.cpu().numpy(), a.params.cpu().numpy()) < 2e-3
        assert abs(alone[i].psnr("known") - st.fits[i].psnr("known")) < 0.1


def test_segmentation_fits_stack_too(dev):
    """The segmentation loop (NPP_segmentation/train.py:148-286) is the completion loop on other inputs: contextual weight 0.005, no
    LPIPS, a learning rate that never decays -- two such fits in one launch sequence equal their stand-alone runs."""
    from npp_amd.stack import StackedFit
    H, K, M, iters = 256, 1, 2, 12
    kw = dict(task="segmentation", contextual_weight=0.005, use_perceptual_loss=False)
    probe = StackedFit(_fits(dev, M, H, K, None, **kw))
    ks = probe.ksplit
    del probe
    alone = _fits(dev, M, H, K, ks, **kw)
    for f in alone:
        for _ in range(iters):
            f.step_full()
    st = StackedFit(_fits(dev, M, H, K, ks, **kw), ksplit=ks)
    for _ in range(iters):
        st.step_full()
    torch.cuda.synchronize()
    for i in range(M):
        a, b = alone[i].net, st.fits[i].net
        assert a.lr_clock is False and b.lr_clock is False and a.lr == b.lr and a.opt_step == b.opt_step
        assert rel_l2(b.params.cpu().numpy(), a.params.cpu().numpy()) < 1e-3


def _remap_fits(dev, M, H, K, ksplit, lpips=False):
    """M remapping fits (NPP_remapping/train.py:158-300): a box-blurred band per image (another band each), clear mask = the rest."""
    from npp_amd.fit import CompletionFit
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    out = []
    for i in range(M):
        clean, _ = oracle.synthetic_image(H, noise=0.01, seed=i)
        k = 9
        pad = np.pad(clean, ((k // 2, k // 2), (k // 2, k // 2), (0, 0)), mode="edge")
        box = sum(pad[dy:dy + H, dx:dx + H] for dy in range(k) for dx in range(k)) / (k * k)
        band = slice(64 + 32 * i, 128 + 32 * i)
        blurred = clean.copy()
        blurred[band] = box[band]
        clear = np.ones((H, H, 1), np.float32)
        clear[band] = 0
        out.append(CompletionFit(blurred, np.ones((H, H, 1), np.float32), angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=i),
                                 device=dev, N_rand=4096, shifts=shifts, seed=20 + i, ksplit=ksplit, task="remapping", clear_mask=clear,
                                 contextual_weight=0.01, use_perceptual_loss=lpips))
    return out


@pytest.mark.parametrize("lpips", [False, True])
def test_remapping_fits_stack_too(dev, lpips):
    """The remapping loop adds two things to the completion loop: per-pixel loss weights (blurry pixels 0.3: they ride in the stacked
    pixel-loss launch, one mask row per image) and the Gram-matrix style term with ITS OWN adaptive latents per image (run per image
    beside the contextual chain, its gradient joined in the stacked backward launch).  Two such fits in one launch sequence end where
    their stand-alone runs end: network parameters, style latents, pixel-weight masks in use.  lpips: with the LPIPS term switched on
    as well (not the task's default): 'same' iterations then add BOTH gradients of an image in its side-stream branch."""
    from npp_amd.stack import StackedFit
    # (8 iterations: Adam's first steps are sign-like, so the rounding differences of the larger trunk batch -- other tile shapes, other
    #  summation order -- grow along the trajectory: 2e-4 after one step, 6e-4 .. 7e-4 after 8, 1e-3 after 12; a stack of ONE image starts
    #  at 3e-7 and reaches 6e-4 after 8 just the same)
    H, K, M, iters = 256, 1, 2, 8
    probe = StackedFit(_remap_fits(dev, M, H, K, None, lpips))
    ks = probe.ksplit
    assert "pmask" in probe._sets[0] and all(f.style is not None for f in probe.fits)
    del probe
    alone = _remap_fits(dev, M, H, K, ks, lpips)
    for f in alone:
        for _ in range(iters):
            f.step_full()
    st = StackedFit(_remap_fits(dev, M, H, K, ks, lpips), ksplit=ks)
    lat0 = [[l.clone() for l in f.style.latents] for f in st.fits]
    for _ in range(iters):
        assert st.step_full() == M
    torch.cuda.synchronize()
    for s_ in st._sets:                                                    # blurry pixel rows were drawn and weighted
        pm = s_["pmask"].cpu().numpy()
        assert ((pm == 0.0) | (pm == 1.0)).all() and (pm == 0.0).any() and (pm == 1.0).any()
    for i in range(M):
        a, b = alone[i], st.fits[i]
        assert a.net.opt_step == b.net.opt_step == iters and a.style.lat_step == b.style.lat_step == iters
        assert rel_l2(b.net.params.cpu().numpy(), a.net.params.cpu().numpy()) < 1e-3
        for la, lb, l0 in zip(a.style.latents, b.style.latents, lat0[i]):
            assert (lb - l0).abs().max() > 0                               # trained ...
            assert rel_l2(lb.cpu().numpy(), la.cpu().numpy()) < 1e-3       # ... to the stand-alone fit's values
        assert abs(float(a.last_patch_loss) - float(b.last_patch_loss)) <= 5e-3 * abs(float(a.last_patch_loss)) + 1e-6
        if lpips:
            assert a.percepLoss.lat_step == b.percepLoss.lat_step
            for ta, tb in zip(a.percepLoss.latents, b.percepLoss.latents):
                np.testing.assert_allclose(tb.cpu().numpy(), ta.cpu().numpy(), atol=2e-4)


def test_directory_driver_fits_several_images_in_one_launch_sequence(dev, tmp_path):
    """npp_amd.train.main_stacked (what `python -m npp_amd.run --stack M` calls): three detected/ directories -- two of one patch size,
    one of another -- fitted as one stack of two and one plain loop, test sets written per image, and the stacked images end where
    their own `train.main` runs end."""
    from npp_amd import io as nio, train
    H, K = 256, 3
    a, p, s = oracle.synthetic_periodicity(H, K)
    dirs = []
    for i, scale in enumerate((1.0, 1.0, 2.0)):                            # the third image: twice the period -> another patch size
        img, mask = oracle.synthetic_image(H, seed=i)
        pp = np.asarray(p, np.float64) * scale
        dirs.append(nio.write_detected_dir(str(tmp_path / "detected" / f"img{i}"), img, mask, np.ones_like(mask), a, pp, s))
    flags = ["--p_topk", "3", "--N_iters", "41", "--i_testset", "40", "--i_print", "40", "--rng_mode", "fast", "--random-trunks",
             "--netwidth", "256", "--N_rand", "4096"]
    fits = train.main_stacked([["--datadir", d, "--basedir", str(tmp_path / "stacked")] + flags for d in dirs], max_stack=8)
    assert train.main_stacked.last_error is None and all(f is not None for f in fits)
    assert fits[0].patch_size == fits[1].patch_size != fits[2].patch_size
    for i in range(3):
        out = tmp_path / "stacked" / "completion_top3" / f"img{i}" / "testset_000040"
        assert out.is_dir() and len(list(out.iterdir())) == 6
    single = train.main(["--datadir", dirs[1], "--basedir", str(tmp_path / "single"), "--prefetch", "0"] + flags)
    assert abs(single.psnr() - fits[1].psnr()) < 0.3 and fits[1].psnr() > 20.0
    # a second call finds every output directory in place and fits nothing (train.py:42-44)
    again = train.main_stacked([["--datadir", d, "--basedir", str(tmp_path / "stacked")] + flags for d in dirs])
    assert again == [None, None, None]


def test_directory_driver_isolates_a_failing_image_and_releases_finished_groups(dev, tmp_path):
    """(high) one image with a missing detection fails alone -- the others are fitted, nothing is left behind under its
    name, a re-run fits only what is missing; (medium) what main_stacked returns are FitResult summaries, the fits' device memory is
    released group by group."""
    import torch
    from npp_amd import io as nio, train
    H, K = 256, 3
    a, p, s = oracle.synthetic_periodicity(H, K)
    dirs = []
    for i in range(2):
        img, mask = oracle.synthetic_image(H, seed=40 + i)
        dirs.append(nio.write_detected_dir(str(tmp_path / "detected" / f"img{i}"), img, mask, np.ones_like(mask), a, p, s))
    dirs.insert(1, str(tmp_path / "detected" / "img_missing"))
    flags = ["--p_topk", "3", "--N_iters", "21", "--i_testset", "20", "--i_print", "20", "--rng_mode", "fast", "--random-trunks",
             "--netwidth", "256", "--N_rand", "4096"]
    res = train.main_stacked([["--datadir", d, "--basedir", str(tmp_path / "out")] + flags for d in dirs])
    assert res[1] is None and train.main_stacked.errors[1] is not None and train.main_stacked.errors[0] is None
    assert isinstance(res[0], train.FitResult) and isinstance(res[2], train.FitResult) and res[0].opt_step == 20
    assert not (tmp_path / "out" / "completion_top3" / "img_missing").exists()
    assert (tmp_path / "out" / "completion_top3" / "img0" / "testset_000020").is_dir()
    free0 = torch.cuda.mem_get_info()[0]
    again = train.main_stacked([["--datadir", d, "--basedir", str(tmp_path / "out")] + flags for d in dirs])
    assert again == [None, None, None] and train.main_stacked.errors[0] is None and train.main_stacked.errors[1] is not None
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)               # nothing of the finished groups is still allocated


def test_directory_driver_stacks_remapping_images(dev, tmp_path):
    """train.main_stacked on two remapping runs (--task remapping: blur detection -> clear mask -> per-pixel loss weights, style term):
    one stack of two, test sets written per image, style latents trained."""
    from npp_amd import io as nio, train
    H, K = 256, 3
    a, p, s = oracle.synthetic_periodicity(H, K)
    dirs = []
    for i in range(2):
        img, mask = oracle.synthetic_image(H, seed=30 + i)
        dirs.append(nio.write_detected_dir(str(tmp_path / "detected" / f"img{i}"), img, np.ones_like(mask), np.ones_like(mask), a, p, s))
    flags = ["--task", "remapping", "--p_topk", "3", "--N_iters", "21", "--i_testset", "20", "--i_print", "20", "--rng_mode", "fast", "--random-trunks",
             "--netwidth", "256", "--N_rand", "4096"]
    fits = train.main_stacked([["--datadir", d, "--basedir", str(tmp_path / "out")] + flags for d in dirs])
    assert train.main_stacked.last_error is None and all(f is not None and f.has_style and f.has_pixel_mask for f in fits)
    assert all(f.style_lat_step == 20 and f.opt_step == 20 for f in fits)
    for i in range(2):
        out = tmp_path / "out" / "remapping_top3" / f"img{i}" / "testset_000020"
        assert out.is_dir() and len(list(out.iterdir())) >= 4

