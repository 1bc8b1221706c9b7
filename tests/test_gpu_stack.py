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
