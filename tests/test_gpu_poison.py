"""Nothing on the path may depend on what freshly allocated memory happens to hold.

Round 6 found a score that depended on which tests had run before: padding entries of a tiled matrix were masked by a multiplication
with 0, the never-written bytes behind them were NaN patterns left by an earlier allocation, and `fmaxf` silently dropped the
NaN rows.  A fresh process hides that class of bug -- the allocator hands out zeroed pages.  These tests make every `torch.empty`
/ `empty_like` / `new_empty` of a floating or byte dtype return memory filled with 0xFF bytes (NaN for fp32 / bf16 / fp16, -1 for
fp8-style byte codes) while a fit is built and stepped, clear the per-stream workspace caches first so that they are re-made under the
poison too, and ask for the bits of the un-poisoned twin.  Integer index buffers are left alone: an index read before it is written
would fault the GPU rather than fail a test."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402

pytestmark = pytest.mark.gpu
_POISON = (torch.float32, torch.float16, torch.bfloat16, torch.uint8)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import npp_amd
    npp_amd.lib()
    return torch.device("cuda:0")


@contextlib.contextmanager
def poisoned_empty():
    from npp_amd import ops
    real_empty, real_like = torch.empty, torch.empty_like
    real_new = torch.Tensor.new_empty

    def spoil(t):
        if t.is_cuda and t.dtype in _POISON and t.numel():
            t.view(torch.uint8).fill_(255) if t.is_contiguous() else None
        return t

    def empty(*a, **k):
        return spoil(real_empty(*a, **k))

    def empty_like(*a, **k):
        return spoil(real_like(*a, **k))

    def new_empty(self, *a, **k):
        return spoil(real_new(self, *a, **k))
    saved = {n: dict(getattr(ops, n)) for n in ("_cx_ws",)}
    ops._cx_ws.clear()                                     # (torch.empty workspaces: re-made under the poison)
    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty
    try:
        yield
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = real_empty, real_like, real_new
        ops._cx_ws.clear()
        ops._cx_ws.update(saved["_cx_ws"])


def _fit_state(dev, task="completion", n_steps=24):
    from npp_amd.fit import CompletionFit
    H, K = 256, 3
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    kw = {}
    if task == "remapping":
        clear = np.ones((H, H, 1), np.float32)
        clear[H // 3:H // 2] = 0.0
        mask = np.ones((H, H, 1), np.float32)
        kw = dict(task="remapping", clear_mask=clear, contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False)
    fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), device=dev, N_rand=4096,
                        shifts=shifts, seed=4, **kw)
    seen, losses = set(), []
    for _ in range(n_steps):
        if fit.step_full():
            seen.add(fit.last_source)
            losses.append(fit.last_patch_loss.clone())
    pred = fit.render_image()
    torch.cuda.synchronize()
    state = [fit.net.params.clone(), fit.net.m.clone(), fit.net.v.clone(), fit.net.latents.clone(), pred.clone(), torch.stack(losses)]
    if task == "completion":
        state.append(fit.percepLoss._lat.clone())
    else:
        state += [l.clone() for l in fit.style.latents]
    return seen, state


@pytest.mark.parametrize("task", ["completion", "remapping"])
def test_complete_iterations_do_not_read_unwritten_memory(dev, task):
    """24 complete iterations (all three patch sources; LPIPS branch or style branch) + the full-grid render: the same bits whether
    fresh allocations hold zeros or NaN patterns."""
    seen, clean = _fit_state(dev, task)
    assert seen == {"val", "train", "same"}
    with poisoned_empty():
        seen_p, dirty = _fit_state(dev, task)
    assert seen_p == seen
    for k, (a, b) in enumerate(zip(clean, dirty)):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
        if k == 5:
            # the REPORTED patch loss of an iteration is one word that the contextual core (main stream) and the LPIPS / style branch
            # (side stream) each add their term to, in arrival order: a + b against b + a onto the word's previous content can differ in
            # the last bit (seen once in ~12 runs, 7e-9 on 7e-2).  No gradient is computed from it.
            torch.testing.assert_close(b, a, rtol=1e-6, atol=0)
            continue
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
    # (the remapping leg is exact since the Gram matrices' split contraction and the style term's loss word are summed in a fixed order,
    #  npp_gram_fwd_det / npp_robust_elem: before, two CLEAN runs differed in the 8th digit of the second iteration's loss)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_candidate_ranking_does_not_read_unwritten_memory(dev, precision):
    """The proposal ranking (fused fp32 candidate fits, ordered-split weight gradients, whole-crop score: plain LPIPS + the value-only
    contextual core) under poisoned allocations: the same distances to the last bit."""
    from npp_amd.light import ProposalRanker
    H = 256
    img, _ = oracle.synthetic_image(H, noise=0.01)
    angles, periods, shifts = oracle.synthetic_periodicity(H, 3)
    pseudo = np.ones((H, H), np.float32)
    pseudo[24:232, 20:236] = 0                               # (a val crop of 208 x 216 pixels: 52 x 54 = 2808 > 2048 feature positions)
    i_train, i_val = np.stack(np.nonzero(pseudo), 1), np.stack(np.nonzero(1 - pseudo), 1)
    cands = [(angles[i], periods[i], shifts[i]) for i in range(3)]

    def run():
        rk = ProposalRanker(img, i_train, i_val, device=dev, N_iters=40, N_rand=2048, carry_latents=True, precision=precision)
        d, order, details = rk.rank(cands, topk=3)
        return np.asarray(d), list(order), details
    d0, o0, det0 = run()
    with poisoned_empty():
        d1, o1, det1 = run()
    assert np.all(np.isfinite(d0)) and o0 == o1 and d0.tolist() == d1.tolist() and det0 == det1


def test_stacked_images_do_not_read_unwritten_memory(dev):
    """Three images in one launch sequence (stack.StackedFit: job tables, per-image slabs, grouped contextual core, batched LPIPS
    branch), 20 stacked iterations: the same parameter bits under poisoned allocations."""
    from npp_amd.fit import CompletionFit
    from npp_amd.stack import StackedFit
    H, K, M = 256, 3, 3
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)

    def run():
        fits = []
        for i in range(M):
            img, mask = oracle.synthetic_image(H, seed=i)
            fits.append(CompletionFit(img, mask, np.asarray(angles, np.float64) + 0.3 * i, periods, oracle.SEED0_FREQS,
                                      oracle.init_params(K, seed=i), device=dev, N_rand=4096, shifts=shifts, seed=10 + i))
        st = StackedFit(fits)
        seen = set()
        for _ in range(20):
            st.step_full()
            seen |= {s for s in st.last_sources if s}
        torch.cuda.synchronize()
        return seen, [f.net.params.clone() for f in st.fits] + [f.net.latents.clone() for f in st.fits] + [f.percepLoss._lat.clone() for f in st.fits]
    seen, clean = run()
    assert "same" in seen
    with poisoned_empty():
        seen_p, dirty = run()
    assert seen_p == seen
    for a, b in zip(clean, dirty):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.parametrize("stacked", [False, True])
def test_remapping_with_lpips_does_not_read_unwritten_memory(dev, stacked):
    """The remapping loop with the LPIPS term switched on as well (style term + LPIPS branch + contextual chain in one iteration), alone
    and as a stack of two.  Found by running the suite's files in another order: the stacked / stand-alone comparison of
    tests/test_gpu_stack.py moved from 7e-4 to 2.5e-3 -- not unwritten memory but a run-to-run drift of the LAST stacked image's style
    latent gradients while its style term ran on the side stream beside the contextual chain; StackedFit now runs the style terms on
    the main stream (style_side_stream = False) and two clean runs, and the poisoned one, agree to the bit."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_stack import _remap_fits
    from npp_amd.stack import StackedFit

    def run():
        fits = _remap_fits(dev, 2, 256, 1, 4, True)
        if stacked:
            st = StackedFit(fits, ksplit=4)
            for _ in range(8):
                st.step_full()
            fits = st.fits
        else:
            for f in fits:
                for _ in range(8):
                    f.step_full()
        torch.cuda.synchronize()
        out = []
        for f in fits:
            out += [f.net.params.clone(), f.net.latents.clone(), f.percepLoss._lat.clone()] + [l.clone() for l in f.style.latents]
        return out
    clean = run()
    with poisoned_empty():
        dirty = run()
    for k, (a, b) in enumerate(zip(clean, dirty)):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), k
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
