"""The native MT19937 stream of libnpp_hip.so (csrc/npp_host_rng.hip) against numpy.random.RandomState, bit for bit: the
reference's sampler draws from NumPy's global legacy generator (models/sampler.py:260,324; NPP_completion/train.py:172)."""
import time

import numpy as np
import pytest


@pytest.fixture(scope="module")
def NRS():
    from npp_amd.host_rng import NativeRandomState
    return NativeRandomState


@pytest.mark.parametrize("seed", [0, 1, 12345, 2 ** 32 - 1])
def test_state_and_stream_match_numpy(NRS, seed):
    a, b = np.random.RandomState(seed), NRS(seed)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]                 # init_genrand seeding
    for n, size in ((10, 10), (1000, 7), (245760, 8192), (17, 1), (1, 1), (100000, 2), (2 ** 20 + 3, 5)):
        u1, u2 = a.uniform(0, 1), b.uniform(0, 1)
        assert u1 == u2                                                    # the 53-bit double
        np.testing.assert_array_equal(a.choice(n, size=[size], replace=False), b.choice(n, size=[size], replace=False))
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]                 # same number of words consumed


@pytest.mark.parametrize("env", [{"NPP_RNG_AVX2": "0"}, {"NPP_RNG_THREADS": "0"}, {"NPP_RNG_AVX2": "0", "NPP_RNG_THREADS": "0"}, {}])
def test_every_shuffle_form_is_the_numpy_stream(env):
    """The shuffle has four forms (AVX2 runs of 64 words or the scalar groups of 8; helper-thread generation for >= 200 000 elements
    or one thread), chosen once per process: each in its own interpreter against numpy.random.RandomState, including the state
    afterwards (same number of generator words consumed) and populations around the block / run / threshold sizes."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from npp_amd.host_rng import NativeRandomState\n"
        "a, b = np.random.RandomState(3), NativeRandomState(3)\n"
        "for n in (1, 2, 63, 64, 65, 2047, 2048, 2049, 4097, 199999, 200000, 262144, 1048576 + 17):\n"
        "    assert np.array_equal(a.choice(n, size=[min(n, 300)], replace=False), b.choice(n, size=[min(n, 300)], replace=False)), n\n"
        "    assert a.uniform(0, 1) == b.uniform(0, 1)\n"
        "sa, sb = a.get_state(), b.get_state()\n"
        "assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_state_exchange_with_numpy(NRS):
    a = np.random.RandomState(7)
    a.uniform(size=1000)
    b = NRS(0)
    b.set_state(a.get_state())
    np.testing.assert_array_equal(a.choice(5000, size=[64], replace=False), b.choice(5000, size=[64], replace=False))
    a2 = np.random.RandomState(0)
    st = b.get_state()
    a2.set_state(("MT19937", st[1], st[2], 0, 0.0))
    assert a2.uniform(2.0, 5.0) == b.uniform(2.0, 5.0)


def test_same_draw_timing_report(NRS):
    """Same stream as NumPy on the loop's real draw (245 760 choose 8192); prints the per-draw times and the two-thread
    ratio (ctypes releases the GIL; NumPy's legacy shuffle does not).  Timing is reported, not asserted: shared CI hosts
    make wall-clock ratios meaningless (a known GIL-free hashlib call shows 1.3x .. 2x for two threads here)."""
    import threading
    a, b = np.random.RandomState(3), NRS(3)
    n, size, reps = 245760, 8192, 12
    t0 = time.perf_counter()
    for _ in range(reps):
        ra = a.choice(n, size=[size], replace=False)
    t_np = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(reps):
        rb = b.choice(n, size=[size], replace=False)
    t_nat = time.perf_counter() - t0
    np.testing.assert_array_equal(ra, rb)
    rs = [NRS(i) for i in range(2)]

    def work(r):
        for _ in range(reps):
            r.choice(n, size=[size], replace=False)
    best = 1e9
    for attempt in range(4):                       # wall-clock ratio on a shared machine: take the best of a few attempts
        t0 = time.perf_counter()
        work(rs[0])
        t1 = time.perf_counter() - t0
        th = [threading.Thread(target=work, args=(r,)) for r in rs]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        t2 = time.perf_counter() - t0
        best = min(best, t2 / t1)
        if best < 1.6:
            break
    print(f"numpy {t_np / reps * 1e3:.2f} ms, native {t_nat / reps * 1e3:.2f} ms per draw; 2 threads / 1 thread = {best:.2f}x")
    assert best > 0


def _samplers(H, P, n, seed, shifts=None, mask_fn=None):
    import torch
    import oracle
    from npp_amd.sampler import GridPatchSampler
    from npp_amd.host_rng import NativeRandomState
    img, mask = oracle.synthetic_image(H)
    if mask_fn is not None:
        mask = mask_fn(mask)
    _, _, sh = oracle.synthetic_periodicity(H, 1)
    sh = shifts if shifts is not None else sh
    i_train = np.stack(np.nonzero(mask[..., 0]), 1)
    i_val = np.stack(np.nonzero(1 - mask[..., 0]), 1)
    mk = lambda rng: GridPatchSampler(torch.from_numpy(img * mask)[None], torch.from_numpy(mask)[None], n, P, H, H,   # noqa: E731
                                      i_train, i_val, sh, rng=rng)
    return mk(np.random.RandomState(seed)), mk(NativeRandomState(seed))


@pytest.mark.parametrize("H,P,n,seed", [(256, 64, 2, 0), (256, 32, 4, 3), (512, 96, 2, 1)])
def test_native_sampler_draw_equals_python_draw(H, P, n, seed):
    """npp_sampler_draw (csrc/npp_host_rng.hip) against GridPatchSampler.draw's NumPy restatement of models/sampler.py:242-354
    driven by numpy.random.RandomState: same patch source, centres, k, lattice candidates (as float64, bit for bit), 1/d weights
    and the same generator state afterwards, call after call."""
    a, b = _samplers(H, P, n, seed)
    assert a._native is None and b._native is not None
    seen = set()
    for it in range(150):
        da, db = a.draw(3, 0.3), b.draw(3, 0.3)
        seen.add((da["source"], da["k"]))
        assert (da["source"], da["k"]) == (db["source"], db["k"]), it
        np.testing.assert_array_equal(da["cen"], db["cen"])
        if da["real_cen"] is None:
            assert db["real_cen"] is None
        else:
            np.testing.assert_array_equal(da["real_cen"], db["real_cen"])            # float64 candidate positions, exact
            np.testing.assert_array_equal(da["weights"], db["weights"])
        if it % 25 == 0:                                                             # patch-size decay path: reset + re-filter
            for s_ in (a, b):
                s_.reset_patchsize(None, None, P if it % 50 else P // 2, n if it % 50 else 2 * n)
                s_.reset_pool(*_pools(H))
    assert {"val", "train", "same"} <= {s_ for s_, _ in seen}
    assert a.rng.get_state()[2] == b.rng.get_state()[2] and np.array_equal(a.rng.get_state()[1], b.rng.get_state()[1])


def _pools(H):
    import oracle
    _, mask = oracle.synthetic_image(H)
    return np.stack(np.nonzero(mask[..., 0]), 1), np.stack(np.nonzero(1 - mask[..., 0]), 1)


def test_native_sampler_draw_skip_and_short_k():
    """A lattice that leaves the image after one step: few or no valid candidates -> k < topk and the k == 0 'skip' result
    (train.py:160-161), identical in both implementations."""
    sh = [[(200.0, 10.0), (-15.0, 180.0)]]
    a, b = _samplers(256, 64, 2, 5, shifts=sh)
    ks = set()
    for it in range(60):
        da, db = a.draw(3, 0.3), b.draw(3, 0.3)
        assert (da["source"], da["k"]) == (db["source"], db["k"])
        ks.add(da["k"])
        if da["real_cen"] is not None:
            np.testing.assert_array_equal(da["real_cen"], db["real_cen"])
            np.testing.assert_array_equal(da["weights"], db["weights"])
    assert 0 in ks


def test_random_patch_mode_matches_reference_rule():
    """no_reg_sampling=True (models/sampler.py:66-85,219-228): real patches are N * topk windows drawn without replacement
    from the stride-P/10 windows that contain no unknown pixel; no weights; k = topk."""
    import torch
    import oracle
    from npp_amd.sampler import GridPatchSampler
    H, P, n = 256, 64, 2
    img, mask = oracle.synthetic_image(H)
    _, _, sh = oracle.synthetic_periodicity(H, 1)
    i_train, i_val = _pools(H)
    S = GridPatchSampler(torch.from_numpy(img * mask)[None], torch.from_numpy(mask)[None], n, P, H, H, i_train, i_val, sh,
                         no_reg_sampling=True, rng=np.random.RandomState(0))
    st = P // 10
    want = [(y + P // 2, x + P // 2) for y in range(0, H - P + 1, st) for x in range(0, H - P + 1, st)
            if mask[y:y + P, x:x + P, 0].min() >= 0.5]
    assert [tuple(c) for c in S.random_centres.tolist()] == want                    # unfold order, zero unknown pixels
    ref = np.random.RandomState(0)
    for it in range(20):
        d = S.draw(3, 0.3)
        prob = ref.uniform(0, 1)
        src = "val" if prob < 0.5 else ("train" if 0.5 < prob < 0.8 else "same")
        pool = S.pool_val if src == "val" else S.pool_train
        sel = ref.choice(pool.shape[0], size=[n], replace=False)
        assert d["source"] == src and np.array_equal(d["cen"], pool[sel])
        if src != "same":
            rs = ref.choice(len(want), size=[n * 3], replace=False)                  # sampler.py:220
            assert d["k"] == 3 and d["weights"] is None
            np.testing.assert_array_equal(d["real_cen"], np.array(want, np.float64)[rs])
