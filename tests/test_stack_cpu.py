"""Host-side checks of the stacked-launch plumbing (no GPU): the work-item numbering the kernels use (csrc/npp_common.h stack_decode,
csrc/npp_mlp_wgrad.hip), restated here, is a bijection onto (image, item) for every stack size and gives each image its own XCDs;
the per-iteration record has the layout the kernels read; the split-K choice fills the chip in whole rounds."""
import ctypes as C

import numpy as np
import pytest


def _decode(M, n_items, b):
    """stack_decode of csrc/npp_common.h for linear workgroup id b -> (image, item) or None for a surplus workgroup."""
    g = 8 // M if M in (1, 2, 4, 8) else 0
    if g:
        xcd, slot = b & 7, b >> 3
        img, item = xcd // g, slot * g + xcd % g
    else:
        img, item = b // n_items, b % n_items
    return (img, item) if item < n_items else None


def _grid(M, n_items):
    g = 8 // M if M in (1, 2, 4, 8) else 0
    return 8 * ((n_items + g - 1) // g) if g else M * n_items


def _decode_wgrad(M, n_items, b):
    """the grouped weight-gradient launch's numbering: every XCD of an image owns a CONTIGUOUS item range."""
    g = 8 // M if M in (1, 2, 4, 8) else 0
    if g:
        xcd, slot = b & 7, b >> 3
        img, xl, gx, lslot = xcd // g, xcd % g, g, slot
    else:
        img, xl, gx, lslot = b // n_items, 0, 1, b - (b // n_items) * n_items
    q, r = divmod(n_items, gx)
    if lslot >= (q + 1 if xl < r else q):
        return None
    return img, (xl * (q + 1) if xl < r else r * (q + 1) + (xl - r) * q) + lslot


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("n_items", [1, 7, 63, 252, 416, 417])
def test_work_item_numbering_is_a_bijection_and_images_own_their_xcds(M, n_items):
    for dec in (_decode, _decode_wgrad):
        seen = {}
        for b in range(_grid(M, n_items)):
            t = dec(M, n_items, b)
            if t is None:
                continue
            assert t not in seen, (dec.__name__, t)
            seen[t] = b & 7
        assert len(seen) == M * n_items and {i for i, _ in seen} == set(range(M))
        if M in (1, 2, 4, 8):                                   # image m runs on XCDs [m * 8 / M, (m + 1) * 8 / M) only
            g = 8 // M
            for (img, _), xcd in seen.items():
                assert xcd // g == img
    if M in (1, 2, 4, 8):                                       # wgrad: the items of one XCD are one contiguous range
        per = {}
        for b in range(_grid(M, n_items)):
            t = _decode_wgrad(M, n_items, b)
            if t is not None:
                per.setdefault((t[0], b & 7), []).append(t[1])
        for items in per.values():
            assert sorted(items) == list(range(min(items), max(items) + 1))


def test_stack_iter_record_layout():
    """npp_stack_iter (include/npp_hip.h) == _lib.StackIter == csrc StackIter (static_assert 48 bytes)."""
    from npp_amd._lib import StackIter
    assert C.sizeof(StackIter) == 48
    offs = {n: getattr(StackIter, n).offset for n, _ in StackIter._fields_}
    assert [offs[k] for k in ("active", "k", "comp", "with_lp", "x0", "nk", "same", "step_size", "inv_sqrt_bc2")] == [0, 4, 8, 12, 16, 20, 24, 32, 36]
    it = (StackIter * 3)()
    it[1].active, it[1].nk, it[1].x0, it[1].step_size = 1, 6, 2, 0.5
    raw = np.frombuffer(bytes(it), np.uint8).view(np.int32).reshape(3, 12)
    assert raw[1, 0] == 1 and raw[1, 5] == 6 and raw[1, 4] == 2 and raw[0].sum() == 0
    assert np.frombuffer(bytes(it), np.float32).reshape(3, 12)[1, 8] == 0.5


def test_stacked_split_k_choice_fills_whole_rounds():
    """StackedFit._pick_ksplit: splits per image minimising rounds x rows per workgroup (21 tiles at K = 3, 256 CUs, 416 row tiles)."""
    tiles, cus, n_wg = 21, 256, 416
    pick = lambda M: min(range(1, 25), key=lambda ks: (-(-tiles * ks * M // cus) * -(-n_wg // ks), ks))
    assert [pick(M) for M in (1, 2, 4, 8)] == [12, 6, 3, 3]
    for M in (1, 2, 4):
        assert tiles * pick(M) * M <= cus                       # one round


def test_main_stacked_records_a_failure_per_image_and_creates_nothing(tmp_path):
    """an image whose preparation fails -- here: detection directories that do not exist -- is recorded for that
    image only, aborts nothing, and leaves no result directory behind that a re-run would take for a finished fit
    (NPP_completion/train.py:42-44 skips existing directories)."""
    from npp_amd import train
    base = tmp_path / "results"
    argvs = [["--datadir", str(tmp_path / "detected" / f"missing{i}"), "--basedir", str(base), "--p_topk", "3", "--random-trunks",
              "--netwidth", "256", "--device", "cpu"] for i in range(2)]
    out = train.main_stacked(argvs)
    assert out == [None, None]
    assert all(e is not None for e in train.main_stacked.errors) and train.main_stacked.last_error is train.main_stacked.errors[0]
    assert not base.exists()
