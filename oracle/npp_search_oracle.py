"""NumPy restatement of the brute-force displacement search (NPP_proposal/feature_searching.py:77-156,
208-277,279-339).  TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/g11_search.npz (the reference's own functions)."""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32

__all__ = ["possible_shifts", "shift_losses", "periodicity_from_losses", "feature_search_oracle"]


def possible_shifts(act_hw, repeat_range_x, repeat_range_y):
    """generate_possible_shifts (:267-277): dx in [-w // rx0, w // rx0), dy in [0, h // ry0), dx-major order; tiny shifts
    (|dx| <= w // rx1 and dy <= h // ry1) are dropped.  Python floor division like torch.arange's integer arguments."""
    h, w = act_hw
    dxs = np.arange(-w // repeat_range_x[0], w // repeat_range_x[0])
    dys = np.arange(0, h // repeat_range_y[0])
    dx, dy = np.meshgrid(dxs, dys, indexing="ij")
    s = np.stack([dx.reshape(-1), dy.reshape(-1)], 1).astype(np.int64)
    keep = (np.abs(s[:, 0]) > w // repeat_range_x[1]) | (s[:, 1] > h // repeat_range_y[1])
    return s[keep]


def shift_losses(act, mask, shifts, edge_searching=True):
    """compute_loss (:208-264): for shift (dx, dy), over all positions (y, x) of the map,
    sum_c f(act_c[y + dy, x + dx], act_c[y, x]) * mask[y, x] * mask[y + dy, x + dx], c over all channels but the last,
    f = -a b (edge_searching) or (a - b)^2; everything outside the map is zero (the reference pads a zero canvas)."""
    act = np.asarray(act, F32)
    mask = np.asarray(mask, F32)
    C, h, w = act.shape
    out = np.zeros(len(shifts), F32)
    for i, (dx, dy) in enumerate(np.asarray(shifts)):
        sh_a = np.zeros_like(act)
        sh_m = np.zeros_like(mask)
        y0, y1 = max(0, -dy), min(h, h - dy)
        x0, x1 = max(0, -dx), min(w, w - dx)
        if y1 > y0 and x1 > x0:
            sh_a[:, y0:y1, x0:x1] = act[:, y0 + dy:y1 + dy, x0 + dx:x1 + dx]
            sh_m[y0:y1, x0:x1] = mask[y0 + dy:y1 + dy, x0 + dx:x1 + dx]
        d = -sh_a[:-1] * act[:-1] if edge_searching else (sh_a[:-1] - act[:-1]) ** 2
        out[i] = np.sum(d * mask[None] * sh_m[None], dtype=np.float64)
    return out


def _angle_diff(v1, v2):
    a, b = v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)
    return math.acos(float(np.clip(np.dot(a, b), -1.0, 1.0)))


def periodicity_from_losses(losses, shifts, minimum_angle=20):
    """generate_periodicity (:118-156): best shift = smallest loss; second = the next one (in loss order) whose direction
    differs by more than minimum_angle (and less than 180 - minimum_angle) degrees; angle_i = 180 - atan2(dy, dx) of the
    OTHER shift, period_i = |shift_i| sin(angle between the two)."""
    order = np.argsort(np.asarray(losses), kind="stable")
    s = np.asarray(shifts)[order].astype(F32)
    th = np.degrees(np.arctan2(s[:, 1], s[:, 0]))
    diff = np.abs(th - th[0])
    idx = np.nonzero((diff > minimum_angle) & (diff < 180 - minimum_angle))[0]
    if idx.size == 0:
        return None, None, None
    sel = [s[0], s[idx[0]]]
    ang = [180.0 - math.degrees(math.atan2(sel[1][1], sel[1][0])), 180.0 - math.degrees(math.atan2(sel[0][1], sel[0][0]))]
    phi = _angle_diff(sel[0], sel[1])
    per = [float(np.linalg.norm(sel[0])) * math.sin(phi), float(np.linalg.norm(sel[1])) * math.sin(phi)]
    return ang, per, sel


def feature_search_oracle(act, mask, repeat_range=(3, 6, 1), edge_searching=True):
    """feature_search (:77-115)."""
    A, P, S = [], [], []
    for i in range(repeat_range[0], repeat_range[1], repeat_range[2]):
        r = (i, i + repeat_range[2])
        sh = possible_shifts(act.shape[1:], r, r)
        if len(sh) == 0:
            continue
        a, p, s = periodicity_from_losses(shift_losses(act, mask, sh, edge_searching), sh)
        if a is None:
            continue
        A.append(a); P.append(p); S.append(s)
    return A, P, S
