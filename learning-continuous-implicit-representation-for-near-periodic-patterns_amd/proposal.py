"""Periodicity proposal, the search half: NPP_proposal/feature_searching.py:77-156,208-339 on a given
feature map -- candidate displacement vectors, their brute-force losses (npp_shift_search), the best pair of
displacements and its (angles, periods) -- producing the candidate list that light.ProposalRanker ranks
(NPP_proposal/search.py:85-215).  The feature extraction in front of it (AlexNet conv1 activations + Canny edges,
feature_searching.py:14-75) needs a checkpoint that is not part of the reference tree and is not built: any (C, h, w) map
whose last channel is the reference's extra (gray) channel can be searched.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import lib, check


def generate_possible_shifts(act_hw, repeat_range_x=(1, 10), repeat_range_y=(10, 20)):
    """feature_searching.py:267-277 -> (n, 2) int64 (dx, dy)."""
    h, w = int(act_hw[0]), int(act_hw[1])
    dxs = np.arange(-w // repeat_range_x[0], w // repeat_range_x[0])
    dys = np.arange(0, h // repeat_range_y[0])
    dx, dy = np.meshgrid(dxs, dys, indexing="ij")
    s = np.stack([dx.reshape(-1), dy.reshape(-1)], 1).astype(np.int64)
    keep = (np.abs(s[:, 0]) > w // repeat_range_x[1]) | (s[:, 1] > h // repeat_range_y[1])
    return s[keep]


def compute_loss(activation, mask, possible_shifts, edge_searching=True):
    """feature_searching.py:208-264 on the GPU: activation (C,h,w), mask (h,w) tensors, shifts (n,2) -> losses (n,)."""
    act = activation.contiguous().float()
    m = mask.reshape(act.shape[1], act.shape[2]).contiguous().float()
    sh = torch.as_tensor(np.ascontiguousarray(possible_shifts, dtype=np.int32)).to(act.device)
    out = torch.empty(sh.shape[0], dtype=torch.float32, device=act.device)
    C, h, w = act.shape
    check(lib().npp_shift_search(ops._p(act), ops._p(m), C, h, w, ops._p(sh), sh.shape[0], int(bool(edge_searching)), ops._p(out),
                                 ops._stream()), "npp_shift_search")
    return out


def generate_periodicity(losses, possible_shifts, minimum_angle=20):
    """feature_searching.py:118-156 (+ :279-339): (angles [2], periods [2], shifts [2 x (dx, dy)]) or (None, None, None)."""
    losses = losses.detach().cpu().numpy() if isinstance(losses, torch.Tensor) else np.asarray(losses)
    order = np.argsort(losses, kind="stable")
    s = np.asarray(possible_shifts)[order].astype(np.float32)
    th = np.degrees(np.arctan2(s[:, 1], s[:, 0]))
    diff = np.abs(th - th[0])
    idx = np.nonzero((diff > minimum_angle) & (diff < 180 - minimum_angle))[0]
    if idx.size == 0:
        return None, None, None
    sel = [s[0], s[idx[0]]]
    # the angle of the first displacement is computed from the second one and vice versa (:143-144)
    angles = [180.0 - math.degrees(math.atan2(sel[1][1], sel[1][0])), 180.0 - math.degrees(math.atan2(sel[0][1], sel[0][0]))]
    u0, u1 = sel[0] / np.linalg.norm(sel[0]), sel[1] / np.linalg.norm(sel[1])
    phi = math.acos(float(np.clip(np.dot(u0, u1), -1.0, 1.0)))
    periods = [float(np.linalg.norm(sel[0])) * math.sin(phi), float(np.linalg.norm(sel[1])) * math.sin(phi)]
    return angles, periods, sel


def feature_search(activation, mask, repeat_range=(3, 6, 1), edge_searching=True, scale=1.0):
    """feature_searching.py:77-115: one (angles, periods, shifts) candidate per repeat-range group; `scale` multiplies periods and
    shifts like search_periodicity_by_feat does with image size / map size (:196-203)."""
    cands = []
    for i in range(repeat_range[0], repeat_range[1], repeat_range[2]):
        r = (i, i + repeat_range[2])
        sh = generate_possible_shifts(activation.shape[1:], r, r)
        if len(sh) == 0:
            continue
        a, p, s = generate_periodicity(compute_loss(activation, mask, sh, edge_searching), sh)
        if a is None:
            continue
        cands.append((np.asarray(a, np.float32), np.asarray(p, np.float32) * scale, [list(map(float, v * scale)) for v in s]))
    return cands
