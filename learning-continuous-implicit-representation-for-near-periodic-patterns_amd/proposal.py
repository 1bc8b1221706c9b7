"""Periodicity proposal, the search half: NPP_proposal/feature_searching.py:77-156,208-339 on a given
feature map -- candidate displacement vectors, their brute-force losses (npp_shift_search), the best pair of
displacements and its (angles, periods) -- producing the candidate list that light.ProposalRanker ranks
(NPP_proposal/search.py:85-215) -- and the feature extraction in front of it (feature_searching.py:14-75,158-204):
AlexNet conv1 activations (models/model_def.py:82-117: the hooked `features[0]`, Conv2d(3, 64, 11, stride 4, padding 5) of
models/alexnet.py:19) + gray image + mask, optionally reduced to their Canny edges.  The checkpoint the reference loads
(`alexnet-owt-4df8aa71.pth`) is not part of its tree: AlexConv1 takes the user's state_dict (features.0.weight / bias) and
only builds fixed-seed random filters on explicit request (a warning says so).  The OpenCV steps are restated in cvlite.py
(parity unpinned: no cv2 in this image).  Any (C, h, w) map whose last channel is the mask can be searched.
"""
import math
import warnings

import numpy as np
import torch

from . import ops, cvlite
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


# ---- feature extraction in front of the search (feature_searching.py:14-75) ---------------------------------------------
_IMAGENET_MEAN, _IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


class AlexConv1:
    """The activation the reference hooks (model_def.py:97-117): output of models/alexnet.py's `features[0]` =
    Conv2d(3, 64, kernel 11, stride 4, padding 5), before the ReLU.  state_dict: torchvision's alexnet checkpoint
    (`features.0.weight` (64,3,11,11), `features.0.bias`).  The convolution runs as im2col + the library's exact-fp32 dense
    layer (npp_linear_fwd)."""

    def __init__(self, state_dict=None, device="cuda", allow_random=False, seed=7):
        self.device = ops.select_device(device)
        if state_dict is None:
            if not allow_random:
                raise ValueError("AlexConv1 needs the AlexNet checkpoint the reference downloads (alexnet-owt-4df8aa71.pth: keys "
                                 "features.0.weight / features.0.bias); allow_random=True builds fixed-seed random filters instead")
            warnings.warn("npp_amd.proposal: AlexNet conv1 built with fixed-seed RANDOM filters: the proposals will differ from the "
                          "reference's", stacklevel=2)
            g = torch.Generator().manual_seed(seed)
            w, b = torch.randn(64, 3, 11, 11, generator=g) * (2.0 / 363) ** 0.5, torch.zeros(64)
        else:
            w = state_dict.get("features.0.weight", state_dict.get("0.weight"))
            b = state_dict.get("features.0.bias", state_dict.get("0.bias"))
            if w is None or b is None or tuple(w.shape) != (64, 3, 11, 11) or tuple(b.shape) != (64,):
                raise KeyError("AlexNet state_dict: features.0.weight (64,3,11,11) / features.0.bias (64,) missing or mis-shaped")
            w, b = torch.as_tensor(w).detach().float(), torch.as_tensor(b).detach().float()
        self.w = w.reshape(64, -1).contiguous().to(self.device)
        self.b = b.contiguous().to(self.device)

    def __call__(self, im_u8):
        """im (H,W,3) uint8 -> (64, Hp/4, Wp/4) fp32, Hp/Wp = H/W padded with zeros to multiples of 32 on the right / bottom
        BEFORE ToTensor + Normalize (feature_searching.py:20-24, utils/ops.py:47-53)."""
        im = np.asarray(im_u8)[..., :3]
        H, W = im.shape[:2]
        Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
        pad = np.zeros((Hp, Wp, 3), np.uint8)
        pad[:H, :W] = im
        x = torch.from_numpy(pad).to(self.device).permute(2, 0, 1).float().div(255.0)
        mean = torch.tensor(_IMAGENET_MEAN, device=self.device).view(3, 1, 1)
        std = torch.tensor(_IMAGENET_STD, device=self.device).view(3, 1, 1)
        x = ((x - mean) / std)[None]
        cols, ho, wo = ops.im2col(x.contiguous(), 11, 4, 5)
        y = torch.empty((ho * wo, 64), dtype=torch.float32, device=self.device)
        ops.linear_fwd(cols, self.w, self.b, 0, y)
        return y.reshape(ho, wo, 64).permute(2, 0, 1).contiguous()


def im2act(im_u8, mask, conv1=None, gray_only=False, device="cuda"):
    """feature_searching.py:14-50: (activation (C+2 | 2, h, w) * mask, mask (1, h, w)) at a quarter of the image size.
    Channels: conv1 activations (unless gray_only), the gray image resized image -> 2x(h, w) -> (h, w), the mask."""
    im = np.asarray(im_u8)[..., :3]
    dev = ops.select_device(device) if conv1 is None else conv1.device
    h, w = im.shape[0] // 4, im.shape[1] // 4
    m = torch.from_numpy(np.ascontiguousarray(cvlite.resize_nearest(np.asarray(mask), (w, h))).astype(np.float32)).to(dev)[None]
    g = cvlite.rgb_to_gray_u8(im)
    g = cvlite.resize_linear_u8(g, (w * 2, h * 2))
    g = cvlite.resize_linear_u8(g, (w, h))
    g = torch.from_numpy(g.astype(np.float32)).to(dev)[None]
    if gray_only:
        act = torch.cat([g, m], 0)
    else:
        if conv1 is None:
            raise ValueError("im2act: pass the AlexConv1 feature extractor (or gray_only=True)")
        act = torch.cat([conv1(im)[:, :h, :w], g, m], 0)
    return act * m, m


def act2edge(activation, mask):
    """feature_searching.py:53-69: per channel, normalise to uint8, Canny on the blurred channel inside the mask eroded 4 times
    (utils/miscs.py:22-33); the edge maps are summed (/255).  -> (2, h, w): [edge count, mask]."""
    a = cvlite.normalize_to_uint8(activation.detach().cpu().numpy(), channel_idx=(1, 2))
    m = mask[0].detach().cpu().numpy()
    edge = np.zeros((1,) + a.shape[1:])
    for c in range(a.shape[0]):
        edge += cvlite.canny_masked(a[c], m) / 255
    return torch.cat([torch.from_numpy(edge).float(), torch.from_numpy(m).float()[None]])


def search_periodicity_by_feat(img_u8, mask, repeat_range=(2, 32, 5), edge_searching=False, gray_only=False, threshold=10,
                               conv1=None, device="cuda"):
    """feature_searching.py:158-204: feature map -> (optional) edge masking -> brute-force displacement search per
    repeat-range group -> ([angles], [periods], [shifts]) scaled back to image pixels."""
    act, m = im2act(img_u8, mask, conv1=conv1, gray_only=gray_only, device=device)
    if edge_searching:
        edge = act2edge(act[:-1], m).to(act.device)
        act = act * edge[[0]]
    ratio = float(np.round(np.asarray(img_u8).shape[0] / act.shape[1]))
    cands = feature_search(act, m[0], repeat_range=repeat_range, edge_searching=edge_searching, scale=ratio)
    return [c[0] for c in cands], [c[1] for c in cands], [c[2] for c in cands]
