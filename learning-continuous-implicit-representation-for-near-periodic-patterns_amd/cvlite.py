"""The handful of OpenCV calls in front of the periodicity search (NPP_proposal/feature_searching.py:14-75, utils/miscs.py:22-33),
restated in NumPy because `cv2` is a dependency of the reference that this image does not carry: cvtColor(RGB2GRAY),
resize(INTER_LINEAR / INTER_NEAREST), GaussianBlur((3, 3), 0) and Canny(low, high) on 8-bit single-channel images, each in
OpenCV's own fixed-point arithmetic as published in its sources (imgproc: color_yuv / resize / smooth / canny).
PARITY UNPINNED: without cv2 here the restatement is checked against closed-form cases only (tests/test_proposal_frontend.py),
not against OpenCV outputs.  Host-side, once per image -- not on the hot path."""
import numpy as np
import scipy.ndimage as ndimage

from .io import rgb_to_gray_u8  # noqa: F401  (cv2.cvtColor(..., COLOR_RGB2GRAY))


def resize_nearest(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): source index = min(floor(dst * src / dst_size), src - 1)."""
    w, h = int(dsize[0]), int(dsize[1])
    a = np.asarray(img)
    sy = np.minimum((np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    sx = np.minimum((np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[sy][:, sx]


def _linear_taps(src, dst):
    """OpenCV's bilinear tap table for one axis: source index, and the two 11-bit fixed-point coefficients."""
    scale = src / dst
    f = (np.arange(dst) + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = f - s
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= src - 1
    s[hi], f[hi] = src - 1, 0.0
    c1 = np.rint(f * 2048).astype(np.int64)
    c0 = np.rint((1.0 - f) * 2048).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, c0, c1


def resize_linear_u8(img_u8, dsize):
    """cv2.resize(img, (w, h)) (INTER_LINEAR) on a uint8 single-channel image: horizontal pass into 11-bit fixed point,
    vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 (resize.cpp VResizeLinear, 8u)."""
    w, h = int(dsize[0]), int(dsize[1])
    a = np.asarray(img_u8).astype(np.int64)
    x0, x1, a0, a1 = _linear_taps(a.shape[1], w)
    y0, y1, b0, b1 = _linear_taps(a.shape[0], h)
    rows = a[:, x0] * a0[None, :] + a[:, x1] * a1[None, :]                       # (H, w), scaled by 2048
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def gaussian_blur3_u8(img_u8):
    """cv2.GaussianBlur(img, (3, 3), 0) on uint8: the fixed kernel [1 2 1] / 4 per axis, BORDER_REFLECT_101, one rounding
    at the end: floor((sum of the 3x3 weights [1 2 1; 2 4 2; 1 2 1] * p + 8) / 16)."""
    p = np.pad(np.asarray(img_u8).astype(np.int64), 1, mode="reflect")
    h = p[:, :-2] + 2 * p[:, 1:-1] + p[:, 2:]
    v = h[:-2] + 2 * h[1:-1] + h[2:]
    return ((v + 8) >> 4).astype(np.uint8)


def canny_u8(img_u8, low, high):
    """cv2.Canny(img, low, high) (aperture 3, L2gradient=False) on a uint8 single-channel image -> uint8 {0, 255}.
    Sobel 3x3 with replicated borders; magnitude |dx| + |dy|; non-maximum suppression over the four sectors with OpenCV's
    integer tangent test (tan 22.5 deg = 13573 / 2^15) and its strict / non-strict comparison pattern; hysteresis: surviving
    pixels above `low` that are 8-connected to a surviving pixel above `high`."""
    a = np.pad(np.asarray(img_u8).astype(np.int64), 1, mode="edge")
    dx = (a[:-2, 2:] + 2 * a[1:-1, 2:] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[1:-1, :-2] + a[2:, :-2])
    dy = (a[2:, :-2] + 2 * a[2:, 1:-1] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[:-2, 1:-1] + a[:-2, 2:])
    low, high = int(np.floor(low)), int(np.floor(high))
    mag = np.abs(dx) + np.abs(dy)
    m = np.pad(mag, 1)                                                            # zero magnitude outside the image
    c = m[1:-1, 1:-1]
    x, y = np.abs(dx), np.abs(dy) << 15
    tg22 = x * 13573
    tg67 = tg22 + (x << 16)
    horiz = y < tg22
    vert = (~horiz) & (y > tg67)
    diag = ~(horiz | vert)
    s_neg = ((dx ^ dy) < 0)                                                        # gradient components of opposite sign
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    up_l, up_r, dn_l, dn_r = m[:-2, :-2], m[:-2, 2:], m[2:, :-2], m[2:, 2:]
    keep = np.zeros(c.shape, bool)
    keep |= horiz & (c > left) & (c >= right)
    keep |= vert & (c > up) & (c >= down)
    # s = -1 when the signs differ: compares with (row-1, col+1) and (row+1, col-1); s = +1: (row-1, col-1) and (row+1, col+1)
    keep |= diag & s_neg & (c > up_r) & (c > dn_l)
    keep |= diag & (~s_neg) & (c > up_l) & (c > dn_r)
    weak = keep & (c > low)
    strong = weak & (c > high)
    lab, n = ndimage.label(weak, structure=np.ones((3, 3), int))
    if n == 0:
        return np.zeros(c.shape, np.uint8)
    hit = np.zeros(n + 1, bool)
    hit[np.unique(lab[strong])] = True
    hit[0] = False
    return (hit[lab].astype(np.uint8)) * 255


def canny_masked(img_u8, mask):
    """utils/miscs.py:22-33 `canny(img, mask)`: blur 3x3, Canny(10, 100), edges outside the mask eroded 4 times removed."""
    img = np.asarray(img_u8)
    gray = rgb_to_gray_u8(img) if img.ndim == 3 else img
    blur = gaussian_blur3_u8(gray)
    er = ndimage.binary_erosion(np.asarray(mask) != 0, iterations=4).astype(np.float64)
    return canny_u8(blur, 10, 100) * er


def normalize_to_uint8(array, channel_idx=-1):
    """utils/miscs.py:42-48; a constant slice (0 / 0 in the reference) gives zeros."""
    a = np.asarray(array, np.float64)
    mx, mn = a.max(axis=channel_idx, keepdims=True), a.min(axis=channel_idx, keepdims=True)
    rng = np.where(mx > mn, mx - mn, 1.0)
    return np.uint8((a - mn) / rng * 255)
