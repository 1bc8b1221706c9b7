"""On-disk formats of the completion task, host side only.

* `config.odgt` -- one JSON line written by NPP_proposal/search.py:228-280: `fpath_masked_img`, `fpath_valid_mask`,
  `fpath_mask`, `fpath_gt_img`, `selected_angles`, `selected_periods`, `selected_shifts` (top-k lists), `distances`, ...
  `load_data` re-roots every `fpath_*` entry onto the data directory by file name (loaders/loaders.py:67-80).
* the four PNGs it names; read like cv2.imread(...)[:, :, ::-1] / 255 (RGB in [0,1]) and cv2.imread(..., 0) / 255 for the
  masks (loaders/loaders.py:92-101).  PIL replaces cv2 (not installed here); PNG decoding is exact, so the arrays are identical.
* the evaluation dump of NPP_completion/train.py:270-328: `testset_<iter:06d>/{pred_rgb_train_img, pred_rgb_val_img,
  gt_rgb_img, input_rgb_img, pred_rgb_img, pred_rgb_img_comp}.png`.
"""
import json
import os

import numpy as np


def _imread_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float64) / 255.0


def _imread_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.float64)[:, :, None] / 255.0


def imsave(path, arr):
    """plt.imsave of an (H,W,3) float image in [0,1] (train.py:319-328): clip, 8 bit."""
    from PIL import Image
    a = np.clip(np.asarray(arr, np.float64), 0.0, 1.0)
    Image.fromarray(np.uint8(np.rint(a * 255.0))).save(path)


def load_data(datadir):
    """loaders/loaders.py:67-80."""
    with open(os.path.join(datadir, "config.odgt"), "r") as f:
        info = json.loads(f.readline().rstrip())
    out = {}
    for key, val in info.items():
        if "fpath" in key:
            name = (val[0] if isinstance(val, list) else val).split("/")[-1]
            out[key] = os.path.join(datadir, name)
        else:
            out[key] = val
    return out


def patch_size_from_period(periods_top1):
    """loaders/loaders.py:133-134."""
    mp = float(max(periods_top1))
    return int(np.clip(mp + (32 - mp % 32), 64, 160))


def load_npp_completion(datadir, p_topk=3, invalid_as_unknown=False, normalize_type=1):
    """loaders/loaders.py:82-136 -> dict(img (H,W,3), mask (H,W,1), masked_img, valid_mask, i_train, i_val,
    shifts, angles, periods, patch_size).  Arrays are float32, [0,1]; coordinates (row, col)."""
    info = load_data(datadir)
    masked_img = _imread_rgb(info["fpath_masked_img"])
    img = _imread_rgb(info["fpath_gt_img"])
    valid_mask = _imread_gray(info["fpath_valid_mask"])
    mask = _imread_gray(info["fpath_mask"])
    mask = mask * valid_mask                                        # :103 filter invalid region
    if invalid_as_unknown:
        valid_mask = np.ones_like(valid_mask)
    i_train = np.stack(np.nonzero(mask * valid_mask)[:2], axis=1)    # :107-108
    i_val = np.stack(np.nonzero((1 - mask) * valid_mask)[:2], axis=1)
    if normalize_type == 2:
        img = (img - 0.5) * 2
    shifts = info["selected_shifts"][:p_topk]
    angles = info["selected_angles"][:p_topk]
    periods = info["selected_periods"][:p_topk]
    return dict(img=img.astype(np.float32), mask=mask.astype(np.float32), masked_img=masked_img.astype(np.float32),
                valid_mask=valid_mask.astype(np.float32), i_train=i_train, i_val=i_val, shifts=shifts,
                angles=np.asarray(angles, np.float32), periods=np.asarray(periods, np.float32),
                patch_size=patch_size_from_period(periods[0]), info=info)


def mask2ltrb(mask):
    """utils/miscs.py:17-20: (left, top, right, bottom) of the non-zero pixels of an (H,W) mask."""
    ys, xs = np.nonzero(np.asarray(mask).reshape(np.asarray(mask).shape[0], -1))
    return int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())


def draw_lattice(image_u8, base_xy, first_shift_xy, second_shift_xy, thickness=2):
    """utils/periodicity_visualizer.py:30-66 (GridProgram.gen_ij / draw at the image's own resolution): the two families of
    lattice lines base + i * first + [j_min, j_max] * second and base + j * second + [i_min, i_max] * first over the index range that
    covers the canvas corners, drawn `thickness` wide.  Like the reference (which slices off the LAST channel of its RGB input
    before drawing and puts it back afterwards) the lines set R = G = 255 and leave B alone.  PIL's line rasteriser stands in for
    cv2.line: a visualisation, not pixel-pinned."""
    from PIL import Image, ImageDraw
    img = np.asarray(image_u8, np.uint8)
    base = np.asarray(base_xy, np.float64)
    s1, s2 = np.asarray(first_shift_xy, np.float64), np.asarray(second_shift_xy, np.float64)
    H, W = img.shape[:2]
    corners = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], np.float64) * np.array([W, H], np.float64) - base
    ij = np.linalg.inv(np.stack([s1, s2], 1)) @ corners.T
    i_min, j_min = np.floor(ij.min(1)).astype(int)
    i_max, j_max = np.ceil(ij.max(1)).astype(int)
    lines = []
    for i in range(i_min, i_max):
        p = base + i * s1
        lines.append(np.concatenate([p + j_min * s2, p + j_max * s2]))
    for j in range(j_min, j_max):
        p = base + j * s2
        lines.append(np.concatenate([p + i_min * s1, p + i_max * s1]))
    layer = Image.new("L", (W, H), 0)
    dr = ImageDraw.Draw(layer)
    for x0, y0, x1, y1 in np.round(np.asarray(lines)).astype(np.int64).tolist() if lines else []:
        dr.line([(x0, y0), (x1, y1)], fill=255, width=thickness)
    hit = np.asarray(layer) > 0
    out = img.copy()
    out[..., 0][hit] = 255
    out[..., 1][hit] = 255
    return out


def write_detected_dir(outdir, img, mask, valid_mask, angles, periods, shifts, distances=None, masked_img=None, draw=False):
    """Write what NPP_proposal/search.py:228-280 leaves behind for one image (config.odgt + the four PNGs), e.g. for a
    synthetic image whose periodicity is known.  img (H,W,3) in [0,1]; mask / valid_mask (H,W[,1]) with 1 = known / valid.
    masked_img: the input's own masked image (search.py writes it through unchanged; default img * mask).  draw: also the
    reg_img_<i>.png lattice visualisations of :251-270 and their `fpath_reg_img_<i>` entries (one-element lists: the reference
    stores 1-tuples)."""
    from PIL import Image
    os.makedirs(outdir, exist_ok=True)
    img = np.asarray(img, np.float64)
    mask = np.asarray(mask, np.float64).reshape(img.shape[0], img.shape[1])
    valid = np.asarray(valid_mask, np.float64).reshape(img.shape[0], img.shape[1])
    names = {"fpath_masked_img": "masked_img.png", "fpath_valid_mask": "valid_mask.png", "fpath_mask": "unknown_mask.png",
             "fpath_gt_img": "gt_img.png"}
    jobs = []                                                         # (path, uint8 array): PNG encoding releases the GIL -> a few threads
    gt8 = np.uint8(img * 255)                                         # search.py:249-252 truncating casts
    jobs.append((os.path.join(outdir, names["fpath_gt_img"]), gt8))
    masked8 = np.uint8((img * mask[..., None] if masked_img is None else np.asarray(masked_img, np.float64)) * 255)
    jobs.append((os.path.join(outdir, names["fpath_masked_img"]), masked8))
    jobs.append((os.path.join(outdir, names["fpath_valid_mask"]), np.uint8(valid * 255)))
    jobs.append((os.path.join(outdir, names["fpath_mask"]), np.uint8(mask * 255)))
    odgt = {k: os.path.join(outdir, v) for k, v in names.items()}
    odgt.update(selected_angles=np.asarray(angles, np.float64).tolist(), selected_periods=np.asarray(periods, np.float64).tolist(),
                selected_shifts=[[list(map(float, s)) for s in sh] for sh in shifts],
                distances=list(distances) if distances is not None else [0.0] * len(shifts))
    if draw:
        left, top = mask2ltrb(valid)[:2]
        for i, sh in enumerate(shifts):
            path = os.path.join(outdir, f"reg_img_{i}.png")
            try:
                jobs.append((path, draw_lattice(masked8, (left, top), sh[0], sh[1])))
            except np.linalg.LinAlgError:                             # collinear displacement pair: nothing to draw
                continue
            odgt[f"fpath_reg_img_{i}"] = [path]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(4, len(jobs))) as pool:
        list(pool.map(lambda j: Image.fromarray(j[1]).save(j[0]), jobs))
    with open(os.path.join(outdir, "config.odgt"), "w") as f:
        json.dump(odgt, f)
        f.write("\n")
    return outdir


def dump_testset(savedir, pred_hw3, img, masked_img, mask, valid_mask):
    """The six PNGs of train.py:319-328 from one full-grid render: the reference renders the train and val pixels
    separately into two canvases; with a full-grid prediction these are pred * mask and pred * (1 - mask)."""
    os.makedirs(savedir, exist_ok=True)
    pred = np.asarray(pred_hw3, np.float64)
    m, v = np.asarray(mask, np.float64), np.asarray(valid_mask, np.float64)
    pred_train, pred_val = pred * m * v, pred * (1 - m) * v
    imsave(os.path.join(savedir, "pred_rgb_train_img.png"), pred_train)
    imsave(os.path.join(savedir, "pred_rgb_val_img.png"), pred_val)
    imsave(os.path.join(savedir, "gt_rgb_img.png"), np.asarray(img, np.float64) * v)
    imsave(os.path.join(savedir, "input_rgb_img.png"), np.asarray(masked_img, np.float64) * v)
    imsave(os.path.join(savedir, "pred_rgb_img.png"), pred_val + pred_train)
    imsave(os.path.join(savedir, "pred_rgb_img_comp.png"), pred_val + np.asarray(masked_img, np.float64) * v * m)


def rgb_to_gray_u8(img_u8):
    """cv2.cvtColor(img, cv2.COLOR_RGB2GRAY) on uint8: OpenCV's 14-bit fixed point (R 4899 + G 9617 + B 1868 + 8192) >> 14."""
    a = np.asarray(img_u8).astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def get_blur_map(img_u8, win_size=10, sv_num=3, thresh=50):
    """NPP_remapping/blur_detection.py:14-60: per pixel, the share of the top `sv_num` singular values of the surrounding
    2 win_size x 2 win_size gray block (mirror-padded like the reference's index arithmetic); normalised to [0,1]; pixels
    above the `thresh` percentile are 'blurry' candidates, eroded 20x, dilated 40x; the complement is the CLEAR mask.
    Returns (blur_map float64 (H,W), clear_mask float64 (H,W) in {0, 255}).  Vectorised: one batched SVD per image row."""
    import scipy.ndimage as ndimage
    gray = rgb_to_gray_u8(img_u8).astype(np.float64)
    H, W = gray.shape
    ws = win_size

    def mirror(n):                                                    # blur_detection.py:18-29
        idx = np.arange(n + 2 * ws)
        return np.where(idx < ws, ws - idx, np.where(idx > n + ws - 1, 2 * n - idx, idx - ws))
    pi, qj = mirror(H), mirror(W)
    if pi.max() >= H or qj.max() >= W:
        raise ValueError("image smaller than the blur window")
    new_img = gray[pi[:, None], qj[None, :]]
    blur = np.empty((H, W))
    win = np.lib.stride_tricks.sliding_window_view(new_img, (2 * ws, 2 * ws))       # (H+1, W+1, 2ws, 2ws)
    for i in range(H):
        sv = np.linalg.svd(win[i, :W], compute_uv=False)
        blur[i] = sv[:, :sv_num].sum(1) / (sv.sum(1) + 1e-6)
    blur = (blur - blur.min()) / (blur.max() - blur.min())
    binary = blur > np.percentile(blur, thresh)
    binary = ndimage.binary_erosion(binary, iterations=20)
    binary = ndimage.binary_dilation(binary, iterations=40)
    return blur, (~binary).astype(np.float64) * 255


def load_npp_remapping(datadir, p_topk=3, blur_thresh=50):
    """loaders/loaders.py:244-304 -> dict(img, clear_mask (H,W,1) in [0,1], valid_mask, shifts, angles, periods, patch_size)."""
    from PIL import Image
    info = load_data(datadir)
    img_u8 = np.asarray(Image.open(info["fpath_gt_img"]).convert("RGB"))
    valid = _imread_gray(info["fpath_valid_mask"])
    _, clear = get_blur_map(img_u8, thresh=blur_thresh)
    clear = clear[:, :, None] * valid / 255.0                          # :263-274 (valid already / 255 here)
    periods = info["selected_periods"][:p_topk]
    return dict(img=(img_u8 / 255.0).astype(np.float32), clear_mask=clear.astype(np.float32), valid_mask=valid.astype(np.float32),
                shifts=info["selected_shifts"][:p_topk], angles=np.asarray(info["selected_angles"][:p_topk], np.float32),
                periods=np.asarray(periods, np.float32), patch_size=patch_size_from_period(periods[0]), info=info)


def blur_with_mask(img, mask, sigma=3):
    """utils/ops.py:66-76 (skimage.filters.gaussian(sigma=3, multichannel=True): scipy's gaussian_filter per channel, mode
    'nearest', truncate 4): normalised masked blur  G(img * mask) / (G(mask) + 1e-6) * mask.  img (H,W,3), mask (H,W,1)."""
    from scipy.ndimage import gaussian_filter
    img, mask = np.asarray(img, np.float64), np.asarray(mask, np.float64)
    g = lambda a: gaussian_filter(a, sigma=(sigma, sigma, 0), mode="nearest", truncate=4.0)      # noqa: E731
    return g(img * mask) / (g(mask) + 1e-6) * mask


def load_npp_segmentation(datadir, p_topk=3, period_mask=None, non_period_mask=None):
    """loaders/loaders.py:141-239 -> dict(img, blur_img, period_mask (H,W,1), non_period_mask (H,W,1), valid_mask, shifts,
    angles, periods, patch_size).  The INITIAL coarse segmentation the reference computes with its vendored imsegm package
    (SLIC superpixels + GMM + graph cut, loaders.py:162-205) is outside this build: its result is an input
    here -- `period_mask` / `non_period_mask` arrays, or `period_mask.png` / `non_period_mask.png` (white = member) next to
    config.odgt.  Everything after it is reproduced: the masked Gaussian blur of the image the fit trains on (:157-159), the
    top-k periodicity and the patch size rule (:232-236)."""
    info = load_data(datadir)
    img = _imread_rgb(info["fpath_gt_img"])
    valid = _imread_gray(info["fpath_valid_mask"])
    blur = blur_with_mask(img * 255.0, valid) / 255.0

    def get(arr, name):
        if arr is not None:
            return (np.asarray(arr, np.float64).reshape(img.shape[0], img.shape[1], 1) > 0.5).astype(np.float64)
        path = os.path.join(datadir, name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: the initial periodic / non-periodic segmentation (imsegm in the reference, "
                                    f"loaders/loaders.py:162-205) is not computed by this build; pass it or put {name} there")
        return (_imread_gray(path) > 0.5).astype(np.float64)
    pm, npm = get(period_mask, "period_mask.png"), get(non_period_mask, "non_period_mask.png")
    periods = info["selected_periods"][:p_topk]
    f = lambda a: np.asarray(a, np.float32)          # noqa: E731
    return dict(img=f(img), blur_img=f(blur), period_mask=f(pm * valid), non_period_mask=f(npm), valid_mask=f(valid),
                shifts=info["selected_shifts"][:p_topk], angles=f(info["selected_angles"][:p_topk]), periods=f(periods),
                patch_size=patch_size_from_period(periods[0]), info=info)
