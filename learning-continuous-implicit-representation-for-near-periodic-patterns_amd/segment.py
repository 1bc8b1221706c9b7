"""Segmentation variant: the evaluation of NPP_segmentation/train.py:337-406 -- which pixels of the
non-periodic candidate region does the fitted periodic pattern explain?

Training is the completion loop on the masked-blurred image with the initial periodic region as the known mask
(CompletionFit(task="segmentation"): NPP_segmentation/train.py:148-286).  This module is what runs once at the end:
  1. grayscale of the fitted image and of the blurred input (kornia.rgb_to_grayscale: 0.299 R + 0.587 G + 0.114 B);
  2. criterion 1: |difference| clamped to [0, 0.99] below l1_thresh (:349-353);
  3. criterion 2: LPIPS(net='alex', spatial=True) per-layer distance maps (externel_lib/lpips/lpips.py:92-133 with
     retPerLayer=True: scaling layer, AlexNet taps relu1..relu5, channel-unit normalisation, squared difference, the
     non-negative 1x1 `lin` layer, bilinear upsampling to the image) below lpips_thresh inside the non-periodic candidates,
     for the first lpips_layers layers (:362-385);
  4. union over layers of "not (both criteria)", hole filling, removal of components under 500 pixels (:390-393).
Not on the hot path (once per fit).  The AlexNet convolutions run as im2col + the library's exact-fp32 dense-layer kernel
(npp_linear_fwd); unfold / max-pool / bilinear resize / elementwise steps are torch tensor plumbing.  AlexNet's pretrained
weights are not available offline: pass torchvision's state_dict, as for the VGG trunks; the five `lin`
vectors are lpips' weights/v0.1/alex.pth.
"""
import numpy as np
import torch

from . import ops

# torchvision.models.alexnet().features: (index, out, in, kernel, stride, pad); max-pool(3, 2) before conv 3 and conv 6
_ALEX = [(0, 64, 3, 11, 4, 2), (3, 192, 64, 5, 1, 2), (6, 384, 192, 3, 1, 1), (8, 256, 384, 3, 1, 1), (10, 256, 256, 3, 1, 1)]
_ALEX_POOL_BEFORE = {3, 6}
_SHIFT = (-0.030, -0.088, -0.188)                 # lpips.py:136-143 ScalingLayer
_SCALE = (0.458, 0.448, 0.450)
_RELU = 2                                         # npp_linear_fwd activation code


def rgb_to_grayscale(x):
    """kornia.rgb_to_grayscale on (..., 3, H, W)."""
    return 0.299 * x[..., 0:1, :, :] + 0.587 * x[..., 1:2, :, :] + 0.114 * x[..., 2:3, :, :]


class AlexFeatures:
    """relu1..relu5 of torchvision's AlexNet `features` (externel_lib/lpips/pretrained_networks.py:56-94)."""

    def __init__(self, state_dict=None, device="cuda", seed=99):
        self.device = ops.select_device(device)
        self.layers = []
        if state_dict is None:
            import warnings
            warnings.warn("npp_amd.segment: AlexNet built with fixed-seed RANDOM weights: pass torchvision's alexnet state_dict "
                          "to reproduce the reference's LPIPS(alex) criterion", stacklevel=2)
            g = torch.Generator().manual_seed(seed)
        for idx, co, ci, k, st, pd in _ALEX:
            if state_dict is not None:
                w = state_dict.get(f"features.{idx}.weight", state_dict.get(f"{idx}.weight"))
                b = state_dict.get(f"features.{idx}.bias", state_dict.get(f"{idx}.bias"))
                if w is None or b is None or tuple(w.shape) != (co, ci, k, k):
                    raise KeyError(f"AlexNet state_dict: features.{idx}.weight/bias missing or mis-shaped (need {(co, ci, k, k)})")
                w, b = w.detach().float(), b.detach().float()
            else:
                w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
                b = torch.zeros(co)
            self.layers.append((idx, k, st, pd, w.reshape(co, -1).contiguous().to(self.device), b.contiguous().to(self.device)))

    def features_nhwc(self, x):
        """The five ReLU taps POSITION-MAJOR, (N, h, w, C) each: im2col rows (npp_im2col) x the filter matrix (npp_linear_fwd, bias + ReLU
        in its epilogue), nn.MaxPool2d(3, 2) in front of conv2 / conv3 (npp_maxpool_nhwc); no transposition between the layers."""
        outs, nhwc = [], False
        for idx, k, st, pd, w, b in self.layers:
            if idx in _ALEX_POOL_BEFORE:
                x = ops.maxpool_nhwc(x, 3, 2)
            N = x.shape[0]
            cols, ho, wo = ops.im2col(x, k, st, pd, nhwc=nhwc)
            y = torch.empty((cols.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
            ops.linear_fwd(cols, w, b, _RELU, y)
            x, nhwc = y.view(N, ho, wo, -1), True
            outs.append(x)
        return outs

    def __call__(self, x):
        """The taps as (N, C, h, w) views (the layout of the reference's alexnet slices, pretrained_networks.py:60-96)."""
        return [f.permute(0, 3, 1, 2) for f in self.features_nhwc(x)]


def lpips_alex_spatial(in0, in1, alex, lins, normalize=True):
    """LPIPS(net='alex', spatial=True).forward(in0, in1, use_robust=False, retPerLayer=True, normalize) -> (val, [per-layer
    maps]) each (N,1,H,W).  in0 / in1: (N,1,H,W) or (N,3,H,W) (the reference feeds GRAYSCALE images: the scaling layer's
    (1,3,1,1) constants broadcast them to three channels, lpips.py:141-143).  Per tap: npp_lpips_spatial_layer on the position-major
    features, npp_resize_bilinear to the input size; the sum over the taps is accumulated by the same launches."""
    dev = in0.device
    sh, sc = torch.tensor(_SHIFT, device=dev).view(1, 3, 1, 1), torch.tensor(_SCALE, device=dev).view(1, 3, 1, 1)
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    N, _, H, W = in0.shape
    f0, f1 = alex.features_nhwc(((in0 - sh) / sc).contiguous()), alex.features_nhwc(((in1 - sh) / sc).contiguous())
    res = []
    for a, b, lin in zip(f0, f1, lins):
        lin_t = lin.to(device=dev, dtype=torch.float32).reshape(-1).contiguous() if isinstance(lin, torch.Tensor) else \
            torch.as_tensor(lin, dtype=torch.float32).reshape(-1).contiguous().to(dev)
        d = ops.lpips_spatial_layer(a.contiguous(), b.contiguous(), lin_t)
        res.append(ops.resize_bilinear(d, H, W)[:, None])
    # lpips.py:125-127: `val = res[0]; for l in 1..L-1: val += res[l]` accumulates IN PLACE, so the list entry 0 the caller
    # receives (retPerLayer=True) IS the sum over all layers, not layer 0's own map -- and with the default lpips_layers = 1
    # that sum is what NPP_segmentation/train.py:366-372 thresholds.  Reproduced.
    val = res[0].clone()
    for r in res[1:]:
        val = val + r
    res = [val] + res[1:]
    return val, res


def remove_small_objects(mask, min_size=500):
    """skimage.morphology.remove_small_objects(bool array, min_size, connectivity=1): connected components (face
    neighbours) with fewer than min_size elements are cleared."""
    from scipy import ndimage
    lab, n = ndimage.label(mask)
    if n == 0:
        return mask.copy()
    sizes = np.bincount(lab.ravel())
    keep = sizes >= min_size
    keep[0] = False
    return keep[lab]


def segmentation_eval(pred_img, blur_img, valid_mask, non_period_mask, alex, lins, l1_thresh=0.15, lpips_thresh=0.3, lpips_layers=1):
    """NPP_segmentation/train.py:337-393.  pred_img, blur_img (H,W,3) in [0,1]; valid_mask, non_period_mask (H,W,1).
    -> dict(non_period_mask_final (H,W,1) int, l1_img, l1_mask, lpips_maps [layers])."""
    from scipy import ndimage
    dev = alex.device
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32)).to(dev)                  # noqa: E731
    valid = t(valid_mask).permute(2, 0, 1)[None]
    npm = t(non_period_mask).permute(2, 0, 1)[None]
    pred_g = rgb_to_grayscale((t(pred_img).permute(2, 0, 1)[None]) * valid * valid)     # :337-339 (multiplied twice)
    blur_g = rgb_to_grayscale((t(blur_img).permute(2, 0, 1)[None]) * valid)            # :342-343
    l1 = torch.clamp(torch.sum(torch.abs(pred_g - blur_g), 1, keepdim=True), min=0, max=0.99)
    l1_mask = l1 < l1_thresh                                                            # :349-351
    _, maps = lpips_alex_spatial(pred_g, blur_g, alex, lins, normalize=True)            # :361-362
    final = None
    for i in range(lpips_layers):
        lp_np = npm * maps[i]                                                           # only the non-periodic candidates
        period_i = (lp_np < lpips_thresh) & l1_mask                                     # :376-380
        non_i = (~period_i)[0, 0].float().cpu().numpy()
        final = non_i if final is None else final + non_i
    final = ndimage.binary_fill_holes(final > 0)                                        # :390-391
    final = remove_small_objects(final[..., None].astype(bool), min_size=500).astype(int)   # :392-393, on the (H,W,1) array
    return dict(non_period_mask_final=final, l1_img=(l1 * valid)[0, 0].cpu().numpy(), l1_mask=l1_mask[0, 0].cpu().numpy(),
                lpips_maps=[(npm * m)[0, 0].cpu().numpy() for m in maps[:lpips_layers]])
