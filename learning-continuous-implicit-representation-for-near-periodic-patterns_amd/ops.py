"""Thin torch-tensor wrappers over the C ABI (include/npp_hip.h).  torch is used for device
memory and streams only; every call goes to libnpp_hip.so on the current HIP stream."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, EmbedCfg, param_layout, NPP_ROW_TILE, NPP_E, NPP_WIDTH  # noqa: F401


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def tune(key, value=-1):
    """npp_tune: choose between kernel forms that compute the same result ("conv_wink", "conv_win", "conv_wstat", "conv_pair", "light_det")
    or the training stash's format ("stash8": 1 = 8-bit stash + fp8 weight-gradient MFMA (default), 0 = the 16-bit stash);
    value < 0 only reads.  -> the previous value.  Both fused-width libraries are set (the trunk kernels live in the first)."""
    from ._lib import FUSED_WIDTHS
    old = None
    for w in FUSED_WIDTHS:
        try:
            r = lib(w).npp_tune(key.encode(), int(value))
        except OSError:
            continue
        check(r, "npp_tune", w)
        old = r if old is None else old
    return old


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object per
    call (5-8 us, x ~45 kernel calls per iteration: it was most of the host enqueue time); the raw accessor is ~0.3 us."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def select_device(device):
    """The torch.device a fit / net / ranker lives on, made the CURRENT device of this host thread.  Every entry point of the
    C ABI launches on a stream of the current device (one process -- or at least one host thread -- per GPU);
    objects built for another GPU must therefore select it before they allocate or launch."""
    dev = torch.device(device)
    if dev.type == "cuda":
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(dev)
    return dev


def check_current(device):
    """Raise (instead of enqueueing GPU-x kernels on GPU-y's stream) when `device` is not the thread's current device."""
    if device.type == "cuda" and device.index != torch.cuda.current_device():
        raise RuntimeError(f"npp_amd: object lives on {device} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"call torch.cuda.set_device({device.index}) in this thread (one process / host thread per GPU)")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _req(t, dtype, name, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class _PinnedRing:
    """Persistent pinned staging buffers for small host -> device transfers (sampler indices, patch centres).
    A pageable .to(device) is a synchronous copy queued behind everything already in the stream, i.e. a device sync
    per call: it serialised the host-side sampler with the previous iteration's kernels (+0.5 ms per iteration);
    tensor.pin_memory() registers fresh host memory on every call (3.4 ms each, measured).  Here: per size class a ring
    of pinned blocks allocated once, each guarded by the event recorded after its last copy."""

    def __init__(self, slots=8):
        import threading
        self.slots, self.rings = slots, {}
        self.lock = threading.Lock()          # (several host threads transfer: the images of a rank searched side by side, run.py)

    def __call__(self, a, device):
        a = np.ascontiguousarray(a)
        if device.type != "cuda":
            return torch.from_numpy(a).to(device)
        with self.lock:
            return self._copy(a, device)

    def _copy(self, a, device):
        cap = 1 << max(12, int(a.nbytes - 1).bit_length())
        ring = self.rings.get((device, cap))
        if ring is None:
            ring = self.rings[(device, cap)] = {"buf": [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(self.slots)],
                                                "ev": [None] * self.slots, "i": 0}
        i = ring["i"]
        ring["i"] = (i + 1) % self.slots
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()                       # long complete unless the host ran > slots transfers ahead
        stage = ring["buf"][i][:a.nbytes]
        stage.numpy()[:] = a.reshape(-1).view(np.uint8)
        out = stage.to(device, non_blocking=True).view(torch.from_numpy(a[:0]).dtype).reshape(a.shape)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["ev"][i] = ev
        return out


_pinned_ring = _PinnedRing()
# Sections that save / reseed / restore the PROCESS-WIDE torch generator (the reference's torch.manual_seed(0) before every candidate,
# search.py:92; fixed-seed stand-in trunks) hold this lock: images searched on several host threads would otherwise read each other's
# generator state.
RNG_LOCK = __import__("threading").RLock()


def h2d(a, device):
    """NumPy array -> device tensor through a persistent pinned staging ring, non-blocking (see _PinnedRing)."""
    return _pinned_ring(a, device)


PIXEL_LOSS_SCRATCH = 1024 * 8 + 8       # NPP_PIXEL_LOSS_SCRATCH_FLOATS (include/npp_hip.h)
# False: the float-atomic reductions of the pixel loss and the LPIPS head (arrival order; the A/B comparator of the fixed-order forms)
DETERMINISTIC = True


def pad_rows(n):
    return (n + NPP_ROW_TILE - 1) // NPP_ROW_TILE * NPP_ROW_TILE


def selftest(device="cuda"):
    scratch = torch.empty(1 << 18, dtype=torch.float32, device=device)
    check(lib().npp_selftest_mfma(_p(scratch), _stream()), "npp_selftest_mfma")


def embed_fwd(coords, cfg, out_dtype=torch.float32, precise=True):
    """coords (N,2) int32 (row, col) -> (N, K*462); replaces Embedder_periodic.embed +
    Embedder.embed + cat (models/embedder.py:140-148, :51-56; train.py:93-105)."""
    _req(coords, torch.int32, "coords")
    n = coords.shape[0]
    out = torch.empty((n, cfg.K * NPP_E), dtype=out_dtype, device=coords.device)
    code = {torch.float32: 0, torch.bfloat16: 1}[out_dtype]
    check(lib().npp_embed_fwd(_p(coords), n, C.byref(cfg), _p(out), code, int(bool(precise)), _stream()),
          "npp_embed_fwd")
    return out


def warp_fwd(coords, cfg):
    """(N,2) -> (N, K*22) fp32: the Embedder_periodic stage alone (embedder.py:140-148)."""
    _req(coords, torch.int32, "coords")
    n = coords.shape[0]
    out = torch.empty((n, cfg.K * 22), dtype=torch.float32, device=coords.device)
    check(lib().npp_warp_fwd(_p(coords), n, C.byref(cfg), _p(out), _stream()), "npp_warp_fwd")
    return out


def pack_bytes(K, which, width=NPP_WIDTH):
    n = lib(width).npp_pack_bytes(K, width, which)
    check(n, "npp_pack_bytes", width)
    return int(n)


def pack_weights(params, K, wf=None, wb=None, width=NPP_WIDTH):
    _req(params, torch.float32, "params")
    if wf is None:
        wf = torch.empty(pack_bytes(K, 0, width), dtype=torch.uint8, device=params.device)
    if wb is None:
        wb = torch.empty(pack_bytes(K, 1, width), dtype=torch.uint8, device=params.device)
    check(lib(width).npp_pack_weights(_p(params), _p(wf), _p(wb), K, width, _stream()), "npp_pack_weights", width)
    return wf, wb


def train_workspace(K, Bp, ksplit, width=NPP_WIDTH):
    sizes = (C.c_int64 * 4)()
    check(lib(width).npp_train_workspace(K, width, Bp, ksplit, sizes), "npp_train_workspace", width)
    return [int(s) for s in sizes]


def mlp_fwd(coords, cfg, wf, params, pred=None, actF=None, width=NPP_WIDTH, out_act=1):
    """Fused embedder + MLP + sigmoid: coords (Bp,2) -> pred (Bp,3).  Replaces the table
    gather + render() (train.py:166-189; helpers.py:41-62; networks.py:56-95).  out_act: 1 sigmoid, 2 tanh (--normalize_type 2), 0 raw."""
    _req(coords, torch.int32, "coords")
    bp = coords.shape[0]
    if pred is None:
        pred = torch.empty((bp, 3), dtype=torch.float32, device=coords.device)
    if out_act != 1:
        check(lib(width).npp_mlp_fwd_act(_p(coords), bp, C.byref(cfg), width, _p(wf), _p(params), _p(pred), _p(actF), int(out_act), _stream()),
              "npp_mlp_fwd_act", width)
        return pred
    check(lib(width).npp_mlp_fwd(_p(coords), bp, C.byref(cfg), width, _p(wf), _p(params), _p(pred), _p(actF), _stream()),
          "npp_mlp_fwd", width)
    return pred


def mlp_fwd_emb(emb, K, wf, params, out=None, actF=None, out_act=0, width=NPP_WIDTH):
    """NPP_Net.forward(None, x_periodic) on a materialised (Bp, K*462) embedding (networks.py:56-95);
    out_act: 0 raw network output, 1 sigmoid, 2 tanh (render, helpers.py:55-60)."""
    _req(emb, torch.float32, "emb")
    bp = emb.shape[0]
    if out is None:
        out = torch.empty((bp, 3), dtype=torch.float32, device=emb.device)
    check(lib(width).npp_mlp_fwd_emb(_p(emb), emb.stride(0), bp, K, width, _p(wf), _p(params), _p(out), _p(actF), out_act,
                                _stream()), "npp_mlp_fwd_emb", width)
    return out


def mlp_bwd_act(dout, out, K, wb, params, actF, dzF, out_act, width=NPP_WIDTH):
    _req(dout, torch.float32, "dout")
    _req(out, torch.float32, "out", dout.shape)
    check(lib(width).npp_mlp_bwd_act(_p(dout), _p(out), dout.shape[0], K, width, _p(wb), _p(params), _p(actF), _p(dzF),
                                out_act, _stream()), "npp_mlp_bwd_act", width)


def grad_reduce(gslabs, n_slabs, n, grad, accumulate=False):
    """grad (+)= sum of the split-K slabs (the blob-shaped .grad a torch optimiser consumes)."""
    check(lib().npp_grad_reduce(_p(gslabs), n_slabs, gslabs.numel() // n_slabs, n, _p(grad), int(bool(accumulate)), _stream()),
          "npp_grad_reduce")


def fourier_fwd(x, freqs, include_input=True):
    """Embedder.embed (models/embedder.py:11-56): (N,d) -> (N, d*(2*len(freqs)+include_input))."""
    _req(x, torch.float32, "x")
    n, d = x.shape
    f = (C.c_float * len(freqs))(*[float(v) for v in freqs])
    out = torch.empty((n, d * (2 * len(freqs) + int(bool(include_input)))), dtype=torch.float32, device=x.device)
    check(lib().npp_fourier_fwd(_p(x), n, d, f, len(freqs), int(bool(include_input)), _p(out), _stream()), "npp_fourier_fwd")
    return out


def mlp_bwd(dpred, pred, K, wb, params, actF, dzF, width=NPP_WIDTH, out_act=1):
    _req(dpred, torch.float32, "dpred")
    _req(pred, torch.float32, "pred", dpred.shape)
    if out_act != 1:
        return mlp_bwd_act(dpred, pred, K, wb, params, actF, dzF, out_act, width)
    check(lib(width).npp_mlp_bwd(_p(dpred), _p(pred), dpred.shape[0], K, width, _p(wb), _p(params), _p(actF), _p(dzF),
                            _stream()), "npp_mlp_bwd", width)


def mlp_bwd_patch(dpred, pred, K, wb, params, actF, dzF, dx_a, dx_b, fmask, rmask, row0, n_p, k, P, comp, width=NPP_WIDTH, out_act=1):
    """mlp_bwd with the patch rows' dL/dpred formed in the launch (patch_compose_bwd folded in; rows [row0, row0 + n_p P^2) of
    dpred are written)."""
    from ._lib import PatchGrad
    _req(dpred, torch.float32, "dpred")
    _req(pred, torch.float32, "pred", dpred.shape)
    for nm, t in (("dx_a", dx_a), ("dx_b", dx_b)):                  # the leading n_p k images are read (a [x | y] batch gradient is fine)
        if t is not None:
            _req(t, torch.float32, nm)
            if t.dim() != 4 or t.shape[0] < n_p * k or tuple(t.shape[1:]) != (3, P, P):
                raise ValueError(f"{nm}: expected (>= {n_p * k}, 3, {P}, {P}), got {tuple(t.shape)}")
    _req(fmask, torch.float32, "fmask")
    _req(rmask, torch.float32, "rmask")
    if fmask.numel() != n_p * P * P or rmask.numel() != n_p * k * P * P:
        raise ValueError("fmask / rmask: expected (n_p,1,P,P) / (n_p k,1,P,P)")
    pg = PatchGrad(dx_a.data_ptr(), dx_b.data_ptr() if dx_b is not None else None, fmask.data_ptr(), rmask.data_ptr(), int(row0),
                   int(n_p), int(k), int(P), int(bool(comp)))
    if out_act != 1:
        check(lib(width).npp_mlp_bwd_patch_act(_p(dpred), _p(pred), dpred.shape[0], K, width, _p(wb), _p(params), _p(actF), _p(dzF),
                                               C.byref(pg), int(out_act), _stream()), "npp_mlp_bwd_patch_act", width)
        return
    check(lib(width).npp_mlp_bwd_patch(_p(dpred), _p(pred), dpred.shape[0], K, width, _p(wb), _p(params), _p(actF), _p(dzF),
                                       C.byref(pg), _stream()), "npp_mlp_bwd_patch", width)


def auto_ksplit(K, device, width=NPP_WIDTH):
    """Split-K factor that makes the grouped weight-gradient launch (tiles x ksplit workgroups, one per CU) fill the
    chip in exactly one round."""
    tiles = check(lib(width).npp_mlp_wgrad_tiles(K), "npp_mlp_wgrad_tiles", width)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    return max(1, min(64, cus // tiles))


def pack_weights32(params, K, w32=None, width=NPP_WIDTH):
    """fp32 blob -> the fp32 A-operand pack of npp_mlp_fwd32."""
    n = int(lib(width).npp_pack32_bytes(K, width))
    if w32 is None:
        w32 = torch.empty(n, dtype=torch.uint8, device=params.device)
    check(lib(width).npp_pack_weights32(_p(params), _p(w32), K, width, _stream()), "npp_pack_weights32", width)
    return w32


def mlp_fwd32(coords_yx, cfg, w32, params, out=None, out_act=1, width=NPP_WIDTH):
    """Exact-fp32 fused embedder + MLP forward (render), coords (Bp,2) int32 with Bp % 64 == 0 -> (Bp,3)."""
    _req(coords_yx, torch.int32, "coords")
    Bp = coords_yx.shape[0]
    if out is None:
        out = torch.empty((Bp, 3), dtype=torch.float32, device=coords_yx.device)
    check(lib(width).npp_mlp_fwd32(_p(coords_yx), Bp, C.byref(cfg), width, _p(w32), _p(params), _p(out), int(out_act), _stream()),
          "npp_mlp_fwd32", width)
    return out


def mlp_wgrad(dzT, actT, Bp, K, ksplit, gslabs, width=NPP_WIDTH):
    _req(gslabs, torch.float32, "gslabs")
    check(lib(width).npp_mlp_wgrad(_p(dzT), _p(actT), Bp, K, width, ksplit, _p(gslabs), _stream()), "npp_mlp_wgrad", width)


def load_spline(device):
    """values|tangents fp32 on the device + (n_knots, x_scale); table: resources/partition_spline.npz
    (this repo's own derivation, tools/gen_partition_spline.py; distribution.py:129-141)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "partition_spline.npz")
    with np.load(path) as f:
        vals, tans, xs = f["values"].astype(np.float32), f["tangents"].astype(np.float32), float(f["x_scale"])
    t = torch.from_numpy(np.concatenate([vals, tans])).to(device)
    return t, int(vals.shape[0]), xs


QUAD_COEF = {"robust_loss_adaptive": 0.0, "l2": 1.0, "robust_loss": 50.0}


def quad_coef(loss_type):
    """--loss_type (options/arg_config.py:34, models/mse_calculator.py:19-23) -> the `quad` switch of the pixel-loss kernels: 0 = the
    adaptive robust loss; 'l2' = mean(x^2); 'robust_loss' = lossfun(x, alpha = 2, scale = 0.1) = 0.5 (x / 0.1)^2 = 50 x^2."""
    if loss_type not in QUAD_COEF:
        raise ValueError(f"loss_type {loss_type!r}: one of {sorted(QUAD_COEF)}")
    return QUAD_COEF[loss_type]


def pixel_loss_quad(pred, gt, mask, coef, weight, loss, dpred):
    """img2mse(pred, gt, 'l2' | 'robust_loss', None, mask) + backward for one (N, 3) problem or C stacked ones (pred / dpred (C, N, 3),
    gt (N, 3) shared or (C, N, 3), loss (C)); loss is accumulated into."""
    _req(pred, torch.float32, "pred")
    _req(dpred, torch.float32, "dpred", pred.shape)
    nb = 1 if pred.dim() == 2 else pred.shape[0]
    n = pred.shape[-2]
    assert gt.is_contiguous() and gt.shape[-2:] == (n, 3) and loss.numel() >= nb and (mask is None or mask.numel() == n)
    check(lib().npp_pixel_loss_quad(_p(pred), _p(gt), 0 if gt.dim() == 2 else n * 3, _p(mask), n, nb, float(coef), float(weight), _p(loss),
                                    _p(dpred), _stream()), "npp_pixel_loss_quad")


def pixel_loss(pred, gt, mask, latents, spline, n_knots, x_scale, weight, loss, dpred, dlatent, quad=0.0):
    """img2mse(pred, gt, 'robust_loss_adaptive', adaptive_pix, mask) + backward
    (models/mse_calculator.py:13-27).  loss / dlatent are accumulated into (caller zeroes).  quad > 0: the non-adaptive forms."""
    _req(pred, torch.float32, "pred")
    _req(gt, torch.float32, "gt", pred.shape)
    n = pred.shape[0]
    if quad > 0:
        return pixel_loss_quad(pred, gt, mask, quad, weight, loss, dpred)
    check(lib().npp_pixel_loss(_p(pred), _p(gt), _p(mask), n, _p(latents), _p(spline), n_knots, x_scale, weight,
                               _p(loss), _p(dpred), _p(dlatent), _stream()), "npp_pixel_loss")


def adam_step(p, m, v, gslabs, n_slabs, slab_stride, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam step (helpers.py:164) over the flat blob; g = sum of the slabs."""
    _req(p, torch.float32, "p")
    check(lib().npp_adam_step(_p(p), _p(m), _p(v), _p(gslabs), p.numel(), n_slabs, slab_stride, lr, b1, b2, eps,
                              step, _stream()), "npp_adam_step")


def adam_step_net(p, m, v, gslabs, n_slabs, slab_stride, lat, lat_m, lat_v, dlat, zero, lr, step,
                  b1=0.9, b2=0.999, eps=1e-8):
    """optimizer.step() over the network blob and the adaptive-loss latents in one launch; clears the
    latent gradient and `zero` (the next iteration's zero_grad of the small accumulators)."""
    _req(p, torch.float32, "p")
    check(lib().npp_adam_step_net(_p(p), _p(m), _p(v), _p(gslabs), p.numel(), n_slabs, slab_stride, _p(lat), _p(lat_m),
                                  _p(lat_v), _p(dlat), lat.numel(), _p(zero), 0 if zero is None else zero.numel(),
                                  lr, b1, b2, eps, step, _stream()), "npp_adam_step_net")


def adam_step_net_pack(p, m, v, gslabs, n_slabs, slab_stride, lat, lat_m, lat_v, dlat, zero, lr, step, K, wf, wb, width=NPP_WIDTH,
                       b1=0.9, b2=0.999, eps=1e-8, pl_partials=None, loss_cur=None):
    """adam_step_net + pack_weights in one launch: the updated weights are scattered into the bf16 packs wf / wb as well.
    pl_partials / loss_cur: the per-block sums the iteration's pixel-loss launch left in its scratch are added here, in block order,
    to the latent gradients and to the iteration's loss accumulator (include/npp_hip.h npp_pixel_loss_args.scratch)."""
    _req(p, torch.float32, "p")
    check(lib(width).npp_adam_step_net_pack(_p(p), _p(m), _p(v), _p(gslabs), p.numel(), n_slabs, slab_stride, _p(lat), _p(lat_m),
                                            _p(lat_v), _p(dlat), lat.numel(), _p(zero), 0 if zero is None else zero.numel(),
                                            lr, b1, b2, eps, step, K, width, _p(wf), _p(wb), _p(pl_partials), _p(loss_cur), _stream()),
          "npp_adam_step_net_pack", width)


def patch_gather(img_hwc, mask_hw, centres_yx, P, want_mask=True, out=None):
    """extract_glimpse(mode='nearest', zeros padding) at integer centres
    (utils/extract_glimpse.py:53-79 via models/sampler.py:171-178,284-291).  out = (rgb, mask) buffers with room for at least
    M crops each (contiguous; a stacked fit's slice): written in place, the returned tensors are views of them."""
    _req(img_hwc, torch.float32, "img")
    _req(centres_yx, torch.int32, "centres")
    H, W = img_hwc.shape[:2]
    M = centres_yx.shape[0]
    if out is not None:
        rgb, msk = out[0][:M], out[1][:M]
        _req(rgb, torch.float32, "out rgb", (M, 3, P, P))
        _req(msk, torch.float32, "out mask", (M, 1, P, P))
    else:
        rgb = torch.empty((M, 3, P, P), dtype=torch.float32, device=img_hwc.device)
        msk = torch.empty((M, 1, P, P), dtype=torch.float32, device=img_hwc.device) if want_mask else None
    check(lib().npp_patch_gather(_p(img_hwc), _p(mask_hw), H, W, _p(centres_yx), M, P, _p(rgb), _p(msk), _stream()),
          "npp_patch_gather")
    return rgb, msk


def batch_assemble(i_train, pix, cen, P, bp, img_hwc, pmask_hw=None, out=None):
    """-> (coords int32 (bp,2), gt (n_pix,3), pmask (n_pix,) | None): the input rows of one loop iteration, train.py:166-181.
    out = (coords, gt[, pmask]): written in place (a stacked fit's slices)."""
    n_pix, n_p = pix.shape[0], 0 if cen is None else cen.shape[0]
    H, W = img_hwc.shape[:2]
    dev = img_hwc.device
    pm = None
    if out is not None:
        coords, gt = out[:2]
        pm = out[2] if len(out) > 2 else None
        _req(coords, torch.int32, "out coords", (bp, 2))
        _req(gt, torch.float32, "out gt", (n_pix, 3))
        if pm is not None:
            _req(pm, torch.float32, "out pmask", (n_pix,))
    else:
        coords = torch.empty((bp, 2), dtype=torch.int32, device=dev)
        gt = torch.empty((n_pix, 3), dtype=torch.float32, device=dev)
    if pmask_hw is None:
        pm = None
    elif pm is None:
        pm = torch.empty((n_pix,), dtype=torch.float32, device=dev)
    check(lib().npp_batch_assemble(_p(i_train), i_train.shape[0], _p(pix), n_pix, _p(cen), n_p, P, bp, _p(img_hwc), _p(pmask_hw), H, W,
                                   _p(coords), _p(gt), _p(pm), _stream()), "npp_batch_assemble")
    return coords, gt, pm


_cx_ws = {}


def _cx_workspace(device, nbytes, st):
    """The contextual core's scratch (D, cx, mu, row sums and the last-arriver tickets of its six launches): ONE PER STREAM -- two fits
    of one shape on two streams (bench throughput mode, the LPIPS side stream) would otherwise race on the matrices and the tickets."""
    key = (device, int(nbytes), st.value)
    ws = _cx_ws.get(key)
    if ws is None:
        ws = _cx_ws[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return ws


def cx_fwd_bwd(fx, fy, band_width=0.5, weight=None, scale=1.0, loss=None, want_grad=True):
    """contextual_loss(x, y, band_width, weight, 'cosine') on features (N,C,h,w) + dL/dx
    (externel_lib/contextual_loss/functional.py:9-63).  Returns (loss tensor[1], dfx or None)."""
    _req(fx, torch.float32, "fx")
    _req(fy, torch.float32, "fy", fx.shape)
    N, C = fx.shape[:2]
    hw = fx.shape[2] * fx.shape[3]
    nbytes = lib().npp_cx_workspace_bytes(N, C, hw)
    check(nbytes, "npp_cx_workspace_bytes")
    st = _stream()
    ws = _cx_workspace(fx.device, nbytes, st)
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float32, device=fx.device)
    dfx = torch.empty_like(fx) if want_grad else None
    check(lib().npp_cx_fwd_bwd(_p(fx), _p(fy), N, C, hw, band_width, _p(weight), scale, _p(loss), _p(dfx), _p(ws),
                               int(nbytes), st), "npp_cx_fwd_bwd")
    return loss, dfx


_lp_ws = {}


def lpips_layer(f0, f1, lin, latents, spline, n_knots, x_scale, scale, loss, df0=None, dlatent=None):
    """One VGG16 tap of LPIPS.forward(use_robust=True) (externel_lib/lpips/lpips.py:99-121,130); latents None: the plain head
    (use_robust=False, lpips.py:108-109) with its gradient df0 (spline / dlatent unused)."""
    if latents is None:
        spline, n_knots, x_scale, dlatent = None, 0, 0.0, None
    _req(f0, torch.float32, "f0")
    _req(f1, torch.float32, "f1", f0.shape)
    N, C = f0.shape[:2]
    hw = f0.shape[2] * f0.shape[3]
    key = (f0.device, C, _stream().value)               # (accumulators of a launch in flight: one workspace per stream)
    ws = _lp_ws.get(key)
    if ws is None and DETERMINISTIC:
        ws = _lp_ws[key] = torch.zeros(int(lib().npp_lpips_workspace_bytes(C)), dtype=torch.uint8, device=f0.device)
    check(lib().npp_lpips_layer(_p(f0), _p(f1), N, C, hw, _p(lin), _p(latents), _p(spline), n_knots, x_scale, scale,
                                _p(loss), _p(df0), _p(dlatent), _p(ws), _stream()), "npp_lpips_layer")


def lpips_layers(f0s, f1s, lins, latents, spline, n_knots, x_scale, scale, loss, df0s=None, dlatents=None, dflats=None):
    """All taps of LPIPS.forward in ONE launch (the heads are independent of each other): lists per tap of the arguments of lpips_layer
    (latents None: the plain head for every tap).  dflats[i] = (flat bf16 tensor from trunk_alloc, N_total[, yact]) takes the place of
    df0s[i] (then None): the tap's gradient goes straight into the trunk's flat layout, gated by [yact > 0] when the tapped layer's
    flat activation is given."""
    from ._lib import LpipsTap
    n = len(f0s)
    N = f0s[0].shape[0]
    arr = (LpipsTap * n)()
    st = _stream()
    for i in range(n):
        f0, f1 = f0s[i], f1s[i]
        _req(f0, torch.float32, "f0")
        _req(f1, torch.float32, "f1", f0.shape)
        Cc = f0.shape[1]
        ws = None
        if DETERMINISTIC:
            key = (f0.device, Cc, st.value, i)             # one workspace per (stream, tap): the taps of the launch run side by side
            ws = _lp_ws.get(key)
            if ws is None:
                ws = _lp_ws[key] = torch.zeros(int(lib().npp_lpips_workspace_bytes(Cc)), dtype=torch.uint8, device=f0.device)
        fl = None if dflats is None else dflats[i]
        arr[i] = LpipsTap(f0.data_ptr(), f1.data_ptr(), Cc, f0.shape[2] * f0.shape[3], lins[i].data_ptr(),
                          None if latents is None else latents[i].data_ptr(),
                          None if (df0s is None or df0s[i] is None) else df0s[i].data_ptr(),
                          None if (dlatents is None or latents is None) else dlatents[i].data_ptr(), None if ws is None else ws.data_ptr(),
                          None if fl is None else fl[0].data_ptr(), 0 if fl is None else int(fl[1]), f0.shape[2], f0.shape[3],
                          None if (fl is None or len(fl) < 3 or fl[2] is None) else fl[2].data_ptr())
    check(lib().npp_lpips_layers(n, arr, N, None if latents is None else _p(spline), n_knots if latents is not None else 0,
                                 x_scale if latents is not None else 0.0, scale, _p(loss), st), "npp_lpips_layers")


def adam_step_dev(p, m, v, gslabs, n_slabs, slab_stride, hp, b1=0.9, b2=0.999, eps=1e-8):
    """Adam step whose step_size / bias correction come from the device tensor hp[0:2]
    (graph-replayable form of adam_step)."""
    check(lib().npp_adam_step_dev(_p(p), _p(m), _p(v), _p(gslabs), p.numel(), n_slabs, slab_stride, b1, b2, eps, _p(hp),
                                  _stream()), "npp_adam_step_dev")


# ---- a11 / a13 trunks: conv3x3 + ReLU / MaxPool2d stacks on flat padded bf16 tensors ----------------
def trunk_nposp(N, H, W):
    return int(check(lib().npp_trunk_nposp(N, H, W), "npp_trunk_nposp"))


def trunk_alloc(N, C, H, W, device):
    """Zero-initialised flat tensor for a logical (N,C,H,W) activation (csrc/npp_conv.hip layout)."""
    nbytes = int(check(lib().npp_trunk_act_bytes(N, C, H, W), "npp_trunk_act_bytes"))
    return torch.zeros(nbytes, dtype=torch.uint8, device=device)


def conv_pack(weight, in_natural=False):
    """torch Conv2d weight (Cout,Cin,3,3) fp32 -> (forward pack, data-gradient pack), bf16 MFMA fragments."""
    _req(weight, torch.float32, "weight")
    cout, cin = weight.shape[:2]
    assert tuple(weight.shape[2:]) == (3, 3)
    nf = int(check(lib().npp_conv_pack_bytes(cin, cout, 0), "npp_conv_pack_bytes"))
    nb = int(check(lib().npp_conv_pack_bytes(cin, cout, 1), "npp_conv_pack_bytes"))
    pf = torch.empty(nf, dtype=torch.uint8, device=weight.device)
    pb = torch.empty(nb, dtype=torch.uint8, device=weight.device)
    check(lib().npp_conv_pack(_p(weight), cin, cout, int(bool(in_natural)), _p(pf), _p(pb), _stream()), "npp_conv_pack")
    return pf, pb


def trunk_image_in(img, scale, shift, x0):
    """(N,3,H,W) fp32 -> flat C=16 tensor of img*scale[c] + shift[c]."""
    _req(img, torch.float32, "img")
    N, c, H, W = img.shape
    assert c == 3
    s = (C.c_float * 3)(*[float(v) for v in scale])
    b = (C.c_float * 3)(*[float(v) for v in shift])
    check(lib().npp_trunk_image_in(_p(img), N, H, W, s, b, _p(x0), _stream()), "npp_trunk_image_in")


def trunk_patch_in(pred_rows, fake, fmask, real, rmask, n_p, k, P, comp, scale, shift, x0, xy=None, zero=None, which=0, loss=None):
    """npp_patch_compose_fwd + npp_trunk_image_in in one launch: [x | y] -> flat trunk input x0 (and fp32 xy when given);
    zero (small fp32 tensor) is cleared on the way.  which: 0 both halves, 1 prediction half only, 2 real half only
    (x0 then holds n_p*k images).  loss = the argument tuple of pixel_loss(): that loss rides in the same launch."""
    if which != 2:
        _req(pred_rows, torch.float32, "pred_rows", (n_p * P * P, 3))
    if which != 1:
        _req(real, torch.float32, "real", (n_p * k, 3, P, P))
    _req(rmask, torch.float32, "rmask", (n_p * k, 1, P, P))
    if comp and which != 2:
        _req(fake, torch.float32, "fake", (n_p, 3, P, P))
        _req(fmask, torch.float32, "fmask", (n_p, 1, P, P))
    if xy is not None:
        _req(xy, torch.float32, "xy", (2 * n_p * k, 3, P, P))
    s = (C.c_float * 3)(*[float(v) for v in scale])
    b = (C.c_float * 3)(*[float(v) for v in shift])
    if loss is not None:
        from ._lib import PixelLossArgs
        pred, gt, mask, latents, spline, n_knots, x_scale, weight, loss_buf, dpred, dlatent = loss[:11]
        scratch = loss[11] if len(loss) > 11 else None        # PIXEL_LOSS_SCRATCH floats: fixed-order sums (include/npp_hip.h)
        quad = float(loss[12]) if len(loss) > 12 else 0.0     # > 0: --loss_type l2 / robust_loss (quad_coef)
        _req(pred, torch.float32, "pred")
        _req(gt, torch.float32, "gt", pred.shape)
        la = PixelLossArgs(pred.data_ptr(), gt.data_ptr(), None if mask is None else mask.data_ptr(), pred.shape[0], latents.data_ptr(),
                           spline.data_ptr(), int(n_knots), float(x_scale), float(weight), loss_buf.data_ptr(), dpred.data_ptr(),
                           dlatent.data_ptr(), None if scratch is None else scratch.data_ptr(), quad)
        check(lib().npp_trunk_patch_in_loss(_p(pred_rows), _p(fake), _p(fmask), _p(real), _p(rmask), n_p, k, P, int(bool(comp)), s, b,
                                            _p(x0), _p(xy), _p(zero), 0 if zero is None else zero.numel(), int(which), C.byref(la),
                                            _stream()), "npp_trunk_patch_in_loss")
        return
    check(lib().npp_trunk_patch_in(_p(pred_rows), _p(fake), _p(fmask), _p(real), _p(rmask), n_p, k, P, int(bool(comp)), s, b,
                                   _p(x0), _p(xy), _p(zero), 0 if zero is None else zero.numel(), int(which), _stream()),
          "npp_trunk_patch_in")


def conv3x3(x, N_total, n_run, H, W, cin, cout, pack, bias, mode, mask, y, tap=None, ctap=0, tap_scale=None, next_pack=None):
    """next_pack: the weight pack of the launch that follows (its lines are requested into L2 by this one)."""
    ts = None if tap_scale is None else (C.c_float * len(tap_scale))(*[float(v) for v in tap_scale])
    if next_pack is not None:
        check(lib().npp_conv3x3_pf(_p(x), N_total, n_run, H, W, cin, cout, _p(pack), _p(bias), mode, _p(mask), _p(y), _p(tap),
                                   ctap, ts, _p(next_pack), next_pack.numel() * next_pack.element_size(), _stream()), "npp_conv3x3_pf")
        return
    check(lib().npp_conv3x3(_p(x), N_total, n_run, H, W, cin, cout, _p(pack), _p(bias), mode, _p(mask), _p(y), _p(tap),
                            ctap, ts, _stream()), "npp_conv3x3")


def conv3x3_pool(x, N_total, n_run, H, W, cin, cout, pack, bias, y, ypool, tap=None, ctap=0, tap_scale=None, next_pack=None):
    """Forward layer + the 2 x 2 max-pool that follows it in one launch (y: the layer's output, ypool: the pooled tensor)."""
    ts = None if tap_scale is None else (C.c_float * len(tap_scale))(*[float(v) for v in tap_scale])
    nb = 0 if next_pack is None else next_pack.numel() * next_pack.element_size()
    check(lib().npp_conv3x3_pool(_p(x), N_total, n_run, H, W, cin, cout, _p(pack), _p(bias), _p(y), _p(ypool), _p(tap), ctap, ts,
                                 _p(next_pack), nb, _stream()), "npp_conv3x3_pool")


def conv_pair_fwd_ok(H, W, cin, cmid, cout):
    """Whether the fused conv a -> conv b -> pool launch is built for this shape (and switched on: tune("conv_pair"))."""
    return bool(lib().npp_conv_pair_fwd_ok(H, W, cin, cmid, cout))


def conv_pair_fwd(x, N_total, n_run, n_keep, H, W, cin, cmid, cout, pack_a, bias_a, pack_b, bias_b, y_a, y_b, y_pool, tap_b=None):
    """relu(conv a) -> relu(conv b) -> MaxPool2d(2,2) in one launch; y_a / y_b only for the first n_keep images."""
    check(lib().npp_conv_pair_fwd(_p(x), N_total, n_run, n_keep, H, W, cin, cmid, cout, _p(pack_a), _p(bias_a), _p(pack_b), _p(bias_b),
                                  _p(y_a), _p(y_b), _p(y_pool), _p(tap_b), _stream()), "npp_conv_pair_fwd")


def _pixel_loss_args(loss):
    """The argument tuple of pixel_loss() (NPPNet.pixel_loss_args) as the C struct npp_pixel_loss_args."""
    from ._lib import PixelLossArgs
    pred, gt, mask, latents, spline, n_knots, x_scale, weight, loss_buf, dpred, dlatent = loss[:11]
    scratch = loss[11] if len(loss) > 11 else None
    quad = float(loss[12]) if len(loss) > 12 else 0.0
    _req(pred, torch.float32, "pred")
    _req(gt, torch.float32, "gt", pred.shape)
    return PixelLossArgs(pred.data_ptr(), gt.data_ptr(), None if mask is None else mask.data_ptr(), pred.shape[0], latents.data_ptr(),
                         spline.data_ptr(), int(n_knots), float(x_scale), float(weight), loss_buf.data_ptr(), dpred.data_ptr(),
                         dlatent.data_ptr(), None if scratch is None else scratch.data_ptr(), quad)


def conv_pair_fwd_patch(pred_rows, fake, fmask, real, rmask, n_p, k, P, comp, scale, shift, zero, loss, n_keep, cmid, cout,
                        pack_a, bias_a, pack_b, bias_b, y_a, y_b, y_pool, tap_b=None):
    """trunk_patch_in(loss=...) + conv_pair_fwd of the first block in one launch (no flat input tensor, no fp32 batch copy)."""
    _req(pred_rows, torch.float32, "pred_rows", (n_p * P * P, 3))
    _req(real, torch.float32, "real", (n_p * k, 3, P, P))
    _req(rmask, torch.float32, "rmask", (n_p * k, 1, P, P))
    if comp:
        _req(fake, torch.float32, "fake", (n_p, 3, P, P))
        _req(fmask, torch.float32, "fmask", (n_p, 1, P, P))
    s = (C.c_float * 3)(*[float(v) for v in scale])
    b = (C.c_float * 3)(*[float(v) for v in shift])
    la = None if loss is None else _pixel_loss_args(loss)
    check(lib().npp_conv_pair_fwd_patch(_p(pred_rows), _p(fake), _p(fmask), _p(real), _p(rmask), n_p, k, P, int(bool(comp)), s, b, _p(zero),
                                        0 if zero is None else zero.numel(), None if la is None else C.byref(la), n_keep, cmid, cout,
                                        _p(pack_a), _p(bias_a), _p(pack_b), _p(bias_b), _p(y_a), _p(y_b), _p(y_pool), _p(tap_b), _stream()),
          "npp_conv_pair_fwd_patch")


def conv_pair_dgrad(dz_b, N_total, n_run, H, W, cmid, pack_b_bwd, y_a, pack_a_bwd, dimg, scale):
    """dL/d(pre-activation of conv b) -> dL/dimage through the first block in one launch (conv b dgrad, conv a's ReLU gate, conv a dgrad)."""
    ts = (C.c_float * 3)(*[float(v) for v in scale])
    check(lib().npp_conv_pair_dgrad(_p(dz_b), N_total, n_run, H, W, cmid, _p(pack_b_bwd), _p(y_a), _p(pack_a_bwd), _p(dimg), ts, _stream()),
          "npp_conv_pair_dgrad")


def conv3x3_dgrad_pool(x, N_total, n_run, H, W, cin, cout, pack, xpre, addend, dz, next_pack=None):
    """Data gradient of a convolution that reads a pooled tensor + the pool's backward + the pre-pool ReLU gate (+ tap gradient)
    in one launch; H, W: the pooled geometry, xpre / addend / dz: the pre-pool layer's flat tensors."""
    nb = 0 if next_pack is None else next_pack.numel() * next_pack.element_size()
    check(lib().npp_conv3x3_dgrad_pool(_p(x), N_total, n_run, H, W, cin, cout, _p(pack), _p(xpre), _p(addend), _p(dz),
                                       _p(next_pack), nb, _stream()), "npp_conv3x3_dgrad_pool")


def maxpool2_fwd(x, N, H, W, c, y):
    check(lib().npp_maxpool2_fwd(_p(x), N, H, W, c, _p(y), _stream()), "npp_maxpool2_fwd")


def maxpool2_bwd(dy, x, addend, N_total, n_run, H, W, c, dz):
    check(lib().npp_maxpool2_bwd(_p(dy), _p(x), _p(addend), N_total, n_run, H, W, c, _p(dz), _stream()), "npp_maxpool2_bwd")


def trunk_grad_in(df, y, N_total, n_run, c, H, W, dz, as_f16=False, accumulate=False, next_pack=None):
    _req(df, torch.float32, "df", (n_run, c, H, W))
    if next_pack is not None:
        check(lib().npp_trunk_grad_in_pf(_p(df), _p(y), N_total, n_run, c, H, W, _p(dz), int(bool(as_f16)), int(bool(accumulate)),
                                         _p(next_pack), next_pack.numel() * next_pack.element_size(), _stream()), "npp_trunk_grad_in_pf")
        return
    check(lib().npp_trunk_grad_in(_p(df), _p(y), N_total, n_run, c, H, W, _p(dz), int(bool(as_f16)), int(bool(accumulate)), _stream()),
          "npp_trunk_grad_in")


def trunk_export(act, N_total, n_run, c, H, W, is_f16=False):
    out = torch.empty((n_run, c, H, W), dtype=torch.float32, device=act.device)
    check(lib().npp_trunk_export(_p(act), N_total, n_run, c, H, W, _p(out), int(bool(is_f16)), _stream()), "npp_trunk_export")
    return out


# ---- a10: patch plumbing (train.py:200-236) -----------------------------------------------------------
def patch_compose_fwd(pred_rows, fake, fmask, real, rmask, n_p, k, P, comp, xy=None):
    """-> xy (2*n_p*k, 3, P, P) = [x | y] (see include/npp_hip.h)."""
    _req(pred_rows, torch.float32, "pred_rows", (n_p * P * P, 3))
    _req(real, torch.float32, "real", (n_p * k, 3, P, P))
    _req(rmask, torch.float32, "rmask", (n_p * k, 1, P, P))
    if comp:
        _req(fake, torch.float32, "fake", (n_p, 3, P, P))
        _req(fmask, torch.float32, "fmask", (n_p, 1, P, P))
    if xy is None:
        xy = torch.empty((2 * n_p * k, 3, P, P), dtype=torch.float32, device=pred_rows.device)
    check(lib().npp_patch_compose_fwd(_p(pred_rows), _p(fake), _p(fmask), _p(real), _p(rmask), n_p, k, P, int(bool(comp)),
                                      _p(xy), _stream()), "npp_patch_compose_fwd")
    return xy


def patch_compose_bwd(dx_a, dx_b, fmask, rmask, n_p, k, P, comp, dpred_rows):
    _req(dpred_rows, torch.float32, "dpred_rows", (n_p * P * P, 3))
    check(lib().npp_patch_compose_bwd(_p(dx_a), _p(dx_b), _p(fmask), _p(rmask), n_p, k, P, int(bool(comp)), _p(dpred_rows),
                                      _stream()), "npp_patch_compose_bwd")


# ---- generic dense layers (exact fp32): F.linear + snake and their autograd ----------------
def linear_fwd(x, w, b, act, y, z=None):
    """y = act(x w^T + b) on row-major 2-D tensors; x / y / z may be column blocks of wider buffers (stride(0) = ld)."""
    B, cin = x.shape
    cout = w.shape[0]
    assert w.shape[1] == cin and y.shape == (B, cout) and x.stride(1) == 1 and y.stride(1) == 1 and w.is_contiguous()
    check(lib().npp_linear_fwd(_p(x), x.stride(0), _p(w), _p(b), B, cin, cout, act, _p(y), y.stride(0), _p(z),
                               0 if z is None else z.stride(0), _stream()), "npp_linear_fwd")
    return y


def linear_bwd_data(dz, w, dx, in_used=None, accumulate=False):
    """dx (+)= dz w.  `w` may be a column slice w_full[:, a:b] of the (out, in) matrix (gradient w.r.t. one block of a
    concatenated input, networks.py:71,76,85)."""
    B, cout = dz.shape
    cin = w.stride(0)                                    # leading dimension of the full matrix
    in_used = w.shape[1] if in_used is None else in_used
    assert dx.shape == (B, in_used) and dz.stride(1) == 1 and dx.stride(1) == 1 and w.stride(1) == 1
    check(lib().npp_linear_bwd_data(_p(dz), dz.stride(0), _p(w), B, cin, cout, _p(dx), dx.stride(0), in_used, int(bool(accumulate)),
                                    _stream()), "npp_linear_bwd_data")
    return dx


def linear_bwd_weight(dz, x, dw, db=None, accumulate=False):
    B, cout = dz.shape
    cin = x.shape[1]
    assert dw.shape == (cout, cin) and dw.is_contiguous() and x.stride(1) == 1 and dz.stride(1) == 1
    check(lib().npp_linear_bwd_weight(_p(dz), dz.stride(0), _p(x), x.stride(0), B, cin, cout, _p(dw), _p(db), int(bool(accumulate)),
                                      _stream()), "npp_linear_bwd_weight")


# ---- the same layers over several independent problems of one shape in one launch (light.NPPNetLightBatch) -----------------
def linear_fwd_batched(x, w, b, act, y, z=None):
    """y[c] = act(x[c] w[c]^T + b[c]): x (C,B,in), w (C,out,in), b (C,out), y / z (C,B,out); any batch strides, rows may be column
    blocks of wider buffers.  x may be shared by the problems (stride(0) == 0)."""
    C, B, cin = x.shape
    cout = w.shape[1]
    assert w.shape == (C, cout, cin) and y.shape == (C, B, cout) and b.shape == (C, cout)
    assert x.stride(2) == 1 and y.stride(2) == 1 and w.stride(2) == 1 and w.stride(1) == cin and b.stride(1) == 1
    assert z is None or (z.shape == y.shape and z.stride(2) == 1)
    check(lib().npp_linear_fwd_batched(_p(x), x.stride(1), x.stride(0), _p(w), w.stride(0), _p(b), b.stride(0), C, B, cin, cout, act, _p(y),
                                       y.stride(1), y.stride(0), _p(z), 0 if z is None else z.stride(1), 0 if z is None else z.stride(0),
                                       _stream()), "npp_linear_fwd_batched")
    return y


def linear_bwd_data_batched(dz, w, dx, in_used=None, accumulate=False, zy=None, act=0):
    """dx[c] (+)= dz[c] w[c] (first in_used input columns); with zy (C,B,in_used) and act: dx[c] = (dz[c] w[c]) * act'(zy[c]) -- the
    activation backward of the layer below folded into this launch (act / zy as act_bwd takes them)."""
    C, B, cout = dz.shape
    cin = w.shape[2]
    in_used = cin if in_used is None else in_used
    assert w.shape == (C, cout, cin) and dx.shape == (C, B, in_used) and dz.stride(2) == 1 and dx.stride(2) == 1
    assert w.stride(2) == 1 and w.stride(1) == cin
    assert zy is None or (zy.shape == dx.shape and zy.stride(2) == 1)
    check(lib().npp_linear_bwd_data_batched(_p(dz), dz.stride(1), dz.stride(0), _p(w), w.stride(0), C, B, cin, cout, _p(dx), dx.stride(1),
                                            dx.stride(0), in_used, int(bool(accumulate)), _p(zy), 0 if zy is None else zy.stride(1),
                                            0 if zy is None else zy.stride(0), int(act), _stream()), "npp_linear_bwd_data_batched")
    return dx


def linear_bwd_weight_batched(dz, x, dw, db):
    """dw[c] += dz[c]^T x[c], db[c] += column sums of dz[c] (ACCUMULATES: the contraction is split; clear the gradients first)."""
    C, B, cout = dz.shape
    cin = x.shape[2]
    assert dw.shape == (C, cout, cin) and dw.stride(2) == 1 and dw.stride(1) == cin and db.shape == (C, cout) and db.stride(1) == 1
    assert x.shape[:2] == (C, B) and x.stride(2) == 1 and dz.stride(2) == 1
    check(lib().npp_linear_bwd_weight_batched(_p(dz), dz.stride(1), dz.stride(0), _p(x), x.stride(1), x.stride(0), C, B, cin, cout, _p(dw),
                                              dw.stride(0), _p(db), db.stride(0), _stream()), "npp_linear_bwd_weight_batched")


def pixel_loss_batched(pred, gt, latents, spline, n_knots, x_scale, weight, loss, dpred, dlatent):
    """pixel_loss() for C problems: pred / dpred (C,N,3) contiguous, latents / dlatent (C,6), loss (C); gt (N,3) shared or (C,N,3)."""
    _req(pred, torch.float32, "pred")
    C, n = pred.shape[:2]
    assert pred.is_contiguous() and dpred.is_contiguous() and dpred.shape == pred.shape and gt.is_contiguous()
    assert latents.shape == (C, 6) and dlatent.shape == (C, 6) and loss.numel() == C and latents.is_contiguous() and dlatent.is_contiguous()
    assert gt.shape in ((n, 3), (C, n, 3))
    check(lib().npp_pixel_loss_batched(_p(pred), _p(gt), 0 if gt.dim() == 2 else n * 3, n, C, _p(latents), _p(spline), n_knots, x_scale,
                                       weight, _p(loss), _p(dpred), _p(dlatent), _stream()), "npp_pixel_loss_batched")


def linear_bwd_weight_strided(dz, x, dw, db, feature_major_dz, feature_major_x, x_snake=False):
    """dw[c] += dz[c]^T x[c], db[c] += column sums of dz[c], for operands that are row-major (C, B, n) or feature-major (C, n, B) --
    the stashes of the fused NPP_Net_light chains (x_snake: x holds pre-activations, the layer input is snake(x)).  dw (C, out, ld >= in)
    views of a blob, db (C, out)."""
    C = dz.shape[0]
    B, cout = (dz.shape[2], dz.shape[1]) if feature_major_dz else (dz.shape[1], dz.shape[2])
    cin = dw.shape[2]
    assert dz.stride(2) == 1 and x.stride(2) == 1 and dw.stride(2) == 1 and dw.shape[:2] == (C, cout) and db.shape == (C, cout) and db.stride(1) == 1
    assert (x.shape[1] >= cin and x.shape[2] == B) if feature_major_x else (x.shape[1] == B and x.shape[2] >= cin)
    dz_sr, dz_so = (1, dz.stride(1)) if feature_major_dz else (dz.stride(1), 1)
    x_sr, x_si = (1, x.stride(1)) if feature_major_x else (x.stride(1), 1)
    check(lib().npp_linear_bwd_weight_strided(_p(dz), dz_sr, dz_so, dz.stride(0), _p(x), x_sr, x_si, x.stride(0), int(bool(x_snake)), C, B, cin, cout, _p(dw),
                                              dw.stride(1), dw.stride(0), _p(db), db.stride(0), _stream()), "npp_linear_bwd_weight_strided")


def light_pack(desc, params, pack):
    """MFMA-ordered weight copies of C stacked NPP_Net_light blobs (params (C, n) -> pack (C, npp_light_pack_floats()))."""
    import ctypes
    C = params.shape[0]
    assert params.stride(1) == 1 and pack.stride(1) == 1 and pack.shape[0] == C
    check(lib().npp_light_pack(ctypes.byref(desc), _p(params), params.stride(0), C, _p(pack), pack.stride(0), _stream()), "npp_light_pack")


def light_fwd(desc, params, pack, x_per, x_pos, stash, pred, idx=None):
    """Fused NPP_Net_light forward of C candidates: x_per (C, n, 20), x_pos (n, 42) -> pred (C, B, 3), stash (C, rows, B); batch row r is
    table row idx[r] (idx: B int64 on the device) or r itself (idx None, n == B).
    Multi-image form (x_pos (C, n, 42), idx (C, B)): every candidate is another image's fit with its own tables and pixel rows."""
    import ctypes
    C, n = x_per.shape[:2]
    B = pred.shape[1]
    assert x_per.is_contiguous() and x_pos.is_contiguous() and stash.is_contiguous() and pred.is_contiguous()
    if x_pos.dim() == 3:
        assert x_per.shape == (C, n, 20) and x_pos.shape == (C, n, 42) and pred.shape == (C, B, 3) and stash.shape[0] == C and stash.shape[2] == B
        assert idx is not None and idx.dtype == torch.int64 and idx.is_contiguous() and idx.shape == (C, B)
        check(lib().npp_light_fwd_multi(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(x_per), _p(x_pos), _p(idx), n, C, B,
                                        _p(stash), _p(pred), _stream()), "npp_light_fwd_multi")
        return
    assert x_per.shape == (C, n, 20) and x_pos.shape == (n, 42) and pred.shape == (C, B, 3) and stash.shape[0] == C and stash.shape[2] == B
    assert (idx is None and n == B) or (idx is not None and idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == B)
    check(lib().npp_light_fwd(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(x_per), _p(x_pos), _p(idx), n, C, B,
                              _p(stash), _p(pred), _stream()), "npp_light_fwd")


def light_adam_pack(desc, params, m, v, grad, n, pack, lat, lat_m, lat_v, dlat, zero, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """Adam over C stacked blobs (C, stride) and their latents (C, 6) + gradient / loss-word clear + re-pack, one launch."""
    import ctypes
    C = params.shape[0]
    assert all(t.shape == params.shape and t.stride() == params.stride() for t in (m, v, grad)) and params.stride(1) == 1
    assert all(t.shape == (C, 6) and t.is_contiguous() for t in (lat, lat_m, lat_v, dlat)) and (zero is None or (zero.numel() == C and zero.is_contiguous()))
    check(lib().npp_light_adam_pack(ctypes.byref(desc), _p(params), _p(m), _p(v), _p(grad), params.stride(0), n, C, _p(pack), pack.stride(0), _p(lat),
                                    _p(lat_m), _p(lat_v), _p(dlat), _p(zero), lr, b1, b2, eps, step, _stream()), "npp_light_adam_pack")


def light_part_blocks(C, B):
    return int(check(lib().npp_light_part_blocks(C, B), "npp_light_part_blocks"))


def light_bwd_det(desc, params, pack, stash, pred, draw, dstash, gt, lat, spline, n_knots, x_scale, part):
    """light_bwd with the adaptive pixel loss folded in, its per-block sums to part (C, light_part_blocks(C, B), 8) by plain stores."""
    import ctypes
    C, B = pred.shape[:2]
    assert all(t.is_contiguous() for t in (stash, pred, draw, dstash, part, gt, lat)) and part.shape == (C, light_part_blocks(C, B), 8)
    if gt.dim() == 3:                                  # multi-image form: targets per candidate
        assert gt.shape == (C, B, 3)
        check(lib().npp_light_bwd_det_multi(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(stash), _p(pred), _p(gt),
                                            _p(lat), _p(spline), n_knots, x_scale, _p(part), C, B, _p(draw), _p(dstash), _stream()),
              "npp_light_bwd_det_multi")
        return
    check(lib().npp_light_bwd_det(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(stash), _p(pred), _p(gt), _p(lat),
                                  _p(spline), n_knots, x_scale, _p(part), C, B, _p(draw), _p(dstash), _stream()), "npp_light_bwd_det")


def light_adam_pack_det(desc, params, m, v, grad, n, pack, lat, lat_m, lat_v, dlat, zero, lr, step, part, loss_cur, b1=0.9, b2=0.999, eps=1e-8):
    """light_adam_pack after light_bwd_det: the blocks' sums are added in block order (latent gradients, loss_cur (C))."""
    import ctypes
    C = params.shape[0]
    assert part.is_contiguous() and part.shape[0] == C and part.shape[2] == 8
    check(lib().npp_light_adam_pack_det(ctypes.byref(desc), _p(params), _p(m), _p(v), _p(grad), params.stride(0), n, C, _p(pack), pack.stride(0), _p(lat),
                                        _p(lat_m), _p(lat_v), _p(dlat), _p(zero), lr, b1, b2, eps, step, _p(part), part.shape[1], _p(loss_cur), _stream()),
          "npp_light_adam_pack_det")


def light_wgrad(desc, stash, dstash, grad, scratch=None):
    """All seven weight / bias gradients of the C candidates in one launch: grad (C, n) += ... (clear first).
    scratch: light_wgrad_det_scratch(C, B, device) -- the ordered-split form (npp_light_wgrad_det: chip-filling AND bit-reproducible)."""
    import ctypes
    C, B = stash.shape[0], stash.shape[2]
    assert stash.is_contiguous() and dstash.is_contiguous() and grad.stride(1) == 1 and grad.shape[0] == C
    if scratch is not None:
        check(lib().npp_light_wgrad_det(ctypes.byref(desc), _p(stash), _p(dstash), C, B, _p(grad), grad.stride(0), _p(scratch),
                                        scratch.numel() * 4, _stream()), "npp_light_wgrad_det")
        return
    check(lib().npp_light_wgrad(ctypes.byref(desc), _p(stash), _p(dstash), C, B, _p(grad), grad.stride(0), _stream()), "npp_light_wgrad")


def light_wgrad_det_scratch(C, B, device):
    """Zeroed scratch of npp_light_wgrad_det for C candidates of B rows (tickets + partial tiles; the tickets reset themselves)."""
    n = int(lib().npp_light_wgrad_det_scratch_bytes(int(C), int(B)))
    return torch.zeros(n // 4, dtype=torch.float32, device=device)


def light_bwd(desc, params, pack, stash, pred, dpred, draw, dstash, loss_args=None):
    """Fused data-gradient chain: dpred (C, B, 3) -> draw (C, B, 3), dstash (C, rows, B).  loss_args = (gt (B, 3), latents (C, 6), spline,
    n_knots, x_scale, loss (C), dlatent (C, 6)): the adaptive robust pixel loss is evaluated inside the launch instead (dpred unused)."""
    import ctypes
    C, B = pred.shape[:2]
    assert all(t.is_contiguous() for t in (stash, pred, draw, dstash)) and dstash.shape[0] == C and dstash.shape[2] == B
    if loss_args is None:
        assert dpred.is_contiguous()
        gt = lat = spl = loss = dlat = None
        nk, xs = 0, 0.0
    else:
        gt, lat, spl, nk, xs, loss, dlat = loss_args
        assert gt.shape == (B, 3) and gt.is_contiguous() and lat.shape == (C, 6) and lat.is_contiguous() and dlat.shape == (C, 6) and dlat.is_contiguous()
        assert loss.numel() == C and loss.is_contiguous()
    check(lib().npp_light_bwd(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(stash), _p(pred), _p(dpred), _p(gt),
                              _p(lat), _p(spl), nk, xs, _p(loss), _p(dlat), C, B, _p(draw), _p(dstash), _stream()), "npp_light_bwd")


# ---- f1 on the 16-bit matrix pipe (csrc/npp_light16.hip) ------------------------------------------------------------------------
def light16_sizes(B):
    """(pack bytes, forward-stash bytes, gradient-stash bytes) per candidate for batches of B rows (a multiple of 64)."""
    L = lib()
    return int(L.npp_light16_pack_bytes()), int(L.npp_light16_stash_bytes(B, 0)), int(L.npp_light16_stash_bytes(B, 1))


def _bytes2d(t, name):
    assert t.dtype == torch.uint8 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 16 == 0 and t.data_ptr() % 16 == 0, name


def light16_pack(desc, params, pack):
    """bf16 MFMA-ordered copies (forward + transposed) of C stacked NPP_Net_light blobs: params (C, n) fp32 -> pack (C, bytes) uint8."""
    import ctypes
    C = params.shape[0]
    _bytes2d(pack, "pack")
    assert params.stride(1) == 1 and pack.shape[0] == C
    check(lib().npp_light16_pack(ctypes.byref(desc), _p(params), params.stride(0), C, _p(pack), pack.stride(0), _stream()), "npp_light16_pack")


def light16_fwd(desc, params, pack, x_per, x_pos, actF, pred, idx=None):
    """light_fwd on bf16 operands: pred (C, B, 3) fp32 and the W-format forward stash actF (C, bytes) uint8."""
    import ctypes
    C, n = x_per.shape[:2]
    B = pred.shape[1]
    _bytes2d(pack, "pack")
    _bytes2d(actF, "actF")
    assert x_per.is_contiguous() and x_pos.is_contiguous() and pred.is_contiguous()
    if x_pos.dim() == 3:                                   # multi-image set: per candidate its own positional table and pixel rows
        assert x_per.shape == (C, n, 20) and x_pos.shape == (C, n, 42) and pred.shape == (C, B, 3) and actF.shape[0] == C
        assert idx is not None and idx.dtype == torch.int64 and idx.is_contiguous() and idx.shape == (C, B)
        check(lib().npp_light16_fwd_multi(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(x_per), _p(x_pos), _p(idx), n,
                                          C, B, _p(actF), actF.stride(0), _p(pred), _stream()), "npp_light16_fwd_multi")
        return
    assert x_per.shape == (C, n, 20) and x_pos.shape == (n, 42) and pred.shape == (C, B, 3) and actF.shape[0] == C
    assert (idx is None and n == B) or (idx is not None and idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == B)
    check(lib().npp_light16_fwd(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(x_per), _p(x_pos), _p(idx), n, C, B,
                                _p(actF), actF.stride(0), _p(pred), _stream()), "npp_light16_fwd")


def light16_bwd(desc, params, pack, actF, pred, dpred, dzF, loss_args=None):
    """light_bwd on bf16 operands: the W-format gradient stash dzF (C, bytes) uint8; loss_args as light_bwd."""
    import ctypes
    C, B = pred.shape[:2]
    _bytes2d(pack, "pack")
    _bytes2d(actF, "actF")
    _bytes2d(dzF, "dzF")
    assert pred.is_contiguous() and dzF.shape[0] == C
    if loss_args is None:
        assert dpred.is_contiguous()
        gt = lat = spl = loss = dlat = None
        nk, xs = 0, 0.0
    else:
        gt, lat, spl, nk, xs, loss, dlat = loss_args
        assert gt.shape == (B, 3) and gt.is_contiguous() and lat.shape == (C, 6) and lat.is_contiguous() and dlat.shape == (C, 6) and dlat.is_contiguous()
        assert loss.numel() == C and loss.is_contiguous()
    check(lib().npp_light16_bwd(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(actF), actF.stride(0), _p(pred),
                                _p(dpred), _p(gt), _p(lat), _p(spl), nk, xs, _p(loss), _p(dlat), C, B, _p(dzF), dzF.stride(0), _stream()),
          "npp_light16_bwd")


def light16_bwd_det(desc, params, pack, actF, pred, dzF, gt, lat, spline, n_knots, x_scale, part):
    """light16_bwd with the pixel loss folded in and the blocks' loss / latent-gradient sums left in part (C, B / 64, 8) by plain stores
    (added in block order by light16_adam_pack_det): bit-reproducible.  gt (B, 3) shared, or (C, B, 3) per candidate (multi-image set)."""
    import ctypes
    C, B = pred.shape[:2]
    _bytes2d(pack, "pack")
    _bytes2d(actF, "actF")
    _bytes2d(dzF, "dzF")
    multi = gt.dim() == 3
    assert pred.is_contiguous() and dzF.shape[0] == C and gt.is_contiguous() and gt.shape == ((C, B, 3) if multi else (B, 3))
    assert lat.shape == (C, 6) and lat.is_contiguous() and part.is_contiguous() and part.shape == (C, B // 64, 8) and part.dtype == torch.float32
    check(lib().npp_light16_bwd_det(ctypes.byref(desc), _p(params), params.stride(0), _p(pack), pack.stride(0), _p(actF), actF.stride(0), _p(pred),
                                    _p(gt), 3 * B if multi else 0, _p(lat), _p(spline), n_knots, x_scale, _p(part), C, B, _p(dzF), dzF.stride(0),
                                    _stream()), "npp_light16_bwd_det")


def light16_adam_pack_det(desc, params, m, v, n, gslabs, pack, lat, lat_m, lat_v, dlat, zero, lr, step, part, loss_cur, b1=0.9, b2=0.999, eps=1e-8):
    """light16_adam_pack after light16_bwd_det: the blocks' sums of `part` added in block order (latent gradients; loss_cur[c] += loss terms)."""
    import ctypes
    C, ks, ns = gslabs.shape
    _bytes2d(pack, "pack")
    assert params.stride(1) == 1 and params.shape[0] == C and part.is_contiguous() and part.shape[0] == C and part.shape[2] == 8
    check(lib().npp_light16_adam_pack_det(ctypes.byref(desc), _p(params), _p(m), _p(v), params.stride(0), n, C, _p(gslabs), ks, ns, ks * ns, _p(pack),
                                          pack.stride(0), _p(lat), _p(lat_m), _p(lat_v), _p(dlat), _p(zero), lr, b1, b2, eps, step, _p(part),
                                          part.shape[1], _p(loss_cur), _stream()), "npp_light16_adam_pack_det")


def light16_wgrad(desc, actF, dzF, B, gslabs):
    """All seven weight / bias gradients of the C candidates in one launch of the grouped split-K kernel: gslabs (C, ksplit, n) fp32
    receives the ksplit partial sums (plain stores: no clearing needed, bit-reproducible)."""
    import ctypes
    C, ks, n = gslabs.shape
    assert gslabs.is_contiguous() and gslabs.dtype == torch.float32 and actF.shape[0] == C and dzF.shape[0] == C
    check(lib().npp_light16_wgrad(ctypes.byref(desc), _p(actF), actF.stride(0), _p(dzF), dzF.stride(0), C, B, ks, _p(gslabs), n, ks * n, _stream()),
          "npp_light16_wgrad")


def light16_adam_pack(desc, params, m, v, n, gslabs, pack, lat, lat_m, lat_v, dlat, zero, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """Adam over C stacked blobs (gradient = the slabs summed in order) and their latents + loss-word clear + bf16 re-pack, one launch."""
    import ctypes
    C, ks, ns = gslabs.shape
    _bytes2d(pack, "pack")
    assert all(t.shape == params.shape and t.stride() == params.stride() for t in (m, v)) and params.stride(1) == 1 and params.shape[0] == C
    assert all(t.shape == (C, 6) and t.is_contiguous() for t in (lat, lat_m, lat_v, dlat)) and (zero is None or (zero.numel() == C and zero.is_contiguous()))
    check(lib().npp_light16_adam_pack(ctypes.byref(desc), _p(params), _p(m), _p(v), params.stride(0), n, C, _p(gslabs), ks, ns, ks * ns, _p(pack),
                                      pack.stride(0), _p(lat), _p(lat_m), _p(lat_v), _p(dlat), _p(zero), lr, b1, b2, eps, step, _stream()),
          "npp_light16_adam_pack")


def act_bwd(dy, zy, act, dz):
    B, n = dy.shape
    check(lib().npp_act_bwd(_p(dy), dy.stride(0), _p(zy), zy.stride(0), B, n, act, _p(dz), dz.stride(0), _stream()), "npp_act_bwd")
    return dz


def act_fwd(x, act, y):
    check(lib().npp_act_fwd(_p(x), x.numel(), act, _p(y), _stream()), "npp_act_fwd")
    return y


LPIPS_PLAIN_SCRATCH = 264               # NPP_LPIPS_PLAIN_SCRATCH_FLOATS (include/npp_hip.h)


def lpips_plain_layer(f0, f1, lin, scale, out, scratch=None):
    """scratch (LPIPS_PLAIN_SCRATCH zeroed floats, one per tap launched back to back): the fixed-order, bit-reproducible form."""
    _req(f0, torch.float32, "f0")
    _req(f1, torch.float32, "f1", f0.shape)
    N, C = f0.shape[:2]
    if scratch is not None:
        assert scratch.dtype == torch.float32 and scratch.numel() >= LPIPS_PLAIN_SCRATCH and scratch.is_contiguous()
        check(lib().npp_lpips_plain_layer_det(_p(f0), _p(f1), N, C, f0.shape[2] * f0.shape[3], _p(lin), scale, _p(out), _p(scratch), _stream()),
              "npp_lpips_plain_layer_det")
        return
    check(lib().npp_lpips_plain_layer(_p(f0), _p(f1), N, C, f0.shape[2] * f0.shape[3], _p(lin), scale, _p(out), _stream()),
          "npp_lpips_plain_layer")


# ---- around the AlexNet convolutions (segmentation criterion, proposal conv1 features): include/npp_hip.h "f3 / f4 front ends" ----
def im2col(x, k, stride, pad, nhwc=False):
    """Rows (n, oy, ox) x columns (c, ky, kx) of a k x k / stride / zero-pad convolution over x: (N,C,H,W), or (N,H,W,C) with nhwc."""
    _req(x, torch.float32, "x")
    if nhwc:
        N, H, W, Cc = x.shape
    else:
        N, Cc, H, W = x.shape
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cols = torch.empty((N * ho * wo, Cc * k * k), dtype=torch.float32, device=x.device)
    check(lib().npp_im2col(_p(x), N, Cc, H, W, k, stride, pad, 1 if nhwc else 0, _p(cols), _stream()), "npp_im2col")
    return cols, ho, wo


def maxpool_nhwc(x, k, stride):
    _req(x, torch.float32, "x")
    N, H, W, Cc = x.shape
    y = torch.empty((N, (H - k) // stride + 1, (W - k) // stride + 1, Cc), dtype=torch.float32, device=x.device)
    check(lib().npp_maxpool_nhwc(_p(x), N, H, W, Cc, k, stride, _p(y), _stream()), "npp_maxpool_nhwc")
    return y


def lpips_spatial_layer(f0, f1, lin):
    """(N,h,w,C) position-major features of the two images -> the tap's distance map (N,h,w) (lpips.py:99-110, spatial, plain head)."""
    _req(f0, torch.float32, "f0")
    _req(f1, torch.float32, "f1", f0.shape)
    _req(lin, torch.float32, "lin", (f0.shape[-1],))
    d = torch.empty(f0.shape[:-1], dtype=torch.float32, device=f0.device)
    check(lib().npp_lpips_spatial_layer(_p(f0), _p(f1), d.numel(), f0.shape[-1], _p(lin), _p(d), _stream()), "npp_lpips_spatial_layer")
    return d


def resize_bilinear(x, H, W, out=None, accumulate=False):
    """F.interpolate(x[:, None], size=(H, W), mode='bilinear', align_corners=False)[:, 0] of maps (N,h,w); accumulate: out += it."""
    _req(x, torch.float32, "x")
    N, h, w = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    else:
        _req(out, torch.float32, "out", (N, H, W))
    check(lib().npp_resize_bilinear(_p(x), N, h, w, H, W, 1 if accumulate else 0, _p(out), _stream()), "npp_resize_bilinear")
    return out


# ---- remapping variant: Gram-matrix style loss pieces (models/style_loss.py:37-74) ----------------------------
_gram_ws = {}


def gram_fwd(f):
    """Gram matrices (N, C, C) of features (N, C, h, w) (models/style_loss.py:55-58).  ops.DETERMINISTIC: the ordered-split form
    (npp_gram_fwd_det) over a zeroed per-stream scratch."""
    _req(f, torch.float32, "f")
    N, Cc = f.shape[:2]
    hw = f.shape[2] * f.shape[3]
    g = torch.empty((N, Cc, Cc), dtype=torch.float32, device=f.device)
    if DETERMINISTIC:
        key = (f.device, N, Cc, hw, _stream().value)
        ws = _gram_ws.get(key)
        if ws is None:
            ws = _gram_ws[key] = torch.zeros(int(lib().npp_gram_fwd_det_scratch_bytes(N, Cc, hw)) // 4, dtype=torch.float32, device=f.device)
        check(lib().npp_gram_fwd_det(_p(f), N, Cc, hw, _p(g), _p(ws), ws.numel() * 4, _stream()), "npp_gram_fwd_det")
        return g
    check(lib().npp_gram_fwd(_p(f), N, Cc, hw, _p(g), _stream()), "npp_gram_fwd")
    return g


def gram_bwd(dg, f):
    df = torch.empty_like(f)
    N, Cc = f.shape[:2]
    check(lib().npp_gram_bwd(_p(dg), _p(f), N, Cc, f.shape[2] * f.shape[3], _p(df), _stream()), "npp_gram_bwd")
    return df


_re_ws = {}


def robust_elem(a, b, latents, spline, n_knots, x_scale, coef_n, loss, want_grad=True, dlatent=None):
    """per-element adaptive robust NLL of (a - b) over (N, D); returns d(loss)/d(a) (N, D) or None."""
    N, D = a.shape
    key = (a.device, D, _stream().value)                 # (it holds the launch's partial sums and arrival ticket: one per stream)
    ws = _re_ws.get(key)
    if ws is None:
        ws = _re_ws[key] = torch.empty(int(lib().npp_robust_elem_workspace_bytes(D)), dtype=torch.uint8, device=a.device)
    diff = torch.empty_like(a)
    dd = torch.empty_like(a) if want_grad else None
    cf = (C.c_float * N)(*[float(v) for v in coef_n])
    check(lib().npp_robust_elem(_p(a), _p(b), N, D, _p(latents), _p(spline), n_knots, x_scale, cf, _p(loss), _p(diff), _p(dd),
                                _p(dlatent) if want_grad else None, _p(ws), _stream()), "npp_robust_elem")
    return dd


# ---- stacked launches (include/npp_hip.h "stacked launches"): M images per launch -----------------------------------------
def embed_dev_blob(cfgs, device):
    """The fused kernels' embedder constants of M images as one device blob (npp_embed_dev_build per image)."""
    nb = lib().npp_embed_dev_bytes()
    host = np.zeros((len(cfgs), nb), np.uint8)
    for i, cfg in enumerate(cfgs):
        check(lib().npp_embed_dev_build(C.byref(cfg), host[i].ctypes.data_as(C.c_void_p)), "npp_embed_dev_build")
    return torch.from_numpy(host).to(device)


def mlp_fwd_stack(coords, edev, M, K, wf, params, pred, actF, it, width=NPP_WIDTH):
    """coords (M,Bp,2) int32, wf (M, pack bytes) uint8, params (M, stride) fp32, pred (M,Bp,3), actF (M, bytes) uint8."""
    _req(coords, torch.int32, "coords")
    Bp = coords.shape[1]
    check(lib(width).npp_mlp_fwd_stack(_p(coords), Bp, _p(edev), M, K, width, _p(wf), wf.stride(0), _p(params), params.stride(0), _p(pred),
                                       _p(actF), actF.stride(0), _p(it), _stream()), "npp_mlp_fwd_stack", width)


def trunk_patch_in_loss_stack(pred, row0, crops, cmasks, M, n_p, P, X, N_total, scale, shift, x0, xy, zero, it, loss, gt_stride,
                              lat_stride, loss_stride):
    from ._lib import PixelLossArgs
    s = (C.c_float * 3)(*[float(v) for v in scale])
    b = (C.c_float * 3)(*[float(v) for v in shift])
    pr, gt, mask, latents, spline, n_knots, x_scale, weight, loss_buf, dpred, dlatent, n_rows, scratch = loss[:13]
    la = PixelLossArgs(pr.data_ptr(), gt.data_ptr(), None if mask is None else mask.data_ptr(), int(n_rows), latents.data_ptr(),
                       spline.data_ptr(), int(n_knots), float(x_scale), float(weight), loss_buf.data_ptr(), dpred.data_ptr(),
                       dlatent.data_ptr(), scratch.data_ptr(), float(loss[13]) if len(loss) > 13 else 0.0)
    check(lib().npp_trunk_patch_in_loss_stack(_p(pred), pred.shape[1], row0, _p(crops), crops.stride(0), _p(cmasks), cmasks.stride(0), M,
                                              n_p, P, X, N_total, s, b, _p(x0), _p(xy), 0 if xy is None else xy.stride(0), _p(zero),
                                              _p(it), C.byref(la), gt_stride, lat_stride, loss_stride, scratch.stride(0), _stream()),
          "npp_trunk_patch_in_loss_stack")


def cx_fwd_bwd_groups(fx, fy, it, M, band_width, scale, loss, loss_stride=1):
    """npp_cx_fwd_bwd over sample groups (one per image of a stack): fx / fy (X, C, h, w); loss (M * loss_stride,) accumulated."""
    _req(fx, torch.float32, "fx")
    _req(fy, torch.float32, "fy", fx.shape)
    N, Cc = fx.shape[:2]
    hw = fx.shape[2] * fx.shape[3]
    nbytes = lib().npp_cx_workspace_bytes(N, Cc, hw)
    check(nbytes, "npp_cx_workspace_bytes")
    st = _stream()
    ws = _cx_workspace(fx.device, nbytes, st)
    dfx = torch.empty_like(fx)
    check(lib().npp_cx_fwd_bwd_groups(_p(fx), _p(fy), N, Cc, hw, band_width, scale, _p(loss), loss_stride, _p(dfx), _p(it), M, _p(ws),
                                      int(nbytes), st), "npp_cx_fwd_bwd_groups")
    return dfx


def cx_fwd_bwd_flat(fx, fy, yact, dz, N_total, band_width, scale, loss, loss_stride=1, it=None, M=0):
    """The contextual core with dL/dx written straight into the trunk's flat bf16 gradient tensor dz, gated by [yact > 0] (yact: the
    tapped layer's flat fp16 output; both of geometry N_total x C x H x W) -- no fp32 gradient tensor, no npp_trunk_grad_in launch.
    it / M: sample groups of a stacked launch (cx_fwd_bwd_groups)."""
    _req(fx, torch.float32, "fx")
    _req(fy, torch.float32, "fy", fx.shape)
    N, Cc, H, W = fx.shape
    nbytes = lib().npp_cx_workspace_bytes(N, Cc, H * W)
    check(nbytes, "npp_cx_workspace_bytes")
    st = _stream()
    ws = _cx_workspace(fx.device, nbytes, st)
    key2 = (fx.device, "dxh", N * Cc * H * W, st.value)
    dxh = _cx_ws.get(key2)
    if dxh is None:
        dxh = _cx_ws[key2] = torch.empty(N * Cc * H * W, dtype=torch.float32, device=fx.device)
    check(lib().npp_cx_fwd_bwd_flat(_p(fx), _p(fy), N, Cc, H, W, band_width, scale, _p(loss), loss_stride, _p(dxh), _p(yact), _p(dz),
                                    N_total, _p(it), M, _p(ws), int(nbytes), st), "npp_cx_fwd_bwd_flat")


def mlp_bwd_patch_stack(dpred, pred, M, K, wb, params, actF, dzF, dx_a, dx_b, cmasks, row0, n_p, P, it, width=NPP_WIDTH):
    check(lib(width).npp_mlp_bwd_patch_stack(_p(dpred), _p(pred), pred.shape[1], M, K, width, _p(wb), wb.stride(0), _p(params),
                                             params.stride(0), _p(actF), actF.stride(0), _p(dzF), dzF.stride(0), _p(dx_a), _p(dx_b),
                                             0 if dx_b is None else dx_b.stride(0), _p(cmasks), cmasks.stride(0), row0, n_p, P, _p(it),
                                             _stream()), "npp_mlp_bwd_patch_stack", width)


def mlp_wgrad_stack(dzF, actF, Bp, M, K, ksplit, gslabs, it, width=NPP_WIDTH):
    check(lib(width).npp_mlp_wgrad_stack(_p(dzF), dzF.stride(0), _p(actF), actF.stride(0), Bp, M, K, width, ksplit, _p(gslabs),
                                         gslabs.stride(0), _p(it), _stream()), "npp_mlp_wgrad_stack", width)


def adam_step_net_pack_stack(p, m, v, n, gslabs, n_slabs, slab_stride, lat, lat_m, lat_v, dlat, n_lat, zero, M, K, wf, wb, it,
                             pl_partials, loss_cur, width=NPP_WIDTH, b1=0.9, b2=0.999, eps=1e-8):
    """p / m / v (M, stride) blobs, gslabs (M, n_slabs * slab_stride), lat.. (M, >= n_lat), zero (M, n_zero): image m's idle accumulators."""
    check(lib(width).npp_adam_step_net_pack_stack(_p(p), _p(m), _p(v), p.stride(0), _p(gslabs), n, n_slabs, slab_stride, gslabs.stride(0),
                                                  _p(lat), _p(lat_m), _p(lat_v), _p(dlat), n_lat, lat.stride(0), _p(zero),
                                                  zero.shape[1], zero.stride(0), b1, b2, eps, M, K, width, _p(wf), wf.stride(0), _p(wb),
                                                  wb.stride(0), _p(pl_partials), pl_partials.stride(0), _p(loss_cur), _p(it), _stream()),
          "npp_adam_step_net_pack_stack", width)
