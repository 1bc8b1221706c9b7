"""The per-image optimisation loop of NPP_completion/train.py:133-337, host side.

One CompletionFit owns one image on one GPU: the masked input, the train/val pixel
split (loaders/loaders.py:107-108), the NPPNet state and the NumPy RNG whose call order
follows the reference (np.random.choice over i_train per iteration, train.py:172).  All
arithmetic of an iteration runs in libnpp_hip.so; this file samples indices, gathers the
ground-truth colours (torch indexing: glue) and sequences the kernels.
"""
import math

import os

import numpy as np
import torch

from . import ops
from .model import NPPNet
from .sampler import GridPatchSampler
from .losses import ContextualLoss, LPIPS


class CompletionFit:
    def __init__(self, img, mask, angles_deg, periods, freqs, params, device="cuda", N_rand=8192,
                 ksplit=None, seed=0, lrate=5e-4, lrate_decay=500, valid_mask=None, shifts=None,
                 patch_size=None, patch_num=2, num_real_patch_per_sample=3, invalid_ratio=0.3,
                 contextual_weight=1e-3, perceptual_weight=1e-3, use_comp=True, patch_size_decay=2000,
                 vgg19_state_dict=None, vgg16_state_dict=None, lpips_lin_weights=None, rng_mode="reference",
                 prefetch=0, use_perceptual_loss=True, task="completion", clear_mask=None, style_weight=None,
                 vgg16_style_state_dict=None, masked_img=None, width=256, no_reg_sampling=False, use_patch_weight=False,
                 no_pix_loss=False, use_contextual_loss=True, loss_type="robust_loss_adaptive", use_adaptive_perceptual_loss=True, normalize_type=1):
        """img (H,W,3) float in [0,1]; mask (H,W,1) 1 = known (loaders.py:92-101).
        normalize_type: --normalize_type (arg_config.py:31): 1 = sigmoid output, 2 = tanh output (helpers.py:55-58).  The reference
        rescales ONLY its evaluation image to [-1, 1] under 2 (loaders.py:56,111); the loop still trains on masked_img in [0, 1]
        (train.py:173) -- reproduced as it is: pass `img` already rescaled if the evaluation should see it that way.
        width: --netwidth, 256 (BASELINE configs) or 512 (the reference's default, arg_config.py:57); `params` must match.
        use_adaptive_perceptual_loss: False = LPIPS(use_robust=False) in the loop (arg_config.py:78, train.py:241-246).
        loss_type: --loss_type (arg_config.py:34; models/mse_calculator.py:19-23): 'robust_loss_adaptive' | 'l2' | 'robust_loss'.
        Ablation switches of arg_config.py:78-92: no_reg_sampling (random real patches), use_patch_weight (1/d lattice weights:
        weighted-sum forms of the contextual and LPIPS terms, train.py:224-250), no_pix_loss (:197), use_contextual_loss.
        masked_img = img * mask is what the loop trains on (train.py:173).
        rng_mode: "reference" (default) keeps the reference's NumPy random stream call by call
        (np.random.uniform, np.random.choice(replace=False) for the patch centres and the N_rand pixel rows:
        train.py:172, sampler.py:260,324) -- each choice permutes its whole population, 2.7 ms of host time per
        iteration at 512^2, three times the device time of the iteration.  The stream comes from the library's own
        MT19937 (host_rng.NativeRandomState: bit-identical to np.random.RandomState(seed), GIL-free); "numpy" uses NumPy's
        generator itself.  "fast" draws the same uniform without-replacement samples with np.random.Generator.choice
        (O(size)): same distribution, different stream.
        prefetch: > 0 runs the host half of the sampler (draw_batch) that many iterations ahead on a producer thread --
        it never reads network state, so the stream and the results are unchanged; with the native generator the draws
        overlap the training loop instead of preceding it."""
        if rng_mode not in ("reference", "numpy", "fast"):
            raise ValueError("rng_mode must be 'reference', 'numpy' or 'fast'")
        if task not in ("completion", "remapping", "segmentation"):
            raise ValueError("task must be 'completion', 'remapping' or 'segmentation'")
        # False: the folded launches as separate ones (npp_pixel_loss, npp_patch_compose_bwd: the comparator of
        # tests/test_gpu_parity.py::test_folded_launches_equal_the_separate_ones).  Measured and removed in round 5 (they are in the
        # history of rounds 3-4, profiles/r03_rejected_experiments.txt #2 #5): the pixel rows as a row group of their own on a side
        # stream (0.742 -> 0.762 ms) and the next iteration's real-patch trunk pass prefetched on a side stream (0.767 -> 0.776 ms).
        self.fold_launches = True
        img = np.asarray(img, np.float32)
        mask = np.asarray(mask, np.float32).reshape(img.shape[0], img.shape[1], 1)
        self.H, self.W = img.shape[:2]
        self.device = ops.select_device(device)
        valid = np.ones_like(mask) if valid_mask is None else np.asarray(valid_mask, np.float32).reshape(mask.shape)
        self.task = task
        if task in ("completion", "segmentation"):
            # segmentation (NPP_segmentation/train.py:62-157): the same loop with the initial PERIODIC region as the known mask
            # and the masked-blurred image as the image trained and sampled on -- the caller passes them as `mask` and
            # `masked_img` (io.load_npp_segmentation)
            mask = mask * valid
            # loaders.py:107-108: np.nonzero order (row-major) for both splits
            self.i_train = np.stack(np.nonzero(mask[..., 0] * valid[..., 0]), 1).astype(np.int32)
            self.i_val = np.stack(np.nonzero((1 - mask[..., 0]) * valid[..., 0]), 1).astype(np.int32)
            pixel_mask = None                                        # gt_mask = ones (train.py:176)
            # masked_img is what the loop trains and samples patches on (train.py:173; the sampler gets masked_img): the file
            # as loaded when the caller has it (it need not equal gt * mask, e.g. inputs without a clean ground truth)
            train_img = img * mask if masked_img is None else np.asarray(masked_img, np.float32).reshape(img.shape)
        else:
            # NPP_remapping: the whole valid image is trained on (loaders.py:279), the 'val' pool and the sampler mask are the
            # CLEAR (non-blurry) region (:280; NPP_remapping/train.py:147-155), and the pixel loss weighs blurry pixels 0.3
            # through gt_mask = clear_mask (train.py:203; models/mse_calculator.py:17)
            if clear_mask is None:
                raise ValueError("task='remapping' needs clear_mask (H,W[,1]): 1 = sharp region (NPP_remapping/blur_detection.py)")
            mask = np.asarray(clear_mask, np.float32).reshape(mask.shape) * valid
            self.i_train = np.stack(np.nonzero(valid[..., 0]), 1).astype(np.int32)
            self.i_val = np.stack(np.nonzero(mask[..., 0] * valid[..., 0]), 1).astype(np.int32)
            pixel_mask = mask
            train_img = img
        self.pix_w = 0.0 if no_pix_loss else 1.0                      # train.py:197-198 `loss = 0`: the pixel term weighs nothing
        self.use_patch_weight, self.use_contextual_loss = bool(use_patch_weight), bool(use_contextual_loss)
        self.lp_robust = bool(use_adaptive_perceptual_loss)
        self.img = torch.from_numpy(img).to(self.device)
        self.mask = torch.from_numpy(mask).to(self.device)
        self.masked_img = torch.from_numpy(np.ascontiguousarray(train_img, np.float32)).to(self.device).contiguous()
        self.pixel_mask = None if pixel_mask is None else torch.from_numpy(pixel_mask[..., 0].copy()).to(self.device)
        self.net = NPPNet(angles_deg, periods, freqs, (self.H, self.W), params=params, device=self.device,
                          ksplit=ksplit, lrate=lrate, lrate_decay=lrate_decay, width=width, loss_type=loss_type,
                          out_act=int(normalize_type))
        if task == "segmentation":
            self.net.lr_clock = False                              # NPP_segmentation/train.py:408 (see NPPNet.lr_clock)
        self.N_rand = int(min(N_rand, self.i_train.shape[0]))
        if rng_mode == "reference":
            from .host_rng import NativeRandomState
            self.rng = NativeRandomState(seed)
        else:
            self.rng = np.random.RandomState(seed)
        self.fast_rng = np.random.default_rng(seed) if rng_mode == "fast" else None
        self._prefetch, self._producer, self._queue, self._stop = int(prefetch), None, None, False
        self._ahead, self._s_smp, self.side_sampler = None, None, True   # device half of the sampler one iteration ahead (step_full)
        self._draw_iter = 0                                       # iterations drawn so far (== self.iteration without prefetch)
        self.i_train_dev = torch.from_numpy(self.i_train).to(self.device)
        yy, xx = np.meshgrid(np.arange(self.H, dtype=np.int32), np.arange(self.W, dtype=np.int32), indexing="ij")
        self.i_all_dev = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(self.device)
        self.iteration = 0
        # ---- patch losses (train.py:50-51,126-130): only when the periodicity shifts are given
        self.patch_sampler = None
        if shifts is not None:
            self.patch_size = int(patch_size) if patch_size else int(np.clip(max(np.asarray(periods).reshape(-1, 2)[0])
                                                                  + (32 - max(np.asarray(periods).reshape(-1, 2)[0]) % 32), 64, 160))
            self.patch_num = int(patch_num)
            self.topk, self.invalid_ratio = int(num_real_patch_per_sample), float(invalid_ratio)
            self.cx_w, self.lp_w, self.use_comp = float(contextual_weight), float(perceptual_weight), bool(use_comp)
            self.use_perceptual_loss = bool(use_perceptual_loss)             # options/arg_config.py use_perceptual_loss
            self.patch_size_decay = int(patch_size_decay)
            self.patch_sampler = GridPatchSampler(
                img=self.masked_img[None], mask=self.mask[None], N_samples=self.patch_num, patch_size=self.patch_size,
                height=self.H, width=self.W, pool_train=self.i_train, pool_val=self.i_val, selected_shifts=shifts,
                no_reg_sampling=bool(no_reg_sampling), rng=self.rng, fast_rng=self.fast_rng)
            self.contextualLoss = ContextualLoss(use_vgg=True, vgg_state_dict=vgg19_state_dict, device=self.device).to(self.device)
            self.percepLoss = LPIPS(net="vgg", lin_weights=lpips_lin_weights, vgg_state_dict=vgg16_state_dict,
                                    device=self.device)
            self.style, self.style_w = None, 0.0
            if style_weight is not None or task == "remapping":
                from .losses import StyleLoss
                self.style = StyleLoss(vgg_state_dict=vgg16_style_state_dict, device=self.device)
                self.style_w = 1.0 if style_weight is None else float(style_weight)        # arg_config.py: style_weight 1
            self.last_source, self.skipped = None, 0
            self._xy, self._xy_key, self._xy_bufs = None, None, {}
            # the LPIPS branch of a 'same' iteration as ONE captured HIP graph (lpips_branch); lp_graph = False keeps the launches
            self._lp_graphs, self.lp_graph = {}, True
            self._s_lp = torch.cuda.Stream(self.device)
            self.patch_loss_buf = torch.zeros(1, dtype=torch.float32, device=self.device)

    def lpips_branch(self, xy, nk, scale, loss_buf):
        """percepLoss.fused(xy, nk, scale, loss_buf) on the CURRENT stream (the loop's side stream).  The branch is 46 launches at the
        latency floor -- VGG16 on four 96 x 96 patches, 5 heads, the data-gradient pass -- and its 0.39 ms of host enqueue time made the
        'same' iterations host-bound (0.71 ms of enqueue against 0.60 ms of device time for the other sources).  From its third use on a
        given set of buffers it is therefore replayed as ONE captured HIP graph (torch.cuda.CUDAGraph over the same launches: arguments
        and buffers are fixed -- xy / loss_buf / latents are persistent tensors, scale and n_p k part of the key)."""
        lp = self.percepLoss
        # (every address and scalar a captured launch holds is part of the key: re-homed latents or buffers get a capture of their own)
        # -- incl. the stream (the heads' accumulators are per stream) and the trunk's choice of kernel forms
        key = (xy.data_ptr(), loss_buf.data_ptr(), int(nk), float(scale), self.lp_robust, lp._lat.data_ptr(), lp._dlat.data_ptr(),
               lp.lins[0].data_ptr(), lp.grouped_heads, lp.flat_tap_grads, int(lp.hip_trunk.fuse_pairs), lp.hip_trunk.fold_pool_bwd,
               lp.hip_trunk.fold_pool_fwd, ops._stream().value)
        ent = self._lp_graphs.get(key) if self.lp_graph else None
        if ent is not None and ent[0] is not None:
            ent[0].replay()
            lp.touched = lp.touched or self.lp_robust
            return ent[1]
        out = lp.fused(xy, nk, scale, loss_buf, normalize=True, use_robust=self.lp_robust)
        if not self.lp_graph:
            return out
        uses = 1 if ent is None else ent[2] + 1
        if ent is None and len(self._lp_graphs) >= 8:          # each capture keeps a private pool (taps + gradients): bound their number
            self._lp_graphs.pop(next(iter(self._lp_graphs)))
        self._lp_graphs[key] = (None, None, uses)
        if uses == 2:                                           # two eager passes have warmed every buffer / workspace: capture for the next use
            s = torch.cuda.current_stream(self.device)
            g = torch.cuda.CUDAGraph()
            try:                                                # (a capture only RECORDS the launches: nothing runs, no state changes)
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    gout = lp.fused(xy, nk, scale, loss_buf, normalize=True, use_robust=self.lp_robust)
            except Exception as e:                             # a runtime that cannot capture: stay eager, say so once
                import warnings
                warnings.warn(f"npp_amd.fit: LPIPS branch not captured as a HIP graph ({e}); keeping the launch-by-launch form")
                self.lp_graph = False
                return out
            self._lp_graphs[key] = (g, gout, uses)
        return out

    # ---- sampling (train.py:172-181) -------------------------------------------------
    def draw_pixels(self):
        """np.random.choice(n_train, N_rand, replace=False) (train.py:172): host indices."""
        if self.fast_rng is not None:
            return self.fast_rng.choice(self.i_train.shape[0], size=self.N_rand, replace=False)
        return self.rng.choice(self.i_train.shape[0], size=[self.N_rand], replace=False)

    def sample_pixels(self):
        """-> coords (N_rand,2) on device."""
        return self.i_train_dev[ops.h2d(self.draw_pixels(), self.device)]

    def gather_gt(self, coords):
        return self.masked_img[coords[:, 0].long(), coords[:, 1].long()].contiguous()

    # ---- one optimisation iteration (pixel loss; patch losses are added by the caller) ----
    def step(self, coords=None, extra_coords=None, patch_loss=None):
        """coords: pixel-loss rows (sampled when None).  extra_coords: rows of the predicted
        patches appended after them (train.py:181); patch_loss(pred_patch_rows, dpred_patch_rows)
        must fill dL/dpred for those rows and return nothing (its kernels accumulate)."""
        net = self.net
        if coords is None:
            coords = self.sample_pixels()
        n_pix = coords.shape[0]
        allc = coords if extra_coords is None else torch.cat([coords, extra_coords], 0)
        n = allc.shape[0]
        bp = ops.pad_rows(n)
        if bp != n:
            allc = torch.cat([allc, allc.new_zeros((bp - n, 2))], 0)
        ws = net.workspace(bp)
        net.zero_grad()                                  # optimizer.zero_grad() (train.py:192)
        pred = net.forward_train(allc.contiguous())
        if n_pix < bp:
            ws["dpred"][n_pix:].zero_()
        gt = self.gather_gt(coords)
        net.pixel_loss(bp, n_pix, gt)                    # img2mse(pred[:N_rand], gt, ...) (train.py:195)
        if patch_loss is not None:
            patch_loss(pred[n_pix:n], ws["dpred"][n_pix:n])
        net.backward(bp)                                 # loss.backward()
        net.optimizer_step(bp)                           # optimizer.step() + LR rule + global_step
        self.iteration += 1
        return net.loss_buf

    # ---- one iteration of the complete loop body, train.py:133-264 ------------------------
    def step_full(self):
        """Patch sampling -> pixel sampling -> fused forward -> pixel loss + contextual loss
        (+ LPIPS when patch_source == 'same') -> backward -> Adam.  Returns False when the
        sampler found no valid real patch: the iteration is skipped BEFORE zero_grad and
        global_step is not advanced (train.py:160-161)."""
        assert self.patch_sampler is not None, "construct CompletionFit with shifts=... for the patch losses"
        if self._prefetch > 0:
            if self._producer is None:
                self._start_producer()
            d = self._queue.get()
            if isinstance(d, BaseException):
                raise d
        else:
            d = self.draw_batch()
        if self._prefetch > 0 and self.side_sampler:
            # With the producer thread the draws run ahead anyway; the sampler's DEVICE half (one upload, the crop gather, the row
            # assembly: ~35 us of launches) then runs ahead too, on a side stream under the previous iteration's kernels instead of
            # in front of this one's (round 5: end to end 0.683 -> 0.65 ms at c2).  It reads constants only (image, mask, pools).
            cur = self._ahead if self._ahead is not None else self._materialise_ahead(d)
            nxt_d = d if self._ahead is not None else self._next_draw()
            self._ahead = None                                    # consumed: a step that raises must not be replayed by the next call
            d, batch, ev = cur
            self.last_draw = d
            self.iteration += 1
            try:
                if batch is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    self.step_from(batch)
                else:
                    self.skipped += 1
            finally:
                # the draw already taken from the queue belongs to the NEXT iteration whatever happened to this one: a caller that
                # catches the exception and goes on keeps the reference's random stream (as StackedFit.step_full does)
                self._ahead = self._materialise_ahead(nxt_d)      # (behind this iteration's launches in the host's order)
            return batch is not None
        self.last_draw = d                                        # host-side record of this iteration's draws (tests, logging)
        batch = self.materialise_batch(d)
        self.iteration += 1
        if batch is None:
            self.skipped += 1
            return False
        self.step_from(batch)
        return True

    def _next_draw(self):
        d = self._queue.get()
        if isinstance(d, BaseException):
            raise d
        return d

    def _materialise_ahead(self, d):
        """materialise_batch(d) on the sampler's own stream -> (d, batch | None, event); the batch's tensors are handed to the main
        stream (record_stream: the allocator must not recycle them while the iteration still reads them)."""
        if d["k"] == 0:
            return d, None, None
        main = torch.cuda.current_stream(self.device)
        if self._s_smp is None:
            self._s_smp = torch.cuda.Stream(self.device)
            self._s_smp.wait_stream(main)                         # the constructor's uploads
        with torch.cuda.stream(self._s_smp):
            b = self.materialise_batch(d)
            ev = torch.cuda.Event()
            ev.record(self._s_smp)

        def hand_over(v):
            if isinstance(v, torch.Tensor):
                v.record_stream(main)
            elif isinstance(v, dict):
                for x in v.values():
                    hand_over(x)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    hand_over(x)
        hand_over(b)
        return d, b, ev

    def decay_due(self):
        """Whether the NEXT draw_batch() halves the patch size first (train.py:137-141: by iteration index, trange(start=1))."""
        i = self._draw_iter + 1
        return (self.patch_sampler is not None and i % self.patch_size_decay == 0 and i != 1 and self.patch_size > 31
                and getattr(self, "_decayed_at", None) != i)

    def apply_decay(self):
        """The patch-size decay of the next iteration, now (a StackedFit re-forms around the new batch shape before it draws)."""
        self._decayed_at = self._draw_iter + 1
        self.patch_size //= 2
        self.patch_num *= 2
        self.patch_sampler.reset_patchsize(None, None, self.patch_size, self.patch_num)
        self.patch_sampler.reset_pool(self.i_train, self.i_val)

    # ---- host half of one iteration's sampling (no device work: may run ahead on the producer thread) -------------
    def draw_batch(self):
        """train.py:137-141 (patch-size decay, by iteration index), :152-157 (sample_patches) and :172 (pixel draw), in the
        reference's RNG order.  One call per loop iteration, including the ones that end up skipped."""
        if self.decay_due():
            self.apply_decay()
        self._draw_iter += 1
        d = self.patch_sampler.draw(topk=self.topk, invalid_ratio=self.invalid_ratio)
        d["n_p"] = self.patch_num
        if d["k"] > 0:
            d["pix"] = self.draw_pixels()                         # the reference `continue`s before this draw when k == 0
        return d

    def materialise_batch(self, d, out=None):
        """Device half: crops, coordinates, ground-truth colours of a draw_batch().  None when no valid real patch exists.
        out: dict(coords, gt, crops, cmasks[, pmask]) of preallocated buffers (this image's slices of a StackedFit's arrays)."""
        # (want_tuple=False: the loop reads the contiguous crops of last_raw; the reference-shaped views / tiled copies of the
        #  8-tuple would cost two more launches per iteration)
        if d["k"] == 0:
            return None
        # ONE host -> device transfer per iteration: [pixel-row indices (int64) | patch centres (int32)] through one pinned block
        pix = np.ascontiguousarray(d["pix"], np.int64)
        cen = self.patch_sampler.centres_i32(d)
        blob = np.concatenate([pix.view(np.uint8).reshape(-1), cen.view(np.uint8).reshape(-1)])
        blob_dev = ops.h2d(blob, self.device)
        pix_dev = blob_dev[:pix.nbytes].view(torch.int64)
        cen_dev = blob_dev[pix.nbytes:].view(torch.int32).reshape(-1, 2)
        _, _, _, _, _, source, k, weight = self.patch_sampler.materialise(d, want_coords=False, want_tuple=False, cen_dev=cen_dev,
                                                                          crops_out=None if out is None else (out["crops"], out["cmasks"]))
        # coordinates of all rows (N_rand pixel rows, then the fake patches' rows, zero padding) + the pixel rows' colours:
        # one launch (npp_batch_assemble) instead of two index gathers, two concatenations and the colour / mask gathers
        n_pix, P, n_p = d["pix"].shape[0], d["P"], d["cen"].shape[0]
        n = n_pix + n_p * P * P
        bp = ops.pad_rows(n)
        allc, gt, pm = ops.batch_assemble(self.i_train_dev, pix_dev, self.patch_sampler.last_cen_dev, P, bp,
                                          self.masked_img, self.pixel_mask,
                                          out=None if out is None else (out["coords"], out["gt"], out.get("pmask")))
        w_dev = ops.h2d(np.ascontiguousarray(d["weights"], np.float32), self.device) if (self.use_patch_weight and d["weights"] is not None) else None
        return dict(coords=allc, n_pix=n_pix, n=n, bp=bp, gt=gt, source=source, k=k, P=P, n_p=d["n_p"],
                    raw=self.patch_sampler.last_raw, pmask=pm, weight=w_dev)

    def sample_batch(self):
        """Host-side sampling of one iteration + its device half.  None when no valid real patch exists."""
        return self.materialise_batch(self.draw_batch())

    def _start_producer(self):
        import queue
        import threading
        self._queue = queue.Queue(maxsize=self._prefetch)

        def run():
            try:
                while not self._stop:
                    d = self.draw_batch()
                    while not self._stop:
                        try:
                            self._queue.put(d, timeout=0.1)
                            break
                        except queue.Full:
                            continue
            except BaseException as e:                            # surface sampler errors in the training thread
                self._queue.put(e)
        self._producer = threading.Thread(target=run, name="npp-sampler", daemon=True)
        self._producer.start()

    def close(self):
        """Stop the producer thread (if any)."""
        self._stop = True
        if self._producer is not None:
            self._producer.join(timeout=2.0)
            self._producer = None

    def __del__(self):
        self._stop = True

    def step_from(self, b):
        """Device side of one iteration (everything after sampling), train.py:183-264, as explicit kernel
        launches (no autograd): fused forward -> pixel loss -> patch plumbing (npp_patch_compose_fwd) -> VGG19 trunk
        -> contextual loss core -> trunk data-gradient (-> the same through VGG16 / LPIPS head on 'same' iterations)
        -> npp_patch_compose_bwd -> backward chain + wgrad -> Adam."""
        ops.check_current(self.device)
        self.last_source = source = b["source"]
        net, P, n_p, k, n_pix, n, bp = self.net, b["P"], b["n_p"], b["k"], b["n_pix"], b["n"], b["bp"]
        raw = b["raw"]
        comp = self.use_comp and source == "val"                 # train.py:230-231
        nk = n_p * k
        key = (nk, P)
        if self._xy_key != key:                                   # one fp32 batch buffer per (n_p k, P): fixed addresses (see lpips_branch)
            if key not in self._xy_bufs:
                if len(self._xy_bufs) >= 8:                       # (the patch size changes every patch_size_decay iterations)
                    self._xy_bufs.clear()
                    self._lp_graphs.clear()
                self._xy_bufs[key] = torch.empty((2 * nk, 3, P, P), dtype=torch.float32, device=self.device)
            self._xy, self._xy_key = self._xy_bufs[key], key
        with_lp = source == "same" and self.use_perceptual_loss
        cx = self.contextualLoss
        xy = self._xy if (with_lp or self.style is not None) else None      # fp32 batch only when another trunk reads it
        sc, sh = cx.input_norm()
        main = torch.cuda.current_stream(self.device)
        fold = self.fold_launches
        net.zero_grad()
        if self.percepLoss.touched:
            self.percepLoss.zero_latent_grads()
        ws = net.workspace(bp)
        pred = net.forward_train(b["coords"])
        if ws.get("n_rows") != n:                                # rows >= n never receive a gradient
            ws["dpred"][n:].zero_()
            ws["n_rows"] = n
        # The consumers of the prediction are independent and each under-fills the chip (small grids, dependent
        # launches): on 'same' iterations the LPIPS branch runs on a side stream next to the contextual branch and
        # joins before npp_patch_compose_bwd (1.30 -> 1.17 ms).  Measured negative (event record / wait costs more than is
        # hidden): the 12 us pixel loss on a side stream (0.742 -> 0.762 ms per 'val' iteration); the real half of the
        # contextual batch through its own trunk instance on a side stream beside the MLP forward (0.767 -> 0.776 ms).
        # one launch: the adaptive pixel loss of the pixel rows + patch plumbing -> the contextual trunk's flat fp16 input
        # (normalised), the patch-loss accumulator cleared on the way (fold_launches = False: the separate launches, kept as
        # the comparator of tests/test_gpu_parity.py and for A/B timing)
        if not fold:
            net.pixel_loss(bp, n_pix, b["gt"], mask=b.get("pmask"), weight=self.pix_w)
        # Iterations whose other consumers need no fp32 copy of the batch ('val' / 'train' without a style term): the patch plumbing
        # and the pixel loss ride inside the trunk's first launch (ops.conv_pair_fwd_patch) -- no npp_trunk_patch_in launch at all
        x0_src = None
        if fold and xy is None and self.use_contextual_loss and cx.hip_trunk.can_compose_input(P, P):
            x0_src = dict(patch=(pred[n_pix:n], raw["fake"], raw["fmask"], raw["real"], raw["rmask"], n_p, k, P, comp),
                          zero=self.patch_loss_buf, loss=net.pixel_loss_args(bp, n_pix, b["gt"], mask=b.get("pmask"), weight=self.pix_w))
        else:
            ops.trunk_patch_in(pred[n_pix:n], raw["fake"], raw["fmask"], raw["real"], raw["rmask"], n_p, k, P, comp, sc, sh,
                               cx.hip_trunk.input_buffer(2 * nk, P, P), xy, self.patch_loss_buf, which=0,
                               loss=net.pixel_loss_args(bp, n_pix, b["gt"], mask=b.get("pmask"), weight=self.pix_w) if fold else None)
        dx_b = None
        # use_patch_weight (train.py:224-250): contextual term sum_i -log(cx_i w_i + 1e-5) (the core's weighted form), LPIPS term
        # sum_i d_i w_i -- on 'same' iterations the weights are all 1 (sampler.py:338), i.e. nk times the mean
        weight = b.get("weight")
        if with_lp:                                                                                 # train.py:241-250
            self._s_lp.wait_stream(main)
            with torch.cuda.stream(self._s_lp):
                dx_b = self.lpips_branch(xy, nk, self.lp_w * (nk if weight is not None else 1), self.patch_loss_buf)
        cx.hip_trunk.final_next_pack = net.wb        # the backward chain that follows streams this pack: requested into L2 early
        if self.use_contextual_loss:
            dx_a = cx.fused((2 * nk, 3, P, P), nk, self.cx_w, self.patch_loss_buf, weight=weight, x0_ready=True, x0_src=x0_src)   # train.py:238-239
        else:                                                                                       # ablation: no contextual term
            dx_a = torch.zeros((2 * nk, 3, P, P), dtype=torch.float32, device=self.device)
        if dx_b is not None:
            main.wait_stream(self._s_lp)
        if self.style is not None:                                                                  # NPP_remapping/train.py:253-261
            self.style.zero_latent_grads()
            dx_s = self.style.fused(xy, nk, self.style_w, self.patch_loss_buf)
            dx_b = dx_s if dx_b is None else dx_b.add_(dx_s)
        self.last_patch_loss = self.patch_loss_buf
        lr_used = net.lr
        # npp_patch_compose_bwd folded into the backward launch: dL/dpred of the patch rows is formed (and written) there
        if fold:
            net.backward(bp, patch=(dx_a, dx_b, raw["fmask"], raw["rmask"], n_pix, n_p, k, P, comp))
            net.optimizer_step(bp)
        else:
            ops.patch_compose_bwd(dx_a, dx_b, raw["fmask"], raw["rmask"], n_p, k, P, comp, ws["dpred"][n_pix:n])
            net.backward(bp)
            net.optimizer_step(bp)
        if self.percepLoss.touched:                               # only 'same' iterations give them a gradient
            self.percepLoss.adam_step(lr_used)
        if self.style is not None:                                # the style latents are in the same optimiser (helpers.py:153-159)
            self.style.adam_step(lr_used)

    # ---- checkpoint / resume (the reference has none: start = 0, helpers.py:166) ----------------
    def state_dict(self):
        """Everything that determines the continuation of the fit: parameters + Adam moments + step counters of the network and
        of every adaptive-loss latent group, the LR clock, the patch-size schedule and the position of the random streams.
        Host tensors / NumPy arrays; not available while a producer thread is running ahead (prefetch > 0)."""
        if self._producer is not None:
            raise RuntimeError("state_dict() with an active producer thread: the random stream is ahead of the fit; use prefetch=0")
        net = self.net
        sd = {"net": {k: getattr(net, k).detach().cpu().clone() for k in ("params", "m", "v", "latents", "lat_m", "lat_v")},
              "net_steps": (net.global_step, net.opt_step, net.lr), "iteration": self.iteration, "draw_iter": self._draw_iter,
              "skipped": getattr(self, "skipped", 0), "rng": self.rng.get_state(),
              "fast_rng": None if self.fast_rng is None else self.fast_rng.bit_generator.state}
        if self.patch_sampler is not None:
            sd["patch"] = (self.patch_size, self.patch_num)
            lp = self.percepLoss
            sd["lpips"] = {"latents": [t.cpu().clone() for t in lp.latents], "m": [t.cpu().clone() for t in lp.lat_m],
                           "v": [t.cpu().clone() for t in lp.lat_v], "step": lp.lat_step}
            if self.style is not None:
                st = self.style
                sd["style"] = {"latents": [t.cpu().clone() for t in st.latents], "m": [t.cpu().clone() for t in st.lat_m],
                               "v": [t.cpu().clone() for t in st.lat_v], "step": st.lat_step}
        return sd

    def load_state_dict(self, sd):
        if self._producer is not None:
            raise RuntimeError("load_state_dict() with an active producer thread")
        net = self.net
        for k, v in sd["net"].items():
            getattr(net, k).copy_(v.to(self.device))
        net.global_step, net.opt_step, net.lr = sd["net_steps"]
        net.repack()
        net._clean = False
        self.iteration, self._draw_iter, self.skipped = sd["iteration"], sd["draw_iter"], sd["skipped"]
        self.rng.set_state(sd["rng"])
        if self.fast_rng is not None and sd["fast_rng"] is not None:
            self.fast_rng.bit_generator.state = sd["fast_rng"]
        if self.patch_sampler is not None and "patch" in sd:
            if (self.patch_size, self.patch_num) != tuple(sd["patch"]):
                self.patch_size, self.patch_num = sd["patch"]
                self.patch_sampler.reset_patchsize(None, None, self.patch_size, self.patch_num)
                self.patch_sampler.reset_pool(self.i_train, self.i_val)
            for obj, key in ((self.percepLoss, "lpips"), (self.style, "style")):
                if obj is not None and key in sd:
                    for dst, src in zip(obj.latents, sd[key]["latents"]):
                        dst.copy_(src.to(self.device))
                    for dst, src in zip(obj.lat_m, sd[key]["m"]):
                        dst.copy_(src.to(self.device))
                    for dst, src in zip(obj.lat_v, sd[key]["v"]):
                        dst.copy_(src.to(self.device))
                    obj.lat_step = sd[key]["step"]

    # ---- evaluation (train.py:270-331) -----------------------------------------------
    @torch.no_grad()
    def render_image(self):
        """Full H x W grid through the fused forward: the 'fitted pixels/s' pass."""
        return self.net.render(self.i_all_dev).reshape(self.H, self.W, 3)

    def psnr(self, region="known"):
        """-10 log10 MSE over known / unknown pixels against the clean image."""
        pred = self.render_image()
        m = self.mask if region == "known" else (1.0 - self.mask)
        d2 = ((pred - self.img) ** 2) * m
        mse = d2.sum() / (m.sum() * 3).clamp_min(1.0)
        return float(-10.0 * math.log10(max(float(mse), 1e-20)))
