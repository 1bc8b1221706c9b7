"""M independent images per GPU in ONE launch sequence (include/npp_hip.h "stacked launches").

The reference fits its images one after the other (run_completion.sh:8-14: one `python train.py` per input directory) exactly
like it fits its periodicity candidates (NPP_proposal/search.py:85).  The fits are independent -- own weights, own Adam state,
own random stream -- so M of them can share every launch of the loop body (NPP_completion/train.py:133-264): the
image becomes a grid dimension of the fused forward / backward / weight-gradient / Adam launches, and the patch-loss trunks see
M x 12 patches per launch instead of 12.  This is BASELINE config c3's shape at fewer than 8 GPUs (8 / 4 / 2 images per GPU) and
amortises what bounds the single-image iteration: ~34 dependent launches of ~5 us ramp each, trunk layers that fill a quarter of
the chip, 416 workgroups on 512 slots.

StackedFit takes M CompletionFit objects (each keeps its own sampler, random stream, LPIPS latents and evaluation methods),
re-homes their network state into stacked blobs (the fits' tensors become views of them: render / psnr / state_dict keep
working) and replaces CompletionFit.step_full for all of them.  Per image the arithmetic is the single-image path's, launch by
launch; tests/test_gpu_stack.py compares the parameters after 20 iterations with each image's stand-alone fit.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import ops
from ._lib import StackIter


class _Batches(list):
    """The M batches of one stacked iteration + the buffer set (`set`) the sampler filled for them."""
    set = 0


class StackedFit:
    parallel_draws = True       # the images' host draws of one iteration on a thread pool (False: one after the other)

    def __init__(self, fits, ksplit=None):
        if not fits:
            raise ValueError("StackedFit needs at least one CompletionFit")
        f0 = fits[0]
        self.fits, self.M = list(fits), len(fits)
        self.device = f0.device
        ops.check_current(self.device)
        net0 = f0.net
        self.K, self.width = net0.K, net0.width
        for f in fits:
            # (segmentation is the completion loop with other inputs and weights -- NPP_segmentation/train.py:148-286; the remapping
            #  task -- NPP_remapping/train.py:158-300 -- adds per-pixel loss weights, which ride in the stacked pixel-loss launch, and a
            #  style term per image with its own adaptive latents, which runs per image on the side stream like the LPIPS branch)
            if f.patch_sampler is None or f.task != f0.task or (f.style is None) != (f0.style is None) or (
                    (f.pixel_mask is None) != (f0.pixel_mask is None)):
                raise ValueError("StackedFit: fits of ONE task with the patch losses (shifts=...) only")
            if (f.net.K, f.net.width, f.N_rand, f.patch_size, f.patch_num, f.topk, f.device, f.net.quad) != (
                    self.K, self.width, f0.N_rand, f0.patch_size, f0.patch_num, f0.topk, f0.device, net0.quad):
                raise ValueError("StackedFit: the images of a stack share K, width, N_rand, patch size / count, loss_type and the device")
            if f.use_patch_weight or not f.use_contextual_loss or f._prefetch:
                raise ValueError("StackedFit: default loss switches, no producer thread (the stack draws for every image itself)")
            if f.net.out_act != 1:
                raise ValueError("StackedFit: sigmoid output (--normalize_type 1) only")
            # every loss switch / weight the stacked launches take from fits[0] must hold for all images
            same = ("pix_w", "use_comp", "cx_w", "lp_w", "lp_robust", "use_perceptual_loss", "style_w")
            for a_ in same:
                if hasattr(f0, a_) and getattr(f, a_, None) != getattr(f0, a_):
                    raise ValueError(f"StackedFit: the images of a stack share the loss switches and weights ({a_} differs)")
            if (f.net.n_knots, f.net.x_scale) != (net0.n_knots, net0.x_scale) or not torch.equal(f.net.spline, net0.spline):
                raise ValueError("StackedFit: the images of a stack share the robust loss's partition spline")
        M, dev = self.M, self.device
        self.n_p, self.P, self.kmax, self.n_pix = f0.patch_num, f0.patch_size, f0.topk, f0.N_rand
        self.n = self.n_pix + self.n_p * self.P * self.P
        self.Bp = ops.pad_rows(self.n)
        n_par = net0.n_params
        self.n_params = n_par
        stride = (n_par + 3) // 4 * 4
        # Split-K of the grouped weight gradient: one round of workgroups over the chip for the whole stack
        # (tiles x ksplit x M ~ the CU count; an image owns 8 / M of the XCDs)
        self.ksplit = int(ksplit) if ksplit else self._pick_ksplit()
        sizes = ops.train_workspace(self.K, self.Bp, self.ksplit, self.width)
        u8, f32 = torch.uint8, torch.float32
        self.params, self.m, self.v = (torch.zeros((M, stride), dtype=f32, device=dev) for _ in range(3))
        self.wf = torch.zeros((M, net0.wf.numel()), dtype=u8, device=dev)
        self.wb = torch.zeros((M, net0.wb.numel()), dtype=u8, device=dev)
        self.latents, self.lat_m, self.lat_v, self.dlatent = (torch.zeros((M, 8), dtype=f32, device=dev) for _ in range(4))
        self.loss_bufs = torch.zeros((M, 2), dtype=f32, device=dev)
        self.patch_loss = torch.zeros(M, dtype=f32, device=dev)
        self.pl_scratch = torch.zeros((M, ops.PIXEL_LOSS_SCRATCH), dtype=f32, device=dev)
        self.actF = torch.empty((M, sizes[1]), dtype=u8, device=dev)
        self.dzF = torch.empty((M, sizes[2]), dtype=u8, device=dev)
        self.gslabs = torch.empty((M, sizes[3] // 4), dtype=f32, device=dev)
        self.slab_stride = sizes[3] // 4 // self.ksplit
        self.pred = torch.empty((M, self.Bp, 3), dtype=f32, device=dev)
        self.dpred = torch.zeros((M, self.Bp, 3), dtype=f32, device=dev)        # rows >= n stay zero
        # the sampler's outputs, TWO sets: while the launches of iteration i read one, the sampler's host draws and device half
        # (one upload, one crop gather, one row assembly per image) of iteration i + 1 fill the other on a side stream
        nc = self.n_p * (1 + self.kmax)
        self._sets = [dict(coords=torch.zeros((M, self.Bp, 2), dtype=torch.int32, device=dev),
                           gt=torch.empty((M, self.n_pix, 3), dtype=f32, device=dev),
                           crops=torch.zeros((M, nc, 3, self.P, self.P), dtype=f32, device=dev),
                           cmasks=torch.zeros((M, nc, 1, self.P, self.P), dtype=f32, device=dev),
                           filled=None, free=None) for _ in range(2)]
        if f0.pixel_mask is not None:                     # remapping: per-pixel weights of the pixel rows (blurry pixels 0.3)
            for st in self._sets:
                st["pmask"] = torch.ones((M, self.n_pix), dtype=f32, device=dev)
        self._wset, self._ahead = 0, None
        self._s_smp = torch.cuda.Stream(dev)
        nxy = 2 * self.n_p * self.kmax
        self.xy = torch.zeros((M, nxy, 3, self.P, self.P), dtype=f32, device=dev)
        self.dxb = torch.zeros((M, nxy, 3, self.P, self.P), dtype=f32, device=dev)
        self.N_total = M * nxy
        self.edev = ops.embed_dev_blob([f.net.cfg for f in fits], dev)
        # re-home every fit's network state: its tensors become views of the stacked blobs
        for i, f in enumerate(fits):
            net = f.net
            for name, blob in (("params", self.params), ("m", self.m), ("v", self.v)):
                blob[i, :n_par].copy_(getattr(net, name))
                setattr(net, name, blob[i, :n_par])
            self.wf[i].copy_(net.wf); net.wf = self.wf[i]
            self.wb[i].copy_(net.wb); net.wb = self.wb[i]
            for name, blob in (("latents", self.latents), ("lat_m", self.lat_m), ("lat_v", self.lat_v), ("dlatent", self.dlatent)):
                blob[i, :6].copy_(getattr(net, name))
                setattr(net, name, blob[i, :6])
            self.loss_bufs[i].copy_(net._loss_bufs); net._loss_bufs = self.loss_bufs[i]
            net._ws = {}
        self.loss_idx = net0._loss_idx
        for f in fits:
            f.net._loss_idx = self.loss_idx
        # the contextual trunk is shared (frozen weights): the stack runs it once over all images' patches
        self.cx = f0.contextualLoss
        self.cx_w, self.lp_w, self.use_comp = f0.cx_w, f0.lp_w, f0.use_comp
        self._s_lp = torch.cuda.Stream(dev)
        self.style_side_stream = False    # the remapping task's style terms beside the contextual chain (see step_full)
        import os
        from concurrent.futures import ThreadPoolExecutor
        workers = min(M, max(1, (os.cpu_count() or 2) - 1))
        self._pool = ThreadPoolExecutor(workers, thread_name_prefix="npp-stack-draw") if (self.parallel_draws and workers > 1) else None
        self.batch_lpips, self._lp_in = True, None        # (False: every 'same' image's LPIPS branch on its own -- the tests' comparator)
        self.iteration = 0
        self.last_sources = None
        self._clean = False

    def _pick_ksplit(self):
        """Batch splits per image of the grouped weight-gradient launch: the count that minimises rounds x rows per workgroup
        (tiles x splits x M workgroups of one CU each)."""
        from ._lib import lib
        tiles = lib(self.width).npp_mlp_wgrad_tiles(self.K)
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        n_wg = self.Bp // 64
        best = min(range(1, 25), key=lambda ks: (-(-tiles * ks * self.M // cus) * -(-n_wg // ks), ks))
        return best

    # ---- host half: one draw per image (each fit's own sampler and random stream), device half into the stacked buffers ----
    def sample(self):
        """-> the next list of M batches in the serial order of every image's random stream: the look-ahead draw step_full() left
        pending if there is one (consumed), else a fresh draw.  At most TWO draws can be outstanding (two device buffer sets):
        step_from() refuses a list whose set has been refilled since."""
        if self._ahead is not None:
            b, self._ahead = self._ahead, None
            return b
        return self._draw()

    def _host_draws(self):
        """The M host draws of one iteration, started now: -> futures (thread pool) or the finished draws."""
        if self._pool is not None:
            return [self._pool.submit(f.draw_batch) for f in self.fits]
        return [f.draw_batch() for f in self.fits]

    def _draw(self, draws=None):
        """-> list of M batches (None for an image whose sampler found no valid real patch this iteration), materialised on the
        sampler stream into the buffer set that is not in use; the list carries that set (step_from waits for it).  draws: the
        host halves if they were started earlier (_host_draws)."""
        w = self._wset
        self._wset ^= 1
        st = self._sets[w]
        out = _Batches()
        out.set = w
        main = torch.cuda.current_stream(self.device)
        if st["free"] is not None:
            self._s_smp.wait_event(st["free"])            # the iteration that read this set last has been enqueued AND must finish first
        else:
            self._s_smp.wait_stream(main)                 # first use: behind the constructor's copies
        # the host halves first, one thread per image: the images' random streams are independent and the native stream (the
        # reference's MT19937 sequence, csrc/npp_host_rng.hip) draws outside the GIL -- one after the other, 8 reference-stream draws
        # (0.4 ms each) took longer than the stacked iteration they feed (4.7 ms per iteration against 3.5 ms of device time)
        if draws is None:
            draws = self._host_draws()
        draws = [d.result() if hasattr(d, "result") else d for d in draws]
        with torch.cuda.stream(self._s_smp):
            for i, (f, d) in enumerate(zip(self.fits, draws)):
                f.last_draw = d
                f.iteration += 1
                o = dict(coords=st["coords"][i], gt=st["gt"][i], crops=st["crops"][i], cmasks=st["cmasks"][i])
                if "pmask" in st:
                    o["pmask"] = st["pmask"][i]
                b = f.materialise_batch(d, out=o)
                if b is None:
                    f.skipped += 1
                out.append(b)
            ev = torch.cuda.Event()
            ev.record(self._s_smp)
        st["filled"] = ev
        st["gen"] = out.gen = st.get("gen", 0) + 1
        return out

    def shape_change_due(self):
        """Whether the next draw changes the batch shape (an image's patch-size decay, train.py:137-141): the stack must then be
        re-formed (restacked()) before it draws."""
        return self._ahead is None and any(f.decay_due() for f in self.fits)

    def restacked(self):
        """-> a new StackedFit over the same fits, after the patch-size decay that is due (the fits' tensors are views of this stack's
        blobs; the new stack copies them into blobs of the new batch shape).  This object must not be stepped again."""
        if self._ahead is not None:
            raise RuntimeError("StackedFit.restacked: a look-ahead draw is pending")
        torch.cuda.synchronize(self.device)
        for f in self.fits:
            if f.decay_due():
                f.apply_decay()
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        return StackedFit(self.fits)

    def step_full(self):
        """One iteration of the loop body for every image of the stack.  -> number of images that took a step.  The NEXT
        iteration's sampling (host draws + device half) is issued right behind this one's launches: it never reads network state, so
        the random streams and the results are those of the serial order; it runs on a side stream under this iteration's kernels."""
        b = self.sample()
        # the NEXT iteration's host draws run on the pool's threads while this thread enqueues the launches below (no look-ahead
        # across a patch-size decay: the next draw belongs to another batch shape)
        nxt = None if any(f.decay_due() for f in self.fits) else self._host_draws()
        try:
            n = self.step_from(b)
        except BaseException:
            if nxt is not None:                           # the streams have advanced: keep the draws (the serial order holds)
                self._ahead = self._draw(nxt)
            raise
        self._ahead = None if nxt is None else self._draw(nxt)
        return n

    # ---- device half ---------------------------------------------------------------------------------------------------------
    def step_from(self, batches):
        ops.check_current(self.device)
        M, fits = self.M, self.fits
        st = self._sets[getattr(batches, "set", 0)]
        coords, gt, crops, cmasks = st["coords"], st["gt"], st["crops"], st["cmasks"]
        if getattr(batches, "gen", None) is not None and batches.gen != st.get("gen"):
            raise RuntimeError("StackedFit.step_from: this draw's device buffers have been refilled by a later sample() (two draws can be "
                               "outstanding at most)")
        # the whole stack's shape is checked BEFORE any state moves (an image whose patch size decayed would otherwise leave the
        # earlier images' step counters advanced)
        for b in batches:
            if b is not None and (b["P"], b["n_p"], b["n_pix"], b["bp"]) != (self.P, self.n_p, self.n_pix, self.Bp):
                raise RuntimeError("StackedFit: an image's batch left the stack's shape (patch-size decay is not stacked: rebuild the stack)")
        if st["filled"] is not None:
            torch.cuda.current_stream(self.device).wait_event(st["filled"])
        it = (StackIter * M)()
        x0 = 0
        lr_used = [0.0] * M
        for i, (f, b) in enumerate(zip(fits, batches)):
            e = it[i]
            if b is None:
                continue
            net = f.net
            src, k = b["source"], b["k"]
            e.active, e.k, e.nk, e.x0 = 1, k, self.n_p * k, x0
            e.comp = int(self.use_comp and src == "val")
            e.same = int(src == "same")
            # with_lp: the image has a second patch-gradient term (LPIPS on 'same' iterations, the style term on every one) and
            # its fp32 batch is written for that branch
            e.with_lp = int((src == "same" and f.use_perceptual_loss) or f.style is not None)
            net.opt_step += 1
            step = net.opt_step
            # npp_adam_step_net_pack's host arithmetic: float arguments widened to double, pow / sqrt in double, result to float
            b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
            bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
            e.step_size = float(np.float32(net.lr)) / bc1
            e.inv_sqrt_bc2 = 1.0 / math.sqrt(bc2)
            lr_used[i] = net.lr
            x0 += e.nk
        X = x0
        n_active = sum(1 for b in batches if b is not None)
        self.last_sources = [None if b is None else b["source"] for b in batches]
        if n_active == 0:
            self.iteration += 1
            st["free"] = st["filled"]
            return 0
        it_dev = ops.h2d(np.frombuffer(bytes(it), np.uint8).copy(), self.device)
        # zero_grad(): the fused Adam launch of the previous iteration cleared each active image's latent gradient and idle loss
        # accumulator; switch to it (NPPNet.zero_grad).  Images that sat the previous iteration out are cleared here.
        if self._clean:
            self.loss_idx ^= 1
            for i in self._stale:
                self.loss_bufs[i, self.loss_idx].zero_()
                self.dlatent[i].zero_()
        else:
            self.loss_bufs[:, self.loss_idx].zero_()
            self.dlatent.zero_()
        for f in fits:
            f.net._loss_idx = self.loss_idx
            if f.percepLoss.touched:
                f.percepLoss.zero_latent_grads()
        K, W = self.K, self.width
        ops.mlp_fwd_stack(coords, self.edev, M, K, self.wf, self.params, self.pred, self.actF, it_dev, W)
        cx = self.cx
        sc, sh = cx.input_norm()
        net0 = fits[0].net
        loss = (self.pred, gt, st.get("pmask"), self.latents, net0.spline, net0.n_knots, net0.x_scale, fits[0].pix_w,
                self.loss_bufs[:, self.loss_idx:], self.dpred, self.dlatent, self.n_pix, self.pl_scratch, net0.quad)
        t = cx.hip_trunk
        with_lp = [i for i, b in enumerate(batches) if b is not None and it[i].with_lp]
        ops.trunk_patch_in_loss_stack(self.pred, self.n_pix, crops, cmasks, M, self.n_p, self.P, X, self.N_total, sc, sh,
                                      t.input_buffer(self.N_total, self.P, self.P), self.xy if with_lp else None, self.patch_loss,
                                      it_dev, loss, gt.stride(0), self.latents.stride(0), self.loss_bufs.stride(0))
        main = torch.cuda.current_stream(self.device)
        if with_lp:                                       # the LPIPS / style branches of the images beside the contextual chain
            self._s_lp.wait_stream(main)
            with torch.cuda.stream(self._s_lp):
                # 'same' images without a style term: ONE pass of the (shared, frozen) VGG16 trunk for all of them when there are
                # several (batch_lpips; the heads run per image on its own latents) -- a lone one replays its captured graph
                together = [i for i in with_lp if it[i].same and fits[i].use_perceptual_loss and fits[i].style is None]
                if self.batch_lpips and len(together) >= 2:
                    self._lpips_together(together)
                else:
                    together = []
                for i in with_lp:
                    if i in together:
                        continue
                    nk = it[i].nk
                    f = fits[i]
                    dxb = None
                    if it[i].same and f.use_perceptual_loss:
                        dxb = f.lpips_branch(self.xy[i, :2 * nk], nk, self.lp_w, self.patch_loss[i:i + 1])[:nk]   # (a captured graph from its third use on)
                    if f.style is not None and self.style_side_stream:   # NPP_remapping/train.py:253-261: own latents per image
                        f.style.zero_latent_grads()
                        dxs = f.style.fused(self.xy[i, :2 * nk], nk, f.style_w, self.patch_loss[i:i + 1])[:nk]
                        dxb = dxs if dxb is None else dxb + dxs
                    if dxb is not None:
                        self.dxb[i, :nk].copy_(dxb)
        style_main = []
        if with_lp and not self.style_side_stream:
            # the style terms on the MAIN stream (default): on the side stream, beside the contextual chain, the style latents' gradients
            # moved in their last bits from run to run (16-entry groups of the alpha half; needs the side stream's temporaries to be freed
            # and re-used each iteration -- kept alive, every run agrees).  Not resolved (tools/r6_side_stream_repro.py);
            # the main-stream order removes it (tests/test_gpu_poison.py); style_side_stream = True restores the overlap
            for i in with_lp:
                f = fits[i]
                if f.style is not None:
                    nk = it[i].nk
                    f.style.zero_latent_grads()
                    style_main.append((i, nk, f.style.fused(self.xy[i, :2 * nk], nk, f.style_w, self.patch_loss[i:i + 1])[:nk],
                                       bool(it[i].same and f.use_perceptual_loss)))
        shape = (self.N_total, 3, self.P, self.P)
        feats = t._forward(shape, sc, sh, True, n_run=2 * X, n_keep=X)[0]

        def top(y, N, nn, c, H, W, dz):                   # the core's last launch writes the trunk's flat gradient tensor itself
            ops.cx_fwd_bwd_flat(feats[:X], feats[X:2 * X], y, dz, N, cx.band_width, self.cx_w, self.patch_loss, 1, it_dev, M)
        dx_a = t._backward([True], X, sc, shape, zero_rest=False, top_writer=top)
        if with_lp:
            main.wait_stream(self._s_lp)
            for i, nk, dxs, has_lp in style_main:           # (joined: the LPIPS branch of a 'same' image wrote its share already)
                if has_lp:
                    self.dxb[i, :nk].add_(dxs)
                else:
                    self.dxb[i, :nk].copy_(dxs)
        ops.mlp_bwd_patch_stack(self.dpred, self.pred, M, K, self.wb, self.params, self.actF, self.dzF, dx_a,
                                self.dxb if with_lp else None, cmasks, self.n_pix, self.n_p, self.P, it_dev, W)
        ops.mlp_wgrad_stack(self.dzF, self.actF, self.Bp, M, K, self.ksplit, self.gslabs, it_dev, W)
        idle = self.loss_bufs[:, 1 - self.loss_idx:2 - self.loss_idx]
        ops.adam_step_net_pack_stack(self.params, self.m, self.v, self.n_params, self.gslabs, self.ksplit, self.slab_stride,
                                     self.latents, self.lat_m, self.lat_v, self.dlatent, 6, idle, M, K, self.wf, self.wb, it_dev,
                                     self.pl_scratch, self.loss_bufs[:, self.loss_idx:], W)
        self._clean = True
        free = torch.cuda.Event()                          # the sampler may refill this set once everything above has run
        free.record(torch.cuda.current_stream(self.device))
        st["free"] = free
        self._stale = [i for i, b in enumerate(batches) if b is None]
        for i, (f, b) in enumerate(zip(fits, batches)):
            if b is None:
                continue
            net = f.net
            net.lr = net.lrate * (0.1 ** (net.global_step / (net.lrate_decay * 100)))      # train.py:256-262, NPPNet.optimizer_step
            if net.lr_clock:
                net.global_step += 1
            net._clean = True
            f.last_source = b["source"]
            f.last_patch_loss = self.patch_loss[i:i + 1]
            if f.percepLoss.touched:                      # only 'same' iterations give the LPIPS latents a gradient
                f.percepLoss.adam_step(lr_used[i])
            if f.style is not None:                       # the style latents are in the same optimiser (helpers.py:153-159)
                f.style.adam_step(lr_used[i])
        self.iteration += 1
        return n_active

    def _lpips_together(self, idx):
        """The LPIPS branch (train.py:241-250) of the 'same' images idx in one trunk pass: their fp32 batches [x | y] (n_p samples each
        on 'same' draws: k = 1, sampler.py:338) are gathered into [x of all | y of all], LPIPS.fused_groups runs VGG16 forward, one
        heads launch per image (its latents, its loss accumulator), one data-gradient pass; the gradients go back to the images' rows
        of dxb.  Per image the arithmetic is lpips_branch()'s; the trunk launches see 2 n_p len(idx) patches instead of 2 n_p."""
        n_p, M, P = self.n_p, self.M, self.P
        nxy = self.xy.shape[1]
        XL = n_p * len(idx)
        rows = np.concatenate([np.concatenate([i * nxy + h * n_p + np.arange(n_p) for i in idx]) for h in (0, 1)]).astype(np.int64)
        rows_dev = ops.h2d(rows, self.device)
        if self._lp_in is None:                               # fixed geometry (every image a 'same' one): one set of trunk buffers
            self._lp_in = torch.zeros((2 * M * n_p, 3, P, P), dtype=torch.float32, device=self.device)
        torch.index_select(self.xy.view(M * nxy, 3, P, P), 0, rows_dev, out=self._lp_in[:2 * XL])
        groups = [(j * n_p, n_p, self.fits[i].percepLoss, self.patch_loss[i:i + 1]) for j, i in enumerate(idx)]
        f0 = self.fits[0]
        dx = f0.percepLoss.fused_groups(self._lp_in, XL, groups, self.lp_w, normalize=True, use_robust=f0.lp_robust)
        self.dxb.view(M * nxy, 3, P, P).index_copy_(0, rows_dev[:XL], dx[:XL])

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for f in self.fits:
            f.close()

    def psnr(self, region="known"):
        return [f.psnr(region) for f in self.fits]
