"""Host-side mirror of the reference's model plumbing for the hot path:
create_npp_net (models/helpers.py:75-175), render (:41-62), the Adam + LR rule of
NPP_completion/train.py:253-263, with the state kept in device buffers that the HIP
kernels read directly.  Nothing here computes; it owns memory and calls the C ABI."""
import os

import numpy as np
import torch

from . import ops
from ._lib import EmbedCfg, param_layout, NPP_E, NPP_WIDTH

LATENT_ALPHA_INIT = 2.3841858e-07   # logit(0.5) in fp32 (robust_loss_pytorch/util.py:75-83)


class NPPNet:
    """NPP_Net (K>1) / NPP_Net_top1 (K==1) with its embedders, optimiser state and the
    adaptive pixel-loss latents (the reference's module-level `adaptive_pix`,
    models/helpers.py:8-9), all resident on one GPU.

    state_dict()/load_state_dict() use the reference's tensor names and layouts
    (models/networks.py:40-49): the blob is the reference's tensors back to back.
    """

    def __init__(self, angles_deg, periods, freqs, res, params=None, device="cuda", ksplit=None,
                 lrate=5e-4, lrate_decay=500, offsets=(0.0, -1.0, 1.0, 0.5, -0.5), width=NPP_WIDTH, loss_type="robust_loss_adaptive", out_act=1):
        """loss_type: --loss_type of options/arg_config.py:34 (models/mse_calculator.py:19-23): 'robust_loss_adaptive' (default), 'l2',
        'robust_loss' (the two non-adaptive forms leave the adaptive latents untouched: no gradient reaches them)."""
        self.loss_type, self.quad = loss_type, ops.quad_coef(loss_type)
        if out_act not in (1, 2):
            raise ValueError("out_act: 1 (sigmoid, --normalize_type 1) or 2 (tanh, --normalize_type 2: images in [-1, 1]; helpers.py:55-58)")
        self.out_act = int(out_act)
        self.cfg = EmbedCfg.make(angles_deg, periods, freqs, res, offsets)
        self.K = int(self.cfg.K)
        self.width = int(width)      # 256 (BASELINE configs) or 512 (the reference's default --netwidth): one fused library each
        self.device = ops.select_device(device)
        self.layout, self.n_params = param_layout(self.K, self.width)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        # adaptive_pix latents: [latent_alpha(3) | latent_scale(3)]  (adaptive.py:146-181)
        self.latents = torch.tensor([LATENT_ALPHA_INIT] * 3 + [0.0] * 3, dtype=torch.float32, device=self.device)
        self.lat_m = torch.zeros_like(self.latents)
        self.lat_v = torch.zeros_like(self.latents)
        self.dlatent = torch.zeros(6, dtype=torch.float32, device=self.device)
        # two loss accumulators used alternately: the fused Adam launch clears the idle one, so the
        # value of the finished iteration stays readable and zero_grad() needs no fill kernels
        self._loss_bufs = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._loss_idx = 0
        self._clean = False
        self.spline, self.n_knots, self.x_scale = ops.load_spline(self.device)
        self.ksplit = int(ksplit) if ksplit else ops.auto_ksplit(self.K, self.device, self.width)   # None / 0: one round of workgroups
        self.lrate, self.lrate_decay = float(lrate), int(lrate_decay)
        self.lr = float(lrate)
        self.global_step = 0        # train.py:337
        self.lr_clock = True        # False: the LR clock never advances (NPP_segmentation/train.py:408: `global_step += 1` sits
                                    # outside the loop there, so that task trains at a constant lrate) -- reproduced, not fixed
        self.opt_step = 0           # Adam's per-parameter step count
        self.wf = torch.zeros(ops.pack_bytes(self.K, 0, self.width), dtype=torch.uint8, device=self.device)
        self.wb = torch.zeros(ops.pack_bytes(self.K, 1, self.width), dtype=torch.uint8, device=self.device)
        self._ws = {}
        self.fused_repack = True    # False: Adam and the weight re-pack as two launches (comparator of the fused adam_pack launch)
        if params is not None:
            self.load_state_dict(params)

    # ---- parameters -----------------------------------------------------------------
    def load_state_dict(self, sd):
        flat = np.zeros(self.n_params, np.float32)
        for name, off, rows, cols in self.layout:
            a = sd[name]
            a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            if a.size != rows * cols:
                raise ValueError(f"{name}: expected {rows}x{cols}, got {a.shape}")
            flat[off:off + rows * cols] = a.astype(np.float32).reshape(-1)
        self.params.copy_(torch.from_numpy(flat))
        self.repack()

    def state_dict(self):
        flat = self.params.detach().cpu().numpy()
        out = {}
        for name, off, rows, cols in self.layout:
            a = flat[off:off + rows * cols]
            out[name] = a.reshape(rows, cols).copy() if name.endswith("weight") else a.copy()
        return out

    def grads(self):
        """Sum of the split-K slabs, as a reference-named dict (for tests)."""
        ws = self._ws_last
        g = ws["gslabs"].view(self.ksplit, -1)[:, :self.n_params].sum(0).cpu().numpy()     # slab stride = n_params rounded up to 4
        out = {}
        for name, off, rows, cols in self.layout:
            a = g[off:off + rows * cols]
            out[name] = a.reshape(rows, cols).copy() if name.endswith("weight") else a.copy()
        return out

    def repack(self):
        ops.pack_weights(self.params, self.K, self.wf, self.wb, self.width)

    # ---- workspaces -----------------------------------------------------------------
    def workspace(self, Bp):
        ws = self._ws.get(Bp)
        if ws is None:
            s = ops.train_workspace(self.K, Bp, self.ksplit, self.width)
            dev = self.device
            ws = {
                "actT": torch.empty(s[1], dtype=torch.uint8, device=dev),
                "dzT": torch.empty(s[2], dtype=torch.uint8, device=dev),
                "gslabs": torch.empty(s[3] // 4, dtype=torch.float32, device=dev),
                "pred": torch.empty((Bp, 3), dtype=torch.float32, device=dev),
                "dpred": torch.zeros((Bp, 3), dtype=torch.float32, device=dev),
            }
            self._ws[Bp] = ws
        self._ws_last = ws
        return ws

    # ---- the path ---------------------------------------------------------------------
    def render(self, coords):
        """render(None, emb[coords], args, **render_kwargs) of the reference (helpers.py:41-62)
        for arbitrary pixel coordinates; no gradient state is kept (train.py:277-309)."""
        n = coords.shape[0]
        bp = ops.pad_rows(n)
        if bp != n:
            pad = torch.zeros((bp - n, 2), dtype=torch.int32, device=coords.device)
            coords = torch.cat([coords, pad], 0)
        pred = ops.mlp_fwd(coords.contiguous(), self.cfg, self.wf, self.params, width=self.width, out_act=self.out_act)
        return pred[:n]

    def render_fp32(self, coords):
        """The same render in EXACT fp32 (BASELINE config c4): fused chain on v_mfma_f32_32x32x2_f32 (npp_mlp_fwd32), the
        reference's own arithmetic type -- no bf16 operand rounding.  The fp32 weight pack is rebuilt when the parameters have
        changed since the last call."""
        n = coords.shape[0]
        bp = ops.pad_rows(n)
        if bp != n:
            coords = torch.cat([coords, torch.zeros((bp - n, 2), dtype=torch.int32, device=coords.device)], 0)
        stamp = (self.opt_step, self.params._version)
        if getattr(self, "_w32_stamp", None) != stamp:
            self._w32 = ops.pack_weights32(self.params, self.K, getattr(self, "_w32", None), self.width)
            self._w32_stamp = stamp
        return ops.mlp_fwd32(coords.contiguous(), self.cfg, self._w32, self.params, out_act=self.out_act, width=self.width)[:n]

    def forward_train(self, coords_padded):
        """Forward with stashes; coords must already be padded to a multiple of 64 rows."""
        ws = self.workspace(coords_padded.shape[0])
        ops.mlp_fwd(coords_padded, self.cfg, self.wf, self.params, ws["pred"], ws["actT"], self.width, out_act=self.out_act)
        return ws["pred"]

    def backward(self, Bp, patch=None):
        """loss.backward() through the MLP: consumes ws['dpred'] (rows beyond the batch 0).  patch = (dx_a, dx_b, fmask, rmask,
        row0, n_p, k, P, comp): the patch rows' dL/dpred is formed inside the launch from the patch losses' image gradients."""
        ws = self._ws[Bp]
        if patch is not None:
            ops.mlp_bwd_patch(ws["dpred"], ws["pred"], self.K, self.wb, self.params, ws["actT"], ws["dzT"], *patch, width=self.width, out_act=self.out_act)
        else:
            ops.mlp_bwd(ws["dpred"], ws["pred"], self.K, self.wb, self.params, ws["actT"], ws["dzT"], self.width, out_act=self.out_act)
        ops.mlp_wgrad(ws["dzT"], ws["actT"], Bp, self.K, self.ksplit, ws["gslabs"], self.width)

    def pixel_loss(self, Bp, n_rows, gt, mask=None, weight=1.0):
        """img2mse on the first n_rows rows of the current prediction (train.py:195);
        writes dL/dpred for those rows and accumulates the latent gradients."""
        ws = self._ws[Bp]
        ops.pixel_loss(ws["pred"][:n_rows], gt, mask, self.latents, self.spline, self.n_knots, self.x_scale,
                       weight, self.loss_buf, ws["dpred"][:n_rows], self.dlatent, quad=self.quad)

    def pixel_loss_args(self, Bp, n_rows, gt, mask=None, weight=1.0):
        """The argument tuple of pixel_loss() for a launch that carries the loss along (ops.trunk_patch_in(loss=...))."""
        ws = self._ws[Bp]
        if getattr(self, "_pl_scratch", None) is None and ops.DETERMINISTIC and self.fused_repack:
            # the launch leaves its per-block partial sums here; the fused Adam launch of the iteration (_adam) adds them in block
            # order: bit-reproducible sums at no cost (include/npp_hip.h npp_pixel_loss_args.scratch)
            self._pl_scratch = torch.zeros(ops.PIXEL_LOSS_SCRATCH, dtype=torch.float32, device=self.device)
        return (ws["pred"][:n_rows], gt, mask, self.latents, self.spline, self.n_knots, self.x_scale, weight, self.loss_buf,
                ws["dpred"][:n_rows], self.dlatent, getattr(self, "_pl_scratch", None), self.quad)

    def optimizer_step(self, Bp):
        """optimizer.step() + the LR rule of train.py:253-263 + global_step += 1 (:337)."""
        ws = self._ws[Bp]
        self.opt_step += 1
        idle = self._loss_bufs[1 - self._loss_idx:2 - self._loss_idx]
        self._adam(ws["gslabs"], self.ksplit, ws["gslabs"].numel() // self.ksplit, idle)
        self.lr = self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 100)))
        if self.lr_clock:
            self.global_step += 1

    def _adam(self, gslabs, n_slabs, stride, idle):
        """optimizer.step() over the blob + latents and the re-pack of the bf16 MFMA packs: one launch (fused_repack, default)
        or two (npp_adam_step_net, then npp_pack_weights: the comparator)."""
        if self.fused_repack:
            # (the pixel-loss launch of a folded iteration left its block partials in _pl_scratch: summed here in block order;
            #  the buffer's count word is zero whenever no such launch ran since the last step)
            ops.adam_step_net_pack(self.params, self.m, self.v, gslabs, n_slabs, stride, self.latents, self.lat_m, self.lat_v,
                                   self.dlatent, idle, self.lr, self.opt_step, self.K, self.wf, self.wb, self.width,
                                   pl_partials=getattr(self, "_pl_scratch", None), loss_cur=self.loss_buf)
        else:
            ops.adam_step_net(self.params, self.m, self.v, gslabs, n_slabs, stride, self.latents, self.lat_m, self.lat_v,
                              self.dlatent, idle, self.lr, self.opt_step)
            self.repack()
        self._clean = True

    @property
    def loss_buf(self):
        """Accumulator of the current iteration's pixel loss (1 float on the device)."""
        return self._loss_bufs[self._loss_idx:self._loss_idx + 1]

    def zero_grad(self, force=False):
        """optimizer.zero_grad() (train.py:192).  After optimizer_step() the latent gradient and the
        idle loss accumulator are already zero: switch to it instead of launching fill kernels."""
        if self._clean and not force:
            self._loss_idx ^= 1
        else:
            self.dlatent.zero_()
            self.loss_buf.zero_()
        self._clean = False
