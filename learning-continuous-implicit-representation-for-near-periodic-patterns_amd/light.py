"""Proposal-ranking fits: NPP_Net_light + the candidate loop and score of NPP_proposal/search.py:85-215,
host side.  Every operator runs in libnpp_hip.so: the is_search embedders through npp_warp_fwd / npp_fourier_fwd, the MLP
through the generic dense-layer kernels (npp_linear_*: exact fp32 MFMA), the adaptive robust pixel loss and Adam through
the same kernels as the main loop, the score through the trunk / LPIPS / contextual-loss kernels.

The problem is tiny (2048 rows x 0.3 M parameters x 300 iterations per candidate) and launch-bound; candidates are
independent, so they shard one per GPU exactly like images do (parallel.py).
"""
import os

import numpy as np
import torch

from . import ops
from ._lib import EmbedCfg
from .model import LATENT_ALPHA_INIT

_SNAKE, _SIGMOID = 1, 2


def light_layout(W=256, D=4, in_pos=42, in_per=20):
    """[(name, rows, cols)] of the tensors NPP_Net_light's forward uses (models/networks.py:199-214,216-262 with
    len(freq_scales) == 1), in the reference's state_dict order."""
    out = [(f"periodic_linears.{i}", W, in_per if i == 0 else W) for i in range(D)]
    out += [("pos_linears.0", W // 2, W + in_pos), ("feature_linear1", W, W), ("rgb_linear", 3, W // 2)]
    return out


def _stored_cols(c):
    """Columns a weight matrix is STORED with: rounded up to 4 (zero columns; pos_linears.0 is 298 -> 300 wide).  The dense-layer
    kernels fetch 16 bytes per lane along the contraction only when every row starts 16-byte aligned; an odd leading dimension
    drops them to 4-byte loads (measured: the 298-wide layer took 44 us against 18 for its share of the arithmetic).  The pad
    columns meet zero inputs, get zero gradients and stay zero under Adam; state_dict() / grads() / load_state_dict() speak the
    reference's shapes."""
    return (c + 3) // 4 * 4


class NPPNetLight:
    """NPP_Net_light(D, W, activation='snake') with its two is_search embedders, Adam state and the adaptive pixel-loss
    latents.  state_dict names / layouts are the reference's; scale_linears / feature_linear2 / alpha_linear (constructed by
    the reference, never used when len(freq_scales) == 1) are not kept."""

    def __init__(self, angles_deg, periods, freqs, res, params, W=256, D=4, device="cuda", lrate=5e-4, lrate_decay=500, storage=None,
                 loss_type="robust_loss_adaptive"):
        """storage: optional dict of preallocated float32 device vectors (params, grad, m, v: n_params each; latents, lat_m, lat_v,
        dlatent: 6; loss_buf: 1) -- rows of the stacked blobs of an NPPNetLightBatch."""
        self.device = ops.select_device(device)
        self.quad = ops.quad_coef(loss_type)              # --loss_type (models/mse_calculator.py:19-23): 0 = the adaptive robust loss
        self.res = (int(res[0]), int(res[1]))
        self.W, self.D = int(W), int(D)
        self.freqs = [float(f) for f in np.asarray(freqs).reshape(-1)]
        self.in_pos, self.in_per = 2 * (1 + 2 * len(self.freqs)), 20
        self.cfg = EmbedCfg.make(np.asarray(angles_deg, np.float32).reshape(1, 2), np.asarray(periods, np.float32).reshape(1, 2),
                                 np.zeros(10, np.float32), self.res)
        self.layout = light_layout(self.W, self.D, self.in_pos, self.in_per)
        self.kpos = _stored_cols(self.W + self.in_pos)             # width of the [feature1 | input_pos | 0-pad] buffer
        n = sum(r * _stored_cols(c) + r for _, r, c in self.layout)
        self.n_params = n
        if storage is None:
            self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.grad = torch.zeros_like(self.params)
            self.m, self.v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        else:
            self.params, self.grad, self.m, self.v = (storage[k] for k in ("params", "grad", "m", "v"))
            assert all(t.shape == (n,) and t.is_contiguous() for t in (self.params, self.grad, self.m, self.v))
        self.w, self.b, self.dw, self.db = {}, {}, {}, {}
        off = 0
        for name, r, c in self.layout:
            cs = _stored_cols(c)
            self.w[name], self.dw[name] = self.params[off:off + r * cs].view(r, cs), self.grad[off:off + r * cs].view(r, cs)
            off += r * cs
            self.b[name], self.db[name] = self.params[off:off + r], self.grad[off:off + r]
            off += r
        if params is not None:
            self.load_state_dict(params)
        if storage is None:
            self.latents = torch.tensor([LATENT_ALPHA_INIT] * 3 + [0.0] * 3, dtype=torch.float32, device=self.device)
            self.lat_m, self.lat_v = torch.zeros_like(self.latents), torch.zeros_like(self.latents)
            self.dlatent = torch.zeros(6, dtype=torch.float32, device=self.device)
            self.loss_buf = torch.zeros(1, dtype=torch.float32, device=self.device)
        else:
            self.latents, self.lat_m, self.lat_v, self.dlatent, self.loss_buf = (storage[k] for k in ("latents", "lat_m", "lat_v", "dlatent", "loss_buf"))
            self.latents.copy_(torch.tensor([LATENT_ALPHA_INIT] * 3 + [0.0] * 3, dtype=torch.float32))
        self.spline, self.n_knots, self.x_scale = ops.load_spline(self.device)
        self.lrate, self.lrate_decay, self.lr = float(lrate), int(lrate_decay), float(lrate)
        self.global_step, self.opt_step = 0, 0
        self._ws = {}

    def load_state_dict(self, sd):
        for name, r, c in self.layout:
            for part, dst in (("weight", self.w[name][:, :c]), ("bias", self.b[name])):
                a = sd[f"{name}.{part}"]
                a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                dst.copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)).reshape(dst.shape))
            self.w[name][:, c:].zero_()

    def state_dict(self):
        out = {}
        for name, _, c in self.layout:
            out[f"{name}.weight"], out[f"{name}.bias"] = self.w[name][:, :c].cpu().numpy().copy(), self.b[name].cpu().numpy().copy()
        return out

    def grads(self):
        out = {}
        for name, _, c in self.layout:
            out[f"{name}.weight"], out[f"{name}.bias"] = self.dw[name][:, :c].cpu().numpy().copy(), self.db[name].cpu().numpy().copy()
        return out

    # ---- embedders (get_embedder(is_search=True), models/embedder.py:52-54,76-88) ----------------------
    def embed(self, coords_yx):
        """coords (N,2) int32 (row, col) -> (x_pos (N,42), x_per (N,20))."""
        c = coords_yx.to(torch.float32)
        H, W = self.res
        norm = torch.stack([(c[:, 0] / H - 0.5) * 2, (c[:, 1] / W - 0.5) * 2], 1).contiguous()       # embedder.py:52-54
        x_pos = ops.fourier_fwd(norm, self.freqs, include_input=True)
        v = ops.warp_fwd(coords_yx.contiguous(), self.cfg)                                            # (N, 22)
        x_per = torch.cat([v[:, 1:11], v[:, 12:22]], 1).contiguous()                                  # include_input = False
        return x_pos, x_per

    def _work(self, B):
        ws = self._ws.get(B)
        if ws is None:
            f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)      # noqa: E731
            W = self.W
            ws = dict(z=[f(B, W) for _ in range(self.D)], h=[f(B, W) for _ in range(self.D)],
                      hp=torch.zeros(B, self.kpos, dtype=torch.float32, device=self.device),          # pad columns stay zero
                      zp=f(B, W // 2), ap=f(B, W // 2), raw=f(B, 3), pred=f(B, 3), dpred=f(B, 3), draw=f(B, 3), dap=f(B, W // 2),
                      dzp=f(B, W // 2), df1=f(B, W), dh=f(B, W), dz=f(B, W))
            self._ws[B] = ws
        return ws

    # ---- NPP_Net_light.forward + render's sigmoid (networks.py:216-262, helpers.py:55-56) ----------------
    def forward(self, x_pos, x_per):
        B = x_per.shape[0]
        ws = self._work(B)
        ws["x_per"] = x_per
        h = x_per
        for i in range(self.D):
            name = f"periodic_linears.{i}"
            ops.linear_fwd(h, self.w[name], self.b[name], _SNAKE, ws["h"][i], ws["z"][i])
            h = ws["h"][i]
        W = self.W
        ops.linear_fwd(h, self.w["feature_linear1"], self.b["feature_linear1"], 0, ws["hp"][:, :W])      # cat[feature1, input_pos] :247
        ws["hp"][:, W:W + self.in_pos].copy_(x_pos)
        ops.linear_fwd(ws["hp"], self.w["pos_linears.0"], self.b["pos_linears.0"], _SNAKE, ws["ap"], ws["zp"])
        ops.linear_fwd(ws["ap"], self.w["rgb_linear"], self.b["rgb_linear"], 0, ws["raw"])
        ops.act_fwd(ws["raw"], _SIGMOID, ws["pred"])
        return ws["pred"]

    def backward(self, B):
        """loss.backward(): consumes ws['dpred'] = dL/dpred, fills self.grad."""
        ws, W = self._ws[B], self.W
        # the seven weight-gradient launches split their contraction and meet by atomicAdd in a zeroed output: ONE clear of the
        # whole gradient blob + accumulate, instead of a memset per dW and per db (14 fills of ~4.4 us = 15 % of the iteration)
        self.grad.zero_()
        ops.act_bwd(ws["dpred"], ws["pred"], _SIGMOID, ws["draw"])
        ops.linear_bwd_weight(ws["draw"], ws["ap"], self.dw["rgb_linear"], self.db["rgb_linear"], accumulate=True)
        ops.linear_bwd_data(ws["draw"], self.w["rgb_linear"], ws["dap"])
        ops.act_bwd(ws["dap"], ws["zp"], _SNAKE, ws["dzp"])
        ops.linear_bwd_weight(ws["dzp"], ws["hp"], self.dw["pos_linears.0"], self.db["pos_linears.0"], accumulate=True)
        ops.linear_bwd_data(ws["dzp"], self.w["pos_linears.0"], ws["df1"], in_used=W)                    # no gradient to input_pos
        ops.linear_bwd_weight(ws["df1"], ws["h"][self.D - 1], self.dw["feature_linear1"], self.db["feature_linear1"], accumulate=True)
        ops.linear_bwd_data(ws["df1"], self.w["feature_linear1"], ws["dh"])
        for i in range(self.D - 1, -1, -1):
            name = f"periodic_linears.{i}"
            ops.act_bwd(ws["dh"], ws["z"][i], _SNAKE, ws["dz"])
            x_in = ws["h"][i - 1] if i > 0 else ws["x_per"]
            ops.linear_bwd_weight(ws["dz"], x_in, self.dw[name], self.db[name], accumulate=True)
            if i > 0:
                ops.linear_bwd_data(ws["dz"], self.w[name], ws["dh"])

    def train_step(self, x_pos, x_per, gt, hp=None):
        """One iteration of search.py:113-147: render -> zero_grad -> img2mse(robust_loss_adaptive) -> backward -> Adam ->
        LR rule (set after the step) -> global_step += 1.  hp (2 floats on the device: lr / (1 - b1^t), 1 / sqrt(1 - b2^t)):
        the Adam launches read their step-dependent scalars from there (npp_adam_step_dev), so that the launch sequence is
        identical from iteration to iteration and can be replayed as a HIP graph (ProposalRanker.fit_candidate)."""
        B = x_per.shape[0]
        pred = self.forward(x_pos, x_per)
        ws = self._ws[B]
        self.loss_buf.zero_()
        self.dlatent.zero_()
        ops.pixel_loss(pred, gt, None, self.latents, self.spline, self.n_knots, self.x_scale, 1.0, self.loss_buf, ws["dpred"], self.dlatent,
                       quad=self.quad)
        self.backward(B)
        if hp is None:
            self.opt_step += 1
            ops.adam_step(self.params, self.m, self.v, self.grad, 1, self.n_params, self.lr, self.opt_step)
            ops.adam_step(self.latents, self.lat_m, self.lat_v, self.dlatent, 1, 6, self.lr, self.opt_step)
            self.advance_clock()
        else:
            ops.adam_step_dev(self.params, self.m, self.v, self.grad, 1, self.n_params, hp)
            ops.adam_step_dev(self.latents, self.lat_m, self.lat_v, self.dlatent, 1, 6, hp)
        return self.loss_buf

    def advance_clock(self):
        """The host half of an optimiser step: the LR rule of search.py:139-147 (set AFTER the step) and global_step += 1."""
        self.lr = self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 100)))
        self.global_step += 1

    def adam_scalars(self, n_steps, b1=0.9, b2=0.999):
        """(n_steps, 2) float32: [lr_t / (1 - b1^t), 1 / sqrt(1 - b2^t)] of the NEXT n_steps optimiser steps -- exactly the values
        npp_adam_step computes on the host from (lr, step) -- without advancing the clock."""
        out = np.zeros((n_steps, 2), np.float32)
        lr, gs = self.lr, self.global_step
        for i in range(n_steps):
            t = self.opt_step + 1 + i
            out[i, 0] = np.float32(float(lr) / (1.0 - b1 ** t))
            out[i, 1] = np.float32(1.0 / np.sqrt(1.0 - b2 ** t))
            lr = self.lrate * (0.1 ** (gs / (self.lrate_decay * 100)))
            gs += 1
        return out

    @torch.no_grad()
    def render(self, coords_yx, chunk=20000):
        out = []
        for j in range(0, coords_yx.shape[0], chunk):
            x_pos, x_per = self.embed(coords_yx[j:j + chunk])
            out.append(self.forward(x_pos, x_per).clone())
        return torch.cat(out, 0)


class NPPNetLightBatch:
    """C candidates' NPP_Net_light advanced TOGETHER: every layer of every candidate in one launch (npp_linear_*_batched: the
    candidate is a grid dimension), so that an iteration of the whole candidate set is the ~35 launches one candidate's was -- 9 x
    128 workgroups per launch instead of 128 on a 256-CU chip.  The candidates share the pixel rows of every iteration
    (ProposalRanker._pixel_draws) and therefore x_pos and the colours; x_per (their lattice) and all weights are their own.
    State lives in stacked blobs (C, n_pad); .nets are ordinary NPPNetLight objects over the rows (render / score / state_dict)."""

    def __init__(self, cands, freqs, res, params, W=256, D=4, device="cuda", lrate=5e-4, lrate_decay=500, fused=None, precision=None,
                 loss_type="robust_loss_adaptive"):
        """fused (default: NPP_LIGHT_FUSED != 0 and the topology is the searched one, D = 4 / W = 256 / 42 + 20 input columns): forward and
        data-gradient chains as ONE launch each over all candidates (csrc/npp_light.hip) instead of one launch per layer.
        precision: "fp32" (exact fp32 MFMA everywhere) or "bf16" (csrc/npp_light16.hip: bf16 operands, fp32 accumulation, fp32 master
        weights and Adam -- the numeric contract of the main loop's MLP; fused topology, at most 16 candidates, batches of a multiple of
        64 rows; anything else falls back to fp32).  Default: "fp32"."""
        import os
        self.device = ops.select_device(device)
        self.quad = ops.quad_coef(loss_type)              # --loss_type: 0 = adaptive; > 0: 'l2' / 'robust_loss' (no latent gradient)
        self.C, self.W, self.D = len(cands), int(W), int(D)
        C = self.C
        in_pos = 2 * (1 + 2 * len(np.asarray(freqs[0] if isinstance(freqs, list) else freqs).reshape(-1)))
        self.layout = light_layout(self.W, self.D, in_pos, 20)
        self.in_pos = in_pos
        self.kpos = _stored_cols(self.W + in_pos)
        n = sum(r * _stored_cols(c) + r for _, r, c in self.layout)
        self.n_params, self.n_pad = n, (n + 3) // 4 * 4                 # rows 16-byte aligned: the flat Adam launch stays vectorised
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)      # noqa: E731
        self.params, self.grad, self.m, self.v = z(C, self.n_pad), z(C, self.n_pad), z(C, self.n_pad), z(C, self.n_pad)
        self.latents, self.lat_m, self.lat_v = z(C, 6), z(C, 6), z(C, 6)
        self.nets = []
        self._dl_c = z(C, 6)                                            # latent gradients: consumed AND cleared by the Adam launch
        self._loss2, self._li = z(2, C), 0                              # loss words, two sets: the idle one is cleared by the Adam launch
        for ci, (angles_deg, periods) in enumerate(cands):
            st = dict(params=self.params[ci, :n], grad=self.grad[ci, :n], m=self.m[ci, :n], v=self.v[ci, :n], latents=self.latents[ci],
                      lat_m=self.lat_m[ci], lat_v=self.lat_v[ci], dlatent=self._dl_c[ci], loss_buf=self._loss2[0, ci:ci + 1])
            # (multi-image sets: res / freqs may be lists, one entry per candidate -- every candidate another image's fit)
            res_c = res[ci] if isinstance(res, list) else res
            freqs_c = freqs[ci] if isinstance(freqs, list) else freqs
            self.nets.append(NPPNetLight(angles_deg, periods, freqs_c, res_c, params, W=W, D=D, device=self.device, lrate=lrate,
                                         lrate_decay=lrate_decay, storage=st, loss_type=loss_type))
        self.w, self.b, self.dw, self.db = {}, {}, {}, {}
        off = 0
        for name, r, c in self.layout:
            cs = _stored_cols(c)
            self.w[name], self.dw[name] = self.params[:, off:off + r * cs].unflatten(1, (r, cs)), self.grad[:, off:off + r * cs].unflatten(1, (r, cs))
            off += r * cs
            self.b[name], self.db[name] = self.params[:, off:off + r], self.grad[:, off:off + r]
            off += r
        n0 = self.nets[0]
        self.spline, self.n_knots, self.x_scale = n0.spline, n0.n_knots, n0.x_scale
        self._ws = {}
        can_fuse = self.W == 256 and self.D == 4 and in_pos == 42
        if fused is None:
            fused = can_fuse
        if fused and not can_fuse:
            raise ValueError("the fused NPP_Net_light chains are built for D = 4, W = 256, 42 positional and 20 periodic input columns")
        self.fused = bool(fused)
        # comparator switches of tests/test_gpu_light.py: seven weight-gradient launches instead of the grouped one; pack + Adam + clear
        # as separate launches
        self.grouped_wgrad, self.fused_adam = True, self.fused
        self._pack_valid = self._pack16_valid = False
        if precision is None:
            precision = "fp32"                # (the path pinned to the reference's trajectories g10 / g10c / g10d)
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision {precision!r}: 'fp32' or 'bf16'")
        self.bf16 = precision == "bf16" and self.fused and self.fused_adam and self.C <= 16
        if self.fused:
            from ._lib import LightDesc, lib
            order = [f"periodic_linears.{i}" for i in range(4)] + ["pos_linears.0", "feature_linear1", "rgb_linear"]
            shapes = {name: (r, c) for name, r, c in self.layout}
            d = LightDesc()
            for i, name in enumerate(order):
                d.w_off[i] = self.w[name].storage_offset() - self.params.storage_offset()
                d.b_off[i] = self.b[name].storage_offset() - self.params.storage_offset()
                d.n_out[i], d.n_in[i] = shapes[name]
                d.ld[i] = self.w[name].shape[2]
            self._desc = d
            L = lib()
            self._pack = z(C, int(L.npp_light_pack_floats()))
            self._srow = [int(L.npp_light_stash_row(i)) for i in range(8)]       # z0 z1 z2 z3 hp zp xperT | rows
            self._drow = [int(L.npp_light_dstash_row(i)) for i in range(8)]      # dz0 dz1 dz2 dz3 df1 dzp drawT | rows

    # ---- the 16-bit chains (csrc/npp_light16.hip)
    def _work16(self, B):
        ws = self._ws.get(("bf16", B))
        if ws is None:
            C = self.C
            pb, ab, db = ops.light16_sizes(B)
            u8 = lambda nb: torch.zeros(C, nb, dtype=torch.uint8, device=self.device)          # noqa: E731
            n_wg = B // 64
            # split-K of the weight-gradient launch.  Its workgroups take a whole CU's LDS each (one per CU, 256 at a time) and cost
            # ~1 us per 32-row half step plus ~8 us of prologue / epilogue; 9 output tiles per candidate.  Measured, 9 candidates x 2048
            # rows: ksplit 1 / 2 / 4 / 8 = 66 / 42 / 56 / 64 us (4 and 8 need a second / third round of workgroups)
            cost = lambda k: -(-9 * C * k // 256) * (2 * -(-n_wg // k) + 8)                      # noqa: E731
            ks = min(range(1, min(8, n_wg) + 1), key=cost)
            if ops.DETERMINISTIC:
                # (round 6) the slab split fixes the summation order of the weight gradients: it must not depend on how many candidates
                # ride in the launch, or a candidate fitted alone and the same candidate in a stacked set differ in their last bits
                # (light.rank_images: the images of a rank searched together give the bits of the serial loop).  1024 rows per slab.
                ks = max(1, min(8, n_wg // 16))
            ws = dict(actF=u8(ab), dzF=u8(db), pred=torch.empty(C, B, 3, dtype=torch.float32, device=self.device),
                      gslabs=torch.zeros(C, ks, self.n_pad, dtype=torch.float32, device=self.device))
            if getattr(self, "_pack16", None) is None:
                self._pack16 = u8(pb)
            self._ws[("bf16", B)] = ws
        return ws

    def _train_step_bf16(self, x_pos, x_per, gt, idx=None):
        """train_step() on the 16-bit chains: forward -> data gradients with the pixel loss folded in -> ONE grouped split-K
        weight-gradient launch (partial sums by plain stores) -> Adam + bf16 re-pack: 4 launches for the whole candidate set."""
        multi = gt.dim() == 3                               # multi-image set: x_pos (C, n, 42), gt (C, B, 3), idx (C, B)
        B = gt.shape[1] if multi else gt.shape[0]
        ws = self._work16(B)
        if not self._pack16_valid:
            ops.light16_pack(self._desc, self.params, self._pack16)
        ops.light16_fwd(self._desc, self.params, self._pack16, x_per.contiguous(), x_pos.contiguous(), ws["actF"], ws["pred"], idx=idx)
        loss = self._loss2[self._li]
        part = None
        if self.quad > 0:                                   # non-adaptive pixel loss: its own launch, d pred handed to the chain
            assert not multi
            dp = ws.setdefault("dpred", torch.empty_like(ws["pred"]))
            ops.pixel_loss_quad(ws["pred"], gt, None, self.quad, 1.0, loss, dp)
            ops.light16_bwd(self._desc, self.params, self._pack16, ws["actF"], ws["pred"], dp, ws["dzF"])
        elif ops.DETERMINISTIC:
            # bit-reproducible fit: the blocks' loss / latent-gradient sums by plain stores, added in block order by the Adam launch
            part = ws.get("part")
            if part is None:
                part = ws["part"] = torch.zeros(self.C, B // 64, 8, dtype=torch.float32, device=self.device)
            ops.light16_bwd_det(self._desc, self.params, self._pack16, ws["actF"], ws["pred"], ws["dzF"], gt, self.latents, self.spline,
                                self.n_knots, self.x_scale, part)
        else:
            assert not multi
            ops.light16_bwd(self._desc, self.params, self._pack16, ws["actF"], ws["pred"], None, ws["dzF"],
                            loss_args=(gt, self.latents, self.spline, self.n_knots, self.x_scale, loss, self._dl_c))
        ops.light16_wgrad(self._desc, ws["actF"], ws["dzF"], B, ws["gslabs"])
        n0 = self.nets[0]
        step, lr = n0.opt_step + 1, n0.lr
        self._li ^= 1
        if part is not None:
            ops.light16_adam_pack_det(self._desc, self.params, self.m, self.v, self.n_params, ws["gslabs"], self._pack16, self.latents,
                                      self.lat_m, self.lat_v, self._dl_c, self._loss2[self._li], lr, step, part, loss)
        else:
            ops.light16_adam_pack(self._desc, self.params, self.m, self.v, self.n_params, ws["gslabs"], self._pack16, self.latents, self.lat_m,
                                  self.lat_v, self._dl_c, self._loss2[self._li], lr, step)
        self._pack16_valid, self._pack_valid = True, False
        for net in self.nets:
            net.opt_step = step
            net.advance_clock()
        return loss

    def fit_loop_bf16(self, x_pos_all, x_per_all, draws, gt_all, log=None):
        """draws.shape[0] iterations of _train_step_bf16 (iteration j: batch rows draws[j] (B int64) of the tables, targets gt_all[j] (B, 3)) with
        the host's share stripped down: the four entry points are called with arguments marshalled ONCE (only the two row pointers, the loss
        word and Adam's (lr, step) change), the nets' clocks are advanced together at the end.  The same launches with the same arguments
        as the step-by-step path -- 4 x ~9 us of Python per iteration were a fifth of the set's wall time.  log: list that receives a
        copy of every iteration's loss words."""
        import ctypes as C_
        from ._lib import lib
        n_it, B = draws.shape
        assert self.bf16 and B % 64 == 0 and draws.dtype == torch.int64 and draws.is_contiguous() and gt_all.is_contiguous()
        assert gt_all.shape == (n_it, B, 3) and gt_all.dtype == torch.float32 and x_per_all.is_contiguous() and x_pos_all.is_contiguous()
        ws = self._work16(B)
        if not self._pack16_valid:
            ops.light16_pack(self._desc, self.params, self._pack16)
        Lb = lib()
        vp = lambda t: C_.c_void_p(t.data_ptr())                                                # noqa: E731
        desc, st = C_.byref(self._desc), ops._stream()
        n_src, Cn = x_per_all.shape[1], self.C
        assert x_per_all.shape == (Cn, n_src, 20) and x_pos_all.shape == (n_src, 42)
        par, pst, pk, pks = vp(self.params), self.params.stride(0), vp(self._pack16), self._pack16.stride(0)
        act, acs, dz, dzs, pred = vp(ws["actF"]), ws["actF"].stride(0), vp(ws["dzF"]), ws["dzF"].stride(0), vp(ws["pred"])
        xper, xpos = vp(x_per_all), vp(x_pos_all)
        lat, latm, latv, dl, spl = vp(self.latents), vp(self.lat_m), vp(self.lat_v), vp(self._dl_c), vp(self.spline)
        m_, v_ = vp(self.m), vp(self.v)
        gs = ws["gslabs"]
        gsl, ks, ns = vp(gs), gs.shape[1], gs.shape[2]
        loss_p = [vp(self._loss2[0]), vp(self._loss2[1])]
        d0, dstep, g0, gstep = draws.data_ptr(), draws.stride(0) * 8, gt_all.data_ptr(), gt_all.stride(0) * 4
        n0 = self.nets[0]
        step, lr, gstp, li = n0.opt_step, n0.lr, n0.global_step, self._li
        fwd, bwd, wg, adam = Lb.npp_light16_fwd, Lb.npp_light16_bwd, Lb.npp_light16_wgrad, Lb.npp_light16_adam_pack
        nk, xs, npar = self.n_knots, self.x_scale, self.n_params
        det = ops.DETERMINISTIC                              # (round 6) the bit-reproducible launches: npp_light16_bwd_det / _adam_pack_det
        if det:
            part_t = ws.get("part")
            if part_t is None:
                part_t = ws["part"] = torch.zeros(Cn, B // 64, 8, dtype=torch.float32, device=self.device)
            part, n_part = vp(part_t), B // 64
            bwd_d, adam_d = Lb.npp_light16_bwd_det, Lb.npp_light16_adam_pack_det
        for j in range(n_it):
            rc = fwd(desc, par, pst, pk, pks, xper, xpos, C_.c_void_p(d0 + j * dstep), n_src, Cn, B, act, acs, pred, st)
            if det:
                rc = rc or bwd_d(desc, par, pst, pk, pks, act, acs, pred, C_.c_void_p(g0 + j * gstep), 0, lat, spl, nk, xs, part, Cn, B, dz, dzs, st)
            else:
                rc = rc or bwd(desc, par, pst, pk, pks, act, acs, pred, None, C_.c_void_p(g0 + j * gstep), lat, spl, nk, xs, loss_p[li], dl, Cn, B, dz, dzs, st)
            rc = rc or wg(desc, act, acs, dz, dzs, Cn, B, ks, gsl, ns, ks * ns, st)
            step += 1
            if det:
                rc = rc or adam_d(desc, par, m_, v_, pst, npar, Cn, gsl, ks, ns, ks * ns, pk, pks, lat, latm, latv, dl, loss_p[li ^ 1], lr, 0.9, 0.999,
                                  1e-8, step, part, n_part, loss_p[li], st)
            else:
                rc = rc or adam(desc, par, m_, v_, pst, npar, Cn, gsl, ks, ns, ks * ns, pk, pks, lat, latm, latv, dl, loss_p[li ^ 1], lr, 0.9, 0.999, 1e-8, step, st)
            if rc:
                ops.check(rc, "npp_light16_* (fit_loop_bf16)")
            if log is not None:
                log.append(self._loss2[li].clone())
            li ^= 1
            lr = n0.lrate * (0.1 ** (gstp / (n0.lrate_decay * 100)))                          # NPPNetLight.advance_clock
            gstp += 1
        self._li = li
        self._pack16_valid, self._pack_valid = True, False
        for net in self.nets:
            net.opt_step, net.lr, net.global_step = step, lr, gstp

    def invalidate_pack(self):
        """Call after writing parameters from outside (load_state_dict on a member net): the next fused step re-packs first."""
        self._pack_valid = self._pack16_valid = False

    def _work_fused(self, B):
        ws = self._ws.get(("fused", B))
        if ws is None:
            C = self.C
            f = lambda *s_: torch.empty((C,) + s_, dtype=torch.float32, device=self.device)      # noqa: E731
            ws = dict(stash=f(self._srow[7], B), dstash=f(self._drow[7], B), pred=f(B, 3), dpred=f(B, 3), draw=f(B, 3))
            self._ws[("fused", B)] = ws
        return ws

    def _train_step_fused(self, x_pos, x_per, gt, idx=None):
        """train_step() on the fused chains: pack -> forward (gathers the iteration's rows itself when idx is given) -> data gradients
        with the pixel loss folded in -> gradient clear -> ONE grouped weight-gradient launch over the feature-major stashes
        (NPP_LIGHT_GROUPED_WGRAD=0: seven) -> Adam: 6 launches for the whole candidate set."""
        multi = gt.dim() == 3                              # multi-image set: x_pos (C, n, 42), gt (C, B, 3), idx (C, B)
        B = gt.shape[1] if multi else gt.shape[0]
        if multi and not (ops.DETERMINISTIC and self.fused_adam and self.quad == 0):
            raise ValueError("multi-image candidate sets run the deterministic fused chains with the adaptive pixel loss")
        ws = self._work_fused(B)
        S, D_, sr, dr = ws["stash"], ws["dstash"], self._srow, self._drow
        if not self._pack_valid:                            # first iteration (or after invalidate_pack()): later ones get the packs from the Adam launch
            ops.light_pack(self._desc, self.params, self._pack)
            if self.fused_adam:
                self.grad.zero_()
        ops.light_fwd(self._desc, self.params, self._pack, x_per.contiguous(), x_pos.contiguous(), S, ws["pred"], idx=idx)
        loss = self._loss2[self._li]
        if self.quad > 0:
            ops.pixel_loss_quad(ws["pred"], gt, None, self.quad, 1.0, loss, ws["dpred"])
            ops.light_bwd(self._desc, self.params, self._pack, S, ws["pred"], ws["dpred"], ws["draw"], D_)
        elif ops.DETERMINISTIC and self.fused_adam:
            # bit-reproducible fit: the blocks' loss / latent-gradient sums by plain stores, added in block order by the Adam launch
            part = ws.get("part")
            if part is None:
                part = ws["part"] = torch.zeros(self.C, ops.light_part_blocks(self.C, B), 8, dtype=torch.float32, device=self.device)
            ops.light_bwd_det(self._desc, self.params, self._pack, S, ws["pred"], ws["draw"], D_, gt, self.latents, self.spline, self.n_knots,
                              self.x_scale, part)
            self._part = part
        else:
            ops.light_bwd(self._desc, self.params, self._pack, S, ws["pred"], None, ws["draw"], D_,
                          loss_args=(gt, self.latents, self.spline, self.n_knots, self.x_scale, loss, self._dl_c))
        if not self.fused_adam:
            self.grad.zero_()
        if self.grouped_wgrad:
            wsc = ws.get("wgrad_scratch", False)             # (False: not decided yet for this workspace)
            if wsc is False:
                wsc = ws["wgrad_scratch"] = (ops.light_wgrad_det_scratch(self.C, B, self.device)
                                             if ops.DETERMINISTIC and ops.tune("light_det") else None)
            ops.light_wgrad(self._desc, S, D_, self.grad, scratch=wsc)   # all seven layers, one launch
            return loss
        W = self.W
        for i in range(4):                                   # periodic_linears.i: x = x_per (row-major) or snake(z_{i-1}) (feature-major)
            name = f"periodic_linears.{i}"
            dz = D_[:, dr[i]:dr[i] + W]
            if i == 0:
                ops.linear_bwd_weight_strided(dz, S[:, sr[6]:sr[7]], self.dw[name], self.db[name], True, True)
            else:
                ops.linear_bwd_weight_strided(dz, S[:, sr[i - 1]:sr[i - 1] + W], self.dw[name], self.db[name], True, True, x_snake=True)
        ops.linear_bwd_weight_strided(D_[:, dr[4]:dr[4] + W], S[:, sr[3]:sr[3] + W], self.dw["feature_linear1"], self.db["feature_linear1"], True, True,
                                      x_snake=True)
        ops.linear_bwd_weight_strided(D_[:, dr[5]:dr[5] + W // 2], S[:, sr[4]:sr[5]], self.dw["pos_linears.0"], self.db["pos_linears.0"], True, True)
        ops.linear_bwd_weight_strided(ws["draw"], S[:, sr[5]:sr[6]], self.dw["rgb_linear"], self.db["rgb_linear"], False, True, x_snake=True)
        return loss

    def _work(self, B):
        ws = self._ws.get(B)
        if ws is None:
            C, W = self.C, self.W
            f = lambda *s_: torch.empty((C,) + s_, dtype=torch.float32, device=self.device)      # noqa: E731
            ws = dict(z=[f(B, W) for _ in range(self.D)], h=[f(B, W) for _ in range(self.D)],
                      hp=torch.zeros(C, B, self.kpos, dtype=torch.float32, device=self.device), zp=f(B, W // 2), ap=f(B, W // 2), raw=f(B, 3), pred=f(B, 3), dpred=f(B, 3), draw=f(B, 3), dap=f(B, W // 2), dzp=f(B, W // 2),
                      df1=f(B, W), dh=f(B, W), dz=f(B, W))
            self._ws[B] = ws
        return ws

    def train_step(self, x_pos, x_per, gt, idx=None):
        """One iteration of search.py:113-147 for every candidate: x_pos (B, in_pos) and gt (B, 3) shared, x_per (C, B, 20) -- or, with
        idx (B int64), the whole tables x_pos (n, in_pos) / x_per (C, n, 20) whose rows idx are this iteration's batch."""
        C, B = x_per.shape[0], (gt.shape[1] if gt.dim() == 3 else gt.shape[0])
        if gt.dim() == 3:                                  # multi-image set (ops.light_fwd / light_bwd_det multi forms; bf16: light16_*)
            assert self.fused and B % 32 == 0 and idx is not None
            if self.bf16 and B % 64 == 0:
                return self._train_step_bf16(x_pos, x_per, gt, idx)
            return self._adam(self._train_step_fused(x_pos, x_per, gt, idx))
        if self.bf16 and B % 64 == 0:
            return self._train_step_bf16(x_pos, x_per, gt, idx)
        if self.fused and B % 32 == 0:
            loss = self._train_step_fused(x_pos, x_per, gt, idx)
            return self._adam(loss)
        if idx is not None:
            x_pos, x_per = x_pos[idx], x_per[:, idx]
        ws, W, D = self._work(B), self.W, self.D
        rows = lambda t: t.view(C * B, t.shape[2])                                                 # noqa: E731
        # ---- forward (NPPNetLight.forward)
        h = x_per
        for i in range(D):
            name = f"periodic_linears.{i}"
            ops.linear_fwd_batched(h, self.w[name], self.b[name], _SNAKE, ws["h"][i], ws["z"][i])
            h = ws["h"][i]
        ops.linear_fwd_batched(h, self.w["feature_linear1"], self.b["feature_linear1"], 0, ws["hp"][:, :, :W])
        ws["hp"][:, :, W:W + self.in_pos] = x_pos
        ops.linear_fwd_batched(ws["hp"], self.w["pos_linears.0"], self.b["pos_linears.0"], _SNAKE, ws["ap"], ws["zp"])
        ops.linear_fwd_batched(ws["ap"], self.w["rgb_linear"], self.b["rgb_linear"], 0, ws["raw"])
        ops.act_fwd(ws["raw"], _SIGMOID, ws["pred"])
        # ---- loss (its accumulators were cleared by the previous iteration's Adam launch)
        loss = self._loss2[self._li]
        if self.quad > 0:
            ops.pixel_loss_quad(ws["pred"], gt, None, self.quad, 1.0, loss, ws["dpred"])
        else:
            ops.pixel_loss_batched(ws["pred"], gt, self.latents, self.spline, self.n_knots, self.x_scale, 1.0, loss, ws["dpred"], self._dl_c)
        # ---- backward (NPPNetLight.backward)
        self.grad.zero_()
        # (every hidden layer's activation backward rides in the epilogue of the data-gradient launch above it)
        ops.act_bwd(rows(ws["dpred"]), rows(ws["pred"]), _SIGMOID, rows(ws["draw"]))
        ops.linear_bwd_weight_batched(ws["draw"], ws["ap"], self.dw["rgb_linear"], self.db["rgb_linear"])
        ops.linear_bwd_data_batched(ws["draw"], self.w["rgb_linear"], ws["dzp"], zy=ws["zp"], act=_SNAKE)
        ops.linear_bwd_weight_batched(ws["dzp"], ws["hp"], self.dw["pos_linears.0"], self.db["pos_linears.0"])
        ops.linear_bwd_data_batched(ws["dzp"], self.w["pos_linears.0"], ws["df1"], in_used=W)
        ops.linear_bwd_weight_batched(ws["df1"], ws["h"][D - 1], self.dw["feature_linear1"], self.db["feature_linear1"])
        ops.linear_bwd_data_batched(ws["df1"], self.w["feature_linear1"], ws["dz"], zy=ws["z"][D - 1], act=_SNAKE)
        for i in range(D - 1, -1, -1):
            name = f"periodic_linears.{i}"
            dz = ws["dz"] if (D - 1 - i) % 2 == 0 else ws["dh"]
            ops.linear_bwd_weight_batched(dz, ws["h"][i - 1] if i > 0 else x_per, self.dw[name], self.db[name])
            if i > 0:
                ops.linear_bwd_data_batched(dz, self.w[name], ws["dh"] if dz is ws["dz"] else ws["dz"], zy=ws["z"][i - 1], act=_SNAKE)
        return self._adam(loss)

    def _adam(self, loss):
        """Adam over the stacked blobs (the candidates share the step count and the LR clock); pad columns have zero gradient."""
        n0 = self.nets[0]
        step, lr = n0.opt_step + 1, n0.lr
        self._li ^= 1
        if self.fused_adam:
            # optimizer.step() + zero_grad() + the packs of the next forward, one launch (csrc/npp_light.hip)
            part, self._part = getattr(self, "_part", None), None
            if part is not None:
                ops.light_adam_pack_det(self._desc, self.params, self.m, self.v, self.grad, self.n_params, self._pack, self.latents, self.lat_m,
                                        self.lat_v, self._dl_c, self._loss2[self._li], lr, step, part, loss)
            else:
                ops.light_adam_pack(self._desc, self.params, self.m, self.v, self.grad, self.n_params, self._pack, self.latents, self.lat_m, self.lat_v,
                                    self._dl_c, self._loss2[self._li], lr, step)
            self._pack_valid, self._pack16_valid = True, False
        else:
            ops.adam_step_net(self.params.view(-1), self.m.view(-1), self.v.view(-1), self.grad.view(-1), 1, self.params.numel(),
                              self.latents.view(-1), self.lat_m.view(-1), self.lat_v.view(-1), self._dl_c.view(-1), self._loss2[self._li], lr, step)
            self._pack_valid = self._pack16_valid = False
        for net in self.nets:
            net.opt_step = step
            net.advance_clock()
        return loss


def default_light_init(W=256, D=4, in_pos=42, in_per=20, seed=0):
    """The reference's construction order and torch default nn.Linear init (models/networks.py:199-214) after
    torch.manual_seed(seed) -- search.py:92 reseeds with 0 before every candidate, so all candidates start from the same
    weights.  Draws the unused modules too, to keep the generator in step."""
    with ops.RNG_LOCK:
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        # create_npp_net builds the position embedder FIRST (helpers.py:84): its Gaussian Fourier frequencies (embedder.py:26) come out of
        # the same global generator, so the network's init starts n_freq normal draws into the stream (pinned by g10c_light_init.npz, the
        # reference's own construction; until round 3 this function skipped them: a statistically equivalent but DIFFERENT start)
        torch.normal(mean=0.0, std=1.0, size=((in_pos // 2 - 1) // 2, 1))
        mods = {}
        for i in range(D):
            mods[f"periodic_linears.{i}"] = torch.nn.Linear(in_per if i == 0 else W, W)      # skips=[4] is never reached for D = 4
        mods["scale_linears.0"] = torch.nn.Linear(0 + W, W)
        mods["pos_linears.0"] = torch.nn.Linear(in_pos + W, W // 2)
        mods["feature_linear1"] = torch.nn.Linear(W, W)
        mods["feature_linear2"] = torch.nn.Linear(W, W)
        mods["alpha_linear"] = torch.nn.Linear(W, 1)
        mods["rgb_linear"] = torch.nn.Linear(W // 2, 3)
        torch.random.set_rng_state(g)
    return {f"{k}.{p}": getattr(m, p).detach().numpy().copy() for k, m in mods.items() for p in ("weight", "bias")}




class ProposalRanker:
    """The candidate loop of NPP_proposal/search.py:85-215 for one image: per candidate (angles, periods) a fresh
    NPP_Net_light is fitted for N_iters pixel-loss iterations on the known pixels, rendered over the pseudo-mask region
    and scored with perceptual_weight * LPIPS(use_robust=False) + contextual_weight * CX against the (known) content of
    that region; the top-k smallest scores win (search.py:215)."""

    def __init__(self, masked_img, i_train, i_val, device="cuda", N_iters=300, N_rand=2048, W=256, D=4, lrate=5e-4, lrate_decay=500,
                 perceptual_weight=30.0, contextual_weight=1.0, freqs=None, vgg19_state_dict=None, vgg16_state_dict=None,
                 lpips_lin_weights=None, rng_mode="reference", carry_latents=False, record_losses=False, precision=None,
                 loss_type="robust_loss_adaptive"):
        """precision: "fp32" | "bf16" | None (fp32): the arithmetic of the candidate fits (NPPNetLightBatch).
        carry_latents: in the reference the adaptive pixel loss is ONE module-level object (models/helpers.py:8) that every
        candidate's optimiser trains on (helpers.py:144), so candidate k + 1 starts from the latents candidate k left (fresh Adam
        moments) and the ranking depends on the order of the candidates.  False (default; SURVEY 3.3 / 8 e: candidates are independent
        units -- what lets them share launches and shard over GPUs): every candidate starts from the initial latents, as the first
        one does in the reference.  True: the reference's behaviour, candidates fitted one after the other (tests/golden
        g10d_light_fit.npz pins both).  record_losses: keep the per-iteration loss words (self.loss_log, one (N_iters, C) tensor per
        launch group)."""
        from .losses import ContextualLoss, LPIPS
        self.device = ops.select_device(device)
        self.img = torch.as_tensor(np.asarray(masked_img, np.float32)).to(self.device)               # (H,W,3)
        self.H, self.W_img = self.img.shape[:2]
        self.i_train = np.asarray(i_train).astype(np.int32)
        self.i_val = np.asarray(i_val).astype(np.int32)
        self.i_train_dev = torch.from_numpy(self.i_train).to(self.device)
        self.i_val_dev = torch.from_numpy(self.i_val).to(self.device)
        self.N_iters, self.N_rand, self.Wn, self.D = int(N_iters), int(N_rand), int(W), int(D)
        self.lrate, self.lrate_decay = lrate, lrate_decay
        self.pw, self.cw = float(perceptual_weight), float(contextual_weight)
        self.rng_mode = rng_mode
        self.carry_latents, self.record_losses, self.loss_log = bool(carry_latents), bool(record_losses), []
        self.precision, self.loss_type = precision, loss_type
        ops.quad_coef(loss_type)
        if freqs is None:                                      # embedder.py:26 after torch.manual_seed(0) (search.py:92)
            with ops.RNG_LOCK:
                g = torch.random.get_rng_state()
                torch.manual_seed(0)
                freqs = (torch.normal(mean=0.0, std=1.0, size=(10, 1)) * 10).reshape(-1).numpy()
                torch.random.set_rng_state(g)
        self.freqs = np.asarray(freqs, np.float32)
        self._draws, self._gt_all = None, None
        # the two score trunks: objects of this ranker (their activation buffers go with it; images searched side by side on several
        # host threads, run.search_all, score at the same time), device weights and MFMA packs shared by every trunk of the process
        # (losses.HipTrunk._packs) -- until round 5 the rankers shared the objects, which only paid while packing was per object
        self.percep = LPIPS(net="vgg", lin_weights=lpips_lin_weights, vgg_state_dict=vgg16_state_dict, device=self.device)
        self.cx = ContextualLoss(use_vgg=True, vgg_state_dict=vgg19_state_dict, device=self.device)

    def _pixel_draws(self):
        """(N_iters, n_rand) int64 on the device: the pixel rows of every iteration.  search.py:92-93 reseeds NumPy with 0 before EVERY
        candidate, so all candidates of an image walk the same index sequence: drawn once (np.random.choice(n_train, [n_rand],
        replace=False) per iteration -- a full permutation of the known pixels each, 1.3 ms of host time per iteration on a
        676 x 494 image, three times the device time of the iteration) and uploaded once, instead of per candidate and iteration."""
        if self._draws is None:
            n_train = self.i_train.shape[0]
            n_rand = min(self.N_rand, n_train)
            if self.rng_mode == "fast":
                g = np.random.default_rng(0)
                sel = [g.choice(n_train, n_rand, replace=False) for _ in range(self.N_iters)]
            else:
                from .host_rng import NativeRandomState                                              # np.random.RandomState(0)'s stream, GIL-free
                g = NativeRandomState(0)
                sel = [g.choice(n_train, size=[n_rand], replace=False) for _ in range(self.N_iters)]
            self._draws = torch.from_numpy(np.ascontiguousarray(np.stack(sel), np.int64)).to(self.device)   # once per image: plain copy
        return self._draws

    def _draw_stream(self, chunk=32):
        """The iterations' pixel rows and their colours as a stream of device chunks [(draws (n, B) int64, gt (n, B, 3))]: the first
        fit of an image starts on chunk 0 while a helper thread (the native generator releases the GIL) draws the later ones --
        300 shuffles of the known pixels are 50 ms of host time on a 676 x 494 image against 150 ms of device time for the whole
        candidate set.  The assembled table is cached for the image's later fits (_pixel_draws)."""
        if self._draws is not None:
            if self._gt_all is None:
                c_all = self.i_train_dev[self._draws.reshape(-1)].long()
                self._gt_all = self.img[c_all[:, 0], c_all[:, 1]].reshape(self._draws.shape[0], self._draws.shape[1], 3).contiguous()
            yield self._draws, self._gt_all
            return
        import queue
        import threading
        n_train = self.i_train.shape[0]
        n_rand = min(self.N_rand, n_train)
        q = queue.Queue()

        def produce():
            try:
                if self.rng_mode == "fast":
                    g = np.random.default_rng(0)
                    draw = lambda: g.choice(n_train, n_rand, replace=False)                           # noqa: E731
                else:
                    from .host_rng import NativeRandomState                                          # np.random.RandomState(0)'s stream
                    g = NativeRandomState(0)
                    draw = lambda: g.choice(n_train, size=[n_rand], replace=False)                   # noqa: E731
                for it0 in range(0, self.N_iters, chunk):
                    q.put(np.ascontiguousarray(np.stack([draw() for _ in range(min(chunk, self.N_iters - it0))]), np.int64))
                q.put(None)
            except BaseException as e:                                                               # surface generator errors in the caller
                q.put(e)
        threading.Thread(target=produce, name="npp-ranker-draws", daemon=True).start()
        parts_d, parts_g = [], []
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            d = torch.from_numpy(item).to(self.device)
            c = self.i_train_dev[d.reshape(-1)].long()
            gt = self.img[c[:, 0], c[:, 1]].reshape(d.shape[0], d.shape[1], 3).contiguous()
            parts_d.append(d)
            parts_g.append(gt)
            yield d, gt
        self._draws, self._gt_all = torch.cat(parts_d), torch.cat(parts_g)

    def fit_candidate(self, angles_deg, periods, params=None, use_graph=None, fused=None):
        """search.py:85-147 for one candidate.
        fused (default unless NPP_LIGHT_FUSED=0 / use_graph / a topology the chains are not built for): the fused forward and
        data-gradient chains with a candidate set of one (NPPNetLightBatch: 13 launches per iteration, 0.25 ms); else the
        layer-by-layer path below: ~40 small dependent launches on 2048 rows, 0.39 ms per iteration.
        use_graph=True: iterations 2 .. N replay ONE captured HIP graph (torch.cuda.CUDAGraph: the same
        kernels in the same order; the iteration's inputs -- pixel-row indices, Adam's step-dependent scalars -- are read from
        fixed device buffers).  Built in round 3 on the hypothesis that the host's enqueue rate bounded the loop; MEASURED: it does
        not -- eager 0.400 ms per iteration, graph replay 0.417-0.44 (profiles/r03_rejected_experiments.txt): the device time of the
        twenty 2048 x 256 x 256 exact-fp32 GEMM launches (14-18 us each) is the bound.  Kept as an option (parity-tested), off."""
        use_graph = bool(use_graph)
        if fused is None:
            fused = (not use_graph and self.Wn == 256 and self.D == 4
                     and len(self.freqs) == 10 and min(self.N_rand, self.i_train.shape[0]) % 32 == 0)
        if fused:
            return self._fit_candidates_batched([(angles_deg, periods)], init=params)[0]
        draws = self._pixel_draws()
        net = NPPNetLight(angles_deg, periods, self.freqs, (self.H, self.W_img),
                          params if params is not None else default_light_init(self.Wn, self.D), W=self.Wn, D=self.D,
                          device=self.device, lrate=self.lrate, lrate_decay=self.lrate_decay, loss_type=self.loss_type)
        x_pos_all, x_per_all = net.embed(self.i_train_dev)                                           # search.py:104-108 tables

        def eager(it):
            idx = draws[it]
            c = self.i_train_dev[idx]
            gt = self.img[c[:, 0].long(), c[:, 1].long()].contiguous()
            net.train_step(x_pos_all[idx].contiguous(), x_per_all[idx].contiguous(), gt)
        if not use_graph or self.N_iters < 3:
            for it in range(self.N_iters):
                eager(it)
            return net
        eager(0)                                                   # iteration 1 eagerly: first-call setup, workspaces
        n_rest = self.N_iters - 1
        # per-iteration inputs of the captured launches, one table row per iteration: [pixel-row indices | Adam scalars (as int64 bits)]
        hp_tab = torch.from_numpy(net.adam_scalars(n_rest)).to(self.device)
        idx_s = draws[1].clone()
        hp_s = hp_tab[0].clone()

        def body():
            c = self.i_train_dev[idx_s]
            gt = self.img[c[:, 0].long(), c[:, 1].long()].contiguous()
            net.train_step(x_pos_all[idx_s].contiguous(), x_per_all[idx_s].contiguous(), gt, hp=hp_s)
        side = torch.cuda.Stream(self.device)                      # capture needs a non-default stream; one un-captured pass warms its pool
        side.wait_stream(torch.cuda.current_stream(self.device))
        snap = [t.clone() for t in (net.params, net.m, net.v, net.latents, net.lat_m, net.lat_v)]
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        for t, s_ in zip((net.params, net.m, net.v, net.latents, net.lat_m, net.lat_v), snap):     # the warm-up pass was not an iteration
            t.copy_(s_)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            body()
        for j in range(n_rest):
            if j:
                idx_s.copy_(draws[1 + j])
                hp_s.copy_(hp_tab[j])
            graph.replay()
            net.opt_step += 1
            net.advance_clock()
        self._graph_keep = (graph, idx_s, hp_s, hp_tab)            # alive until the replays have run
        return net

    def fit_candidates(self, cands, n_streams=8, batched=None):
        """fit_candidate() for several candidates at once.  The fits are independent and one fit's iteration is ~40 small dependent
        launches that leave the chip mostly idle (128 workgroups each), so:
        batched (default; NPP_LIGHT_BATCH=0 turns it off): every launch carries ALL the candidates (NPPNetLightBatch, the candidate
        is a grid dimension of the dense-layer kernels) -- an iteration of the whole set costs about what one candidate's did;
        else: the candidates advance together, iteration by iteration, each on one of n_streams side streams (round 2; bounded by
        the host's enqueue rate: 9 x 40 launches per iteration).
        Same arithmetic per candidate as the serial form up to the summation order of the split weight-gradient contractions; the
        pixel rows and their colours (identical for every candidate, see _pixel_draws) are gathered once."""
        main = torch.cuda.current_stream(self.device)
        if batched is None:
            batched = True
        if self.carry_latents:                                       # the reference's shared adaptive_pix: strictly one after the other
            nets, lat = [], None
            for cand in cands:
                nets += self._fit_candidates_batched([cand], latents0=lat)
                lat = nets[-1].latents.clone()
            return nets
        if batched:
            return self._fit_candidates_batched(cands)
        draws = self._pixel_draws()
        c_all = self.i_train_dev[draws.reshape(-1)].long()
        gt_all = self.img[c_all[:, 0], c_all[:, 1]].reshape(draws.shape[0], draws.shape[1], 3).contiguous()
        nets, tabs = [], []
        for angles_deg, periods in cands:
            net = NPPNetLight(angles_deg, periods, self.freqs, (self.H, self.W_img), default_light_init(self.Wn, self.D), W=self.Wn, D=self.D,
                              device=self.device, lrate=self.lrate, lrate_decay=self.lrate_decay, loss_type=self.loss_type)
            tabs.append(net.embed(self.i_train_dev))
            net._work(draws.shape[1])            # workspace allocated on the MAIN stream (score() / render() use it there after the join;
            nets.append(net)                      # blocks first touched on a side stream would return to that stream's pool)
        streams = [torch.cuda.Stream(self.device) for _ in range(max(1, min(int(n_streams), len(nets))))]
        for st in streams:
            st.wait_stream(main)
        for it in range(self.N_iters):
            idx, gt = draws[it], gt_all[it]
            for j, net in enumerate(nets):
                with torch.cuda.stream(streams[j % len(streams)]):
                    net.train_step(tabs[j][0][idx], tabs[j][1][idx], gt)
        for st in streams:
            main.wait_stream(st)
        return nets

    def _fit_candidates_batched(self, cands, group=16, init=None, latents0=None):
        """All candidates of the image in ONE launch sequence (NPPNetLightBatch), `group` at a time: the default of fit_candidates.
        latents0 (6): the adaptive-loss latents every candidate of this call starts from (carry_latents).  The pixel rows come from
        _draw_stream(): drawn on a helper thread under the first group's iterations, cached for the rest."""
        nets = []
        if init is None:
            init = default_light_init(self.Wn, self.D)
        for g0 in range(0, len(cands), group):
            part = cands[g0:g0 + group]
            batch = NPPNetLightBatch(part, self.freqs, (self.H, self.W_img), init, W=self.Wn, D=self.D, device=self.device,
                                     lrate=self.lrate, lrate_decay=self.lrate_decay, precision=self.precision, loss_type=self.loss_type)
            tabs = [net.embed(self.i_train_dev) for net in batch.nets]                                # search.py:104-108 tables
            x_pos_all = tabs[0][0]                                                                   # the same for every candidate
            x_per_all = torch.stack([t[1] for t in tabs])                                            # (C, n_train, 20)
            del tabs
            if latents0 is not None:
                batch.latents.copy_(latents0.reshape(1, 6).expand(len(part), 6))
            log = []
            for draws, gt_all in self._draw_stream():
                if batch.bf16 and batch.quad == 0 and draws.shape[1] % 64 == 0:
                    batch.fit_loop_bf16(x_pos_all, x_per_all, draws.contiguous(), gt_all.contiguous(), log if self.record_losses else None)
                    continue
                for j in range(draws.shape[0]):
                    loss = batch.train_step(x_pos_all, x_per_all, gt_all[j], idx=draws[j])
                    if self.record_losses:
                        log.append(loss.clone())
            if self.record_losses:
                self.loss_log.append(torch.stack(log))
            self._batch_keep = batch                                                                 # the nets are views of its blobs
            nets.extend(batch.nets)
        return nets

    @torch.no_grad()
    def score(self, net):
        """search.py:152-197."""
        pred = net.render(self.i_val_dev)
        vy, vx = self.i_val_dev[:, 0].long(), self.i_val_dev[:, 1].long()
        canvas_p = torch.zeros_like(self.img)
        canvas_g = torch.zeros_like(self.img)
        canvas_p[vy, vx] = pred
        canvas_g[vy, vx] = self.img[vy, vx]
        h0, h1, w0, w1 = int(vy.min()), int(vy.max()), int(vx.min()), int(vx.max())                   # [min, max) like :180-187
        p = canvas_p[h0:h1, w0:w1].permute(2, 0, 1)[None].contiguous()
        g = canvas_g[h0:h1, w0:w1].permute(2, 0, 1)[None].contiguous()
        lp = self.percep.plain(p, g, normalize=False)
        cx = self.cx(p, g)
        return float(lp[0]) * self.pw + float(cx) * self.cw, float(lp[0]), float(cx)

    def rank(self, candidates, topk=10):
        """candidates: [(angles (2,), periods (2,), shifts)] -> (sorted distances, order, per-candidate details).
        Under an initialised torch.distributed process group the candidates are sharded over the ranks
        (parallel.shard_units: independent fits, no data-path collective) and the (score, lpips, cx) rows are
        all-gathered once at the end; every rank returns the same ranking."""
        import torch.distributed as dist
        from .parallel import shard_units, gather_unit_scalars
        # the sharded branch runs whenever a process group exists (a world of ONE rank included: shard_units / the all_gather of
        # gather_unit_scalars are then the identity, but it is the same code path the 8-GPU node takes)
        multi = dist.is_available() and dist.is_initialized()
        self.last_rank_collective = None
        if multi and dist.get_world_size() > 1 and self.carry_latents:
            raise ValueError("carry_latents chains the candidates through one set of adaptive-loss latents: it cannot be sharded over ranks")
        mine = shard_units(len(candidates), dist.get_rank(), dist.get_world_size()) if multi else range(len(candidates))
        nets = self.fit_candidates([(candidates[ci][0], candidates[ci][1]) for ci in mine])
        details = [self.score(net) for net in nets]
        if multi:
            t = torch.tensor(details, dtype=torch.float32, device=self.device).reshape(-1, 3)
            details = [tuple(r) for r in gather_unit_scalars(t, len(candidates)).cpu().tolist()]
            self.last_rank_collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "rows": len(details)}
        d = np.array([x[0] for x in details])
        order = np.argsort(d, kind="stable")[:min(topk, len(d))]
        return d[order], order, details


@torch.no_grad()
def rank_images(rankers, cand_lists, topk=10):
    """ProposalRanker.rank for SEVERAL images of one rank at once: candidate k of every image rides in one launch
    sequence -- an NPPNetLightBatch whose "candidates" are the images' k-th candidates, each with its own lattice table, positional
    table, pixel rows and targets (ops.light_fwd / light_bwd_det multi forms).  The reference walks images (run_completion.sh:8-14)
    and candidates (NPP_proposal/search.py:85-215) one after the other; with its shared adaptive-loss latents (carry_latents, the
    single-rank default) the candidates of ONE image are a chain, but the chains of different images are independent: 9 x 300
    iterations of ~0.1 ms are then paid once for all the images instead of once per image.
    Per image the arithmetic is that of the serial deterministic loop (same launches, same block and summation order per candidate:
    identical bits).  Images must share the fit's hyper-parameters; they are grouped by batch rows (min(N_rand, known pixels)).
    -> per image (sorted distances, order, per-candidate details), like ProposalRanker.rank."""
    if not rankers:
        return []
    r0 = rankers[0]
    for rk in rankers:
        if (rk.N_iters, rk.Wn, rk.D, rk.lrate, rk.lrate_decay, rk.loss_type, rk.carry_latents, rk.precision, str(rk.device)) != \
           (r0.N_iters, r0.Wn, r0.D, r0.lrate, r0.lrate_decay, r0.loss_type, r0.carry_latents, r0.precision, str(r0.device)):
            raise ValueError("rank_images: the images' candidate fits must share their hyper-parameters and device")
    if not ops.DETERMINISTIC or ops.quad_coef(r0.loss_type) > 0:
        return [rk.rank(c, topk=topk) for rk, c in zip(rankers, cand_lists)]       # (the multi-image launches are the deterministic fused ones)
    bf16 = r0.precision == "bf16"
    n_img = len(rankers)
    details = [[None] * len(c) for c in cand_lists]
    groups = {}
    for i, rk in enumerate(rankers):
        groups.setdefault(min(rk.N_rand, rk.i_train.shape[0]), []).append(i)
    init = default_light_init(r0.Wn, r0.D)
    for B, members in groups.items():
        if B % (64 if bf16 else 32) or len(members) == 1 or (bf16 and len(members) > 16):   # a batch the fused chains do not take, or nothing to stack: the image's own loop
            for i in members:
                d, order, det = rankers[i].rank(cand_lists[i], topk=topk)
                details[i] = det
            continue
        # (N_iters, B) int64 each: 300 shuffles of the image's known pixels from the reference's NumPy stream -- 60 ms of host time per
        # image, GIL-free (host_rng.NativeRandomState), so the images' streams are drawn side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(8, len(members))) as pool:
            draws = dict(zip(members, pool.map(lambda i: rankers[i]._pixel_draws(), members)))
        gts, xpos = {}, {}
        for i in members:
            rk = rankers[i]
            c_all = rk.i_train_dev[draws[i].reshape(-1)].long()
            gts[i] = rk.img[c_all[:, 0], c_all[:, 1]].reshape(rk.N_iters, B, 3)
        n_max = max(rankers[i].i_train.shape[0] for i in members)
        lat = {i: None for i in members}
        act_prev = idx_all = gt_all = None
        for k in range(max(len(cand_lists[i]) for i in members)):
            act = [i for i in members if k < len(cand_lists[i])]
            batch = NPPNetLightBatch([(cand_lists[i][k][0], cand_lists[i][k][1]) for i in act], [rankers[i].freqs for i in act],
                                     [(rankers[i].H, rankers[i].W_img) for i in act], init, W=r0.Wn, D=r0.D, device=r0.device,
                                     lrate=r0.lrate, lrate_decay=r0.lrate_decay, precision="bf16" if bf16 else "fp32", loss_type=r0.loss_type)
            C = len(act)
            x_pos = torch.zeros(C, n_max, 42, dtype=torch.float32, device=r0.device)
            x_per = torch.zeros(C, n_max, 20, dtype=torch.float32, device=r0.device)
            for j, i in enumerate(act):
                tp, tr = batch.nets[j].embed(rankers[i].i_train_dev)                                    # search.py:104-108 tables
                if i not in xpos:
                    xpos[i] = tp                                                                      # (the positional table is the image's, not the candidate's)
                x_pos[j, :tp.shape[0]] = xpos[i]
                x_per[j, :tr.shape[0]] = tr
                if r0.carry_latents and lat[i] is not None:
                    batch.latents[j].copy_(lat[i])
            if act != act_prev:                                                                       # (the same images as long as all have a k-th candidate)
                idx_all = torch.stack([draws[i] for i in act], 1).contiguous()                        # (N_iters, C, B)
                gt_all = torch.stack([gts[i] for i in act], 1).contiguous()                           # (N_iters, C, B, 3)
                act_prev = act
            for it in range(r0.N_iters):
                batch.train_step(x_pos, x_per, gt_all[it], idx=idx_all[it])
            for j, i in enumerate(act):
                lat[i] = batch.latents[j].clone()
                details[i][k] = rankers[i].score(batch.nets[j])
            del batch, x_pos, x_per
    out = []
    for i in range(n_img):
        d = np.array([x[0] for x in details[i]])
        order = np.argsort(d, kind="stable")[:min(topk, len(d))]
        out.append((d[order], order, details[i]))
    return out

