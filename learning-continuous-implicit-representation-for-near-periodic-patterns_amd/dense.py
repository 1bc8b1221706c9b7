"""NPP_Net / NPP_Net_top1 of ANY width, depth, skip list and activation on the generic dense-layer kernels
(csrc/npp_linear.hip: exact fp32 MFMA).  The fused chain kernels are specialised for the BASELINE configuration
(D = 8, W = 256, snake); the reference's own defaults are wider (options/arg_config.py: netwidth 512) and it also offers
activation='relu' -- this module serves those through one launch per layer: functional drop-in, not the fast path.

Mirrors models/networks.py:8-95 (NPP_Net) and :100-173 (NPP_Net_top1): same constructor arguments, same parameter names
and shapes (ordinary per-layer nn.Parameters, so torch.optim.Adam(model.parameters()) is exactly the reference's optimiser),
forward(None, x_periodic) -> raw (B, output_ch).
"""
import torch
import torch.nn as nn

from . import ops

_ACT = {"snake": (1, 1), "relu": (2, 4)}          # (forward code, act_bwd code)


class _DenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_periodic, net, *params):
        ctx.net = net
        ctx.bufs = net._run_forward(x_periodic.detach().float().contiguous(), [p.detach() for p in params])
        return ctx.bufs["out"].clone()

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.net._run_backward(ctx.bufs, gout.contiguous().float())
        ctx.bufs = None
        return (None, None) + tuple(grads)


class _DenseBase(nn.Module):
    def __init__(self, E1, Ea, D, W, skips, activation, output_ch, device):
        super().__init__()
        if activation not in _ACT:
            raise NotImplementedError(f"activation {activation!r}")
        self.E1, self.Ea, self.D, self.W, self.skips, self.output_ch = int(E1), int(Ea), int(D), int(W), list(skips), int(output_ch)
        self.act_f, self.act_b = _ACT[activation]
        self.multi = self.Ea > 0
        dev = torch.device(device)
        # construction order of the reference (networks.py:40-49 / :128-140): default nn.Linear init from the global generator
        self.periodic_linears = nn.ModuleList([nn.Linear(self.E1, W)] + [nn.Linear(W + self.E1 if i in self.skips else W, W)
                                                                         for i in range(D - 1)])
        if self.multi:
            self.scale_linears = nn.ModuleList([nn.Linear(self.Ea + W, W)])
        self.pos_linears = nn.ModuleList([nn.Linear(2 * W if self.multi else W, W // 2)])
        self.feature_linear1 = nn.Linear(W, W)
        self.feature_linear2 = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, output_ch)
        self.to(dev)

    def _used(self):
        """(name, module) of the layers the forward uses, in evaluation order; alpha_linear (and feature_linear2 for K = 1)
        are constructed but never used, exactly as in the reference (no gradient)."""
        L = [(f"periodic_linears.{i}", m) for i, m in enumerate(self.periodic_linears)]
        L.append(("feature_linear1", self.feature_linear1))
        if self.multi:
            L += [("scale_linears.0", self.scale_linears[0]), ("feature_linear2", self.feature_linear2)]
        L += [("pos_linears.0", self.pos_linears[0]), ("rgb_linear", self.rgb_linear)]
        return L

    def forward(self, x, x_periodic):
        if x_periodic.shape[1] != self.E1 + self.Ea:
            raise ValueError(f"x_periodic has {x_periodic.shape[1]} columns, this net takes {self.E1 + self.Ea}")
        params = [p for _, m in self._used() for p in (m.weight, m.bias)]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _DenseFunction.apply(x_periodic, self, *params)
        with torch.no_grad():
            return self._run_forward(x_periodic.float().contiguous(), [p.detach() for p in params])["out"]

    # ---- networks.py:56-95 / :145-173 layer by layer -------------------------------------------------
    def _run_forward(self, xp, params):
        B, W, E1, D = xp.shape[0], self.W, self.E1, self.D
        dev = xp.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)          # noqa: E731
        P = {}
        for (name, _), i in zip(self._used(), range(0, len(params), 2)):
            P[name] = (params[i].contiguous(), params[i + 1].contiguous())
        emb0 = xp[:, :E1]
        b = dict(xp=xp, P=P, z=[], inp=[])
        h = emb0
        for i in range(D):
            w_, b_ = P[f"periodic_linears.{i}"]
            z = f(B, W)
            if i in self.skips:                                                   # h = cat[input_periodic, h] (:70-71)
                cat = f(B, E1 + W)
                cat[:, :E1].copy_(emb0)
                out = cat[:, E1:]
            else:
                cat, out = None, f(B, W)
            b["inp"].append(h)
            ops.linear_fwd(h, w_, b_, self.act_f, out, z)
            b["z"].append(z)
            h = cat if cat is not None else out
        b["h_last"] = h
        if self.multi:
            scat = f(B, W + self.Ea)                                              # cat[feature1, input_periodic_aux] (:76)
            pcat = f(B, 2 * W)                                                    # cat[feature1, feature2] (:85)
            ops.linear_fwd(h, *P["feature_linear1"], 0, scat[:, :W])
            scat[:, W:].copy_(xp[:, E1:])
            pcat[:, :W].copy_(scat[:, :W])
            zs, s = f(B, W), f(B, W)
            ops.linear_fwd(scat, *P["scale_linears.0"], self.act_f, s, zs)
            ops.linear_fwd(s, *P["feature_linear2"], 0, pcat[:, W:])
            b.update(scat=scat, zs=zs, s=s)
        else:
            pcat = f(B, W)
            ops.linear_fwd(h, *P["feature_linear1"], 0, pcat)
        zp, ap, out = f(B, W // 2), f(B, W // 2), f(B, self.output_ch)
        ops.linear_fwd(pcat, *P["pos_linears.0"], self.act_f, ap, zp)
        ops.linear_fwd(ap, *P["rgb_linear"], 0, out)
        b.update(pcat=pcat, zp=zp, ap=ap, out=out)
        return b

    def _run_backward(self, b, gout):
        B, W, E1, D = gout.shape[0], self.W, self.E1, self.D
        dev = gout.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)          # noqa: E731
        P, G = b["P"], {}

        def wgrad(name, dz, x):
            w_, b_ = P[name]
            dw, db = torch.empty_like(w_), torch.empty_like(b_)
            ops.linear_bwd_weight(dz, x, dw, db)
            G[name] = (dw, db)
        wgrad("rgb_linear", gout, b["ap"])
        dap, dzp = f(B, W // 2), f(B, W // 2)
        ops.linear_bwd_data(gout, P["rgb_linear"][0], dap)
        ops.act_bwd(dap, b["zp"], self.act_b, dzp)
        wgrad("pos_linears.0", dzp, b["pcat"])
        dpcat = f(*b["pcat"].shape)
        ops.linear_bwd_data(dzp, P["pos_linears.0"][0], dpcat)
        df1 = dpcat[:, :W]
        if self.multi:
            df2 = dpcat[:, W:]
            wgrad("feature_linear2", df2, b["s"])
            ds, dzs = f(B, W), f(B, W)
            ops.linear_bwd_data(df2, P["feature_linear2"][0], ds)
            ops.act_bwd(ds, b["zs"], self.act_b, dzs)
            wgrad("scale_linears.0", dzs, b["scat"])
            ops.linear_bwd_data(dzs, P["scale_linears.0"][0][:, :W], df1, accumulate=True)     # + the S path into feature1
        wgrad("feature_linear1", df1, b["h_last"])
        hl_w = b["h_last"].shape[1]
        dh_full = f(B, hl_w)
        ops.linear_bwd_data(df1, P["feature_linear1"][0], dh_full)
        dh = dh_full[:, hl_w - W:]                               # if the last trunk layer is a skip layer its output sits after emb0
        for i in range(D - 1, -1, -1):
            name = f"periodic_linears.{i}"
            dz = f(B, W)
            ops.act_bwd(dh, b["z"][i], self.act_b, dz)
            wgrad(name, dz, b["inp"][i])
            if i > 0:
                w_ = P[name][0]
                cols = w_.shape[1]
                dh = f(B, W)
                ops.linear_bwd_data(dz, w_[:, cols - W:], dh)    # gradient w.r.t. the hidden block of a (possibly concatenated) input
        out = []
        for name, _ in self._used():
            out += list(G[name])
        return out


class DenseNPPNet(_DenseBase):
    """models/networks.py:8-95 for any (D, W, skips, activation)."""

    def __init__(self, input_ch_periodic, input_ch_periodic_aux, freq_scales, freq_offsets, angle_offsets, D=8, W=256, freq_nerf=3,
                 output_ch=3, skips=[4], activation="relu", device="cuda"):
        super().__init__(int(input_ch_periodic) * int(freq_nerf), int(input_ch_periodic_aux) * int(freq_nerf), D, W, skips,
                         activation, output_ch, device)


class DenseNPPNetTop1(_DenseBase):
    """models/networks.py:100-173 for any (D, W, skips, activation)."""

    def __init__(self, input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=8, W=256, freq_nerf=3, output_ch=3, skips=[4],
                 activation="relu", device="cuda"):
        super().__init__(int(input_ch_periodic) * int(freq_nerf), 0, D, W, skips, activation, output_ch, device)


class _LightFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_pos, x_periodic, net, *params):
        ctx.net = net
        ctx.bufs = net._run_forward(x_pos.detach().float().contiguous(), x_periodic.detach().float().contiguous(),
                                    [p.detach() for p in params])
        return ctx.bufs["out"].clone()

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.net._run_backward(ctx.bufs, gout.contiguous().float())
        ctx.bufs = None
        return (None, None, None) + tuple(grads)


class DenseNPPNetLight(nn.Module):
    """NPP_Net_light (models/networks.py:176-263), the network of the proposal-ranking fits (NPP_proposal/search.py:85-215,
    create_npp_net(..., is_search=True), models/helpers.py:92-105), for len(freq_scales) == 1 -- the only configuration the
    reference's search uses: forward(x (B, input_ch), x_periodic (B, 4 * n_offsets * n_angle_offsets)) -> raw (B, output_ch).
    Same constructor arguments, parameter names, shapes and construction order (default nn.Linear init from the global
    generator: scale_linears / feature_linear2 / alpha_linear are built and never used, as there); one dense-layer launch per
    layer (csrc/npp_linear.hip, exact fp32 MFMA)."""

    def __init__(self, input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=8, W=256, input_ch=3, output_ch=3, skips=[4],
                 activation="relu", device="cuda"):
        super().__init__()
        if len(freq_scales) != 1:
            raise NotImplementedError("NPP_Net_light with len(freq_scales) > 1 (the scale MLP) is not used by the reference's search")
        if activation not in _ACT:
            raise NotImplementedError(f"activation {activation!r}")
        self.act_f, self.act_b = _ACT[activation]
        self.D, self.W, self.skips, self.input_ch, self.output_ch = int(D), int(W), list(skips), int(input_ch), int(output_ch)
        self.input_ch_periodic = 2 * (2 * len(freq_offsets) * len(angle_offsets))                  # networks.py:188
        if int(input_ch_periodic) != self.input_ch_periodic:
            raise ValueError(f"input_ch_periodic {input_ch_periodic} != {self.input_ch_periodic} (len(freq_scales) == 1)")
        E, W = self.input_ch_periodic, self.W
        self.periodic_linears = nn.ModuleList([nn.Linear(E, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + E, W)
                                                                   for i in range(D - 1)])
        self.scale_linears = nn.ModuleList([nn.Linear(0 + W, W)])
        self.pos_linears = nn.ModuleList([nn.Linear(self.input_ch + W, W // 2)])
        self.feature_linear1 = nn.Linear(W, W)
        self.feature_linear2 = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, output_ch)
        self.to(torch.device(device))

    def _used(self):
        L = [(f"periodic_linears.{i}", m) for i, m in enumerate(self.periodic_linears)]
        return L + [("feature_linear1", self.feature_linear1), ("pos_linears.0", self.pos_linears[0]), ("rgb_linear", self.rgb_linear)]

    def forward(self, x, x_periodic):
        if x_periodic.shape[1] != self.input_ch_periodic or x.shape[1] != self.input_ch:
            raise ValueError(f"NPP_Net_light takes x (B,{self.input_ch}) and x_periodic (B,{self.input_ch_periodic})")
        params = [p for _, m in self._used() for p in (m.weight, m.bias)]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _LightFunction.apply(x, x_periodic, self, *params)
        with torch.no_grad():
            return self._run_forward(x.float().contiguous(), x_periodic.float().contiguous(), [p.detach() for p in params])["out"]

    def _run_forward(self, xpos, xp, params):
        B, W, E, D = xp.shape[0], self.W, self.input_ch_periodic, self.D
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=xp.device)          # noqa: E731
        P = {}
        for (name, _), i in zip(self._used(), range(0, len(params), 2)):
            P[name] = (params[i].contiguous(), params[i + 1].contiguous())
        b = dict(P=P, z=[], inp=[])
        h = xp
        for i in range(D):                                                            # networks.py:224-233
            z = f(B, W)
            if i in self.skips:
                cat = f(B, E + W)
                cat[:, :E].copy_(xp)
                out = cat[:, E:]
            else:
                cat, out = None, f(B, W)
            b["inp"].append(h)
            ops.linear_fwd(h, *P[f"periodic_linears.{i}"], self.act_f, out, z)
            b["z"].append(z)
            h = cat if cat is not None else out
        b["h_last"] = h
        pcat = f(B, W + self.input_ch)                                                # cat[feature1, input_pos] (:250)
        ops.linear_fwd(h, *P["feature_linear1"], 0, pcat[:, :W])
        pcat[:, W:].copy_(xpos)
        zp, ap, out = f(B, W // 2), f(B, W // 2), f(B, self.output_ch)
        ops.linear_fwd(pcat, *P["pos_linears.0"], self.act_f, ap, zp)
        ops.linear_fwd(ap, *P["rgb_linear"], 0, out)
        b.update(pcat=pcat, zp=zp, ap=ap, out=out)
        return b

    def _run_backward(self, b, gout):
        B, W, D = gout.shape[0], self.W, self.D
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=gout.device)        # noqa: E731
        P, G = b["P"], {}

        def wgrad(name, dz, x):
            w_, b_ = P[name]
            dw, db = torch.empty_like(w_), torch.empty_like(b_)
            ops.linear_bwd_weight(dz, x, dw, db)
            G[name] = (dw, db)
        wgrad("rgb_linear", gout, b["ap"])
        dap, dzp, df1 = f(B, W // 2), f(B, W // 2), f(B, W)
        ops.linear_bwd_data(gout, P["rgb_linear"][0], dap)
        ops.act_bwd(dap, b["zp"], self.act_b, dzp)
        wgrad("pos_linears.0", dzp, b["pcat"])
        ops.linear_bwd_data(dzp, P["pos_linears.0"][0], df1, in_used=W)               # no gradient to input_pos
        wgrad("feature_linear1", df1, b["h_last"])
        hl_w = b["h_last"].shape[1]
        dh_full = f(B, hl_w)
        ops.linear_bwd_data(df1, P["feature_linear1"][0], dh_full)
        dh = dh_full[:, hl_w - W:]
        for i in range(D - 1, -1, -1):
            name = f"periodic_linears.{i}"
            dz = f(B, W)
            ops.act_bwd(dh, b["z"][i], self.act_b, dz)
            wgrad(name, dz, b["inp"][i])
            if i > 0:
                w_ = P[name][0]
                dh = f(B, W)
                ops.linear_bwd_data(dz, w_[:, w_.shape[1] - W:], dh)
        out = []
        for name, _ in self._used():
            out += list(G[name])
        return out
