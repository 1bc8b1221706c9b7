"""The reference's own call signatures for the hot path, served by the HIP kernels.

A maintainer of the reference swaps imports and keeps NPP_completion/train.py's loop unchanged:

    from models.embedder import get_embedder          ->  from npp_amd.reference_api import get_embedder
    from models.helpers import create_npp_net, render ->  from npp_amd.reference_api import create_npp_net, render
    from models.mse_calculator import img2mse         ->  from npp_amd.reference_api import img2mse

Everything computes in libnpp_hip.so; tensors are ordinary torch tensors on the GPU and autograd works
across the boundary (torch.autograd.Function wrappers), so `loss.backward(); optimizer.step()` with the
torch.optim.Adam that create_npp_net returns behaves as in the reference.  This form materialises the
(N, K*462) embedding tables exactly like the reference does, i.e. it pays the 5.5 KB/pixel of HBM traffic
the fused path (npp_amd.fit.CompletionFit, coordinates in -> pixels out) was built to remove: use it for
drop-in checks and incremental migration, the fused path for speed.

Mirrors (reference file:line):
  Embedder / Embedder_periodic / get_embedder   models/embedder.py:6-148
  NPP_Net / NPP_Net_top1                        models/networks.py:8-173
  batchify / run_network / render / create_npp_net   models/helpers.py:14-175
  img2mse                                       models/mse_calculator.py:13-27
  AdaptiveLossFunction (the adaptive_pix object)     robust_loss_pytorch/adaptive.py:90-204
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import EmbedCfg, param_layout, NPP_E, NPP_WIDTH, NPP_N_FREQ, FUSED_WIDTHS  # noqa: F401

_DEFAULT_OFFSETS = (0.0, -1.0, 1.0, 0.5, -0.5)
_OUT_ACT = {None: 0, 0: 0, 1: 1, 2: 2}


def _as_f32(x):
    return x if x.dtype == torch.float32 else x.float()


# ---------------------------------------------------------------------------------------------
# embedders (models/embedder.py)
class Embedder:
    """Fourier features of an (N, d) tensor: cat[x, sin(f0 x), cos(f0 x), ...] (embedder.py:6-56).
    'gaussian' sampling draws the frequencies from torch's GLOBAL generator at construction, like the
    reference (embedder.py:26), so a seeded script sees the same values."""

    def __init__(self, res, **kwargs):
        self.kwargs = kwargs
        self.res = res
        self.is_search = kwargs.get("is_search", False)         # 2-D position embedder of the proposal-ranking fits (:52-54)
        n = kwargs["num_freqs"]
        if n > NPP_N_FREQ:
            raise NotImplementedError(f"num_freqs {n} > {NPP_N_FREQ} (this build's table size)")
        samp = kwargs["sampling"]
        if samp == "gaussian":
            fb = torch.normal(mean=0.0, std=1.0, size=(n, 1)) * 10
        elif samp == "log":
            fb = 2.0 ** torch.linspace(0.0, kwargs["max_freq_log2"], steps=n)
        elif samp == "pure_gaussian":
            raise NotImplementedError("'pure_gaussian' leaves freq_bands undefined in the reference (embedder.py:30-38)")
        else:
            fb = torch.linspace(2.0 ** 0.0, 2.0 ** kwargs["max_freq_log2"], steps=n)
        self.freq_bands = fb.reshape(-1).to(torch.float32)
        self.include_input = bool(kwargs["include_input"])
        d = kwargs["input_dims"]
        self.out_dim = d * (2 * n + int(self.include_input))

    def embed(self, inputs):
        if self.is_search:
            # embedder.py:52-54 normalises the (row, col) pair to [-1, 1] IN PLACE (its callers pass a .clone(),
            # NPP_proposal/search.py:104-105); a floating-point argument is written back the same way
            x = torch.stack([(_as_f32(inputs[:, 0]) / self.res[0] - 0.5) * 2, (_as_f32(inputs[:, 1]) / self.res[1] - 0.5) * 2], 1)
            if inputs.is_floating_point():
                inputs.copy_(x)
            x = x.contiguous()
        else:
            x = _as_f32(inputs).contiguous()
        return ops.fourier_fwd(x, self.freq_bands.tolist(), self.include_input)


class Embedder_periodic:
    """Periodicity-aware input warping (embedder.py:93-148): (N,2) [row y, col x] -> (N,22)."""

    def __init__(self, res, selected_angles, selected_periods, freq_scales, freq_offsets, angle_offsets, **kwargs):
        self.kwargs = kwargs
        self.freq_scales, self.freq_offsets, self.angle_offsets = freq_scales, freq_offsets, angle_offsets
        self.include_input = bool(kwargs.get("include_input", True))      # False for the is_search form (:76-86): 20 columns
        if list(freq_scales) != [1] and list(freq_scales) != [1.0]:
            raise NotImplementedError("kernels are specialised for freq_scales=[1] (the reference's completion config)")
        if len(freq_offsets) != len(_DEFAULT_OFFSETS) or list(angle_offsets) not in ([0], [0.0]):
            raise NotImplementedError("kernels are specialised for 5 freq_offsets and angle_offsets=[0]")
        ang = [float(a) for a in selected_angles]
        per = [float(p) for p in selected_periods]
        # the Fourier frequencies are not used by the warp stage; any finite values do
        self.cfg = EmbedCfg.make([ang], [per], [1.0] * NPP_N_FREQ, res, tuple(float(o) for o in freq_offsets))
        self.out_dim = (2 if self.include_input else 0) + 2 * 2 * len(freq_offsets)

    def embed(self, inputs):
        c = inputs
        if c.dtype != torch.int32:
            if c.is_floating_point() and not bool((c == c.round()).all()):
                raise NotImplementedError("non-integer coordinates: the kernels take pixel indices (train.py:89-105 passes them)")
            c = c.to(torch.int32)
        v = ops.warp_fwd(c.contiguous(), self.cfg)                         # (N, 22): [x_n, 10 x orientation 0, y_n, 10 x orientation 1]
        if self.include_input:
            return v
        return torch.cat([v[:, 1:11], v[:, 12:22]], 1).contiguous()


def get_embedder(multires, i=0, res=None, selected_angles=None, selected_periods=None,
                 freq_scales=None, freq_offsets=None, angle_offsets=None, is_search=False):
    """models/embedder.py:60-90."""
    if i == -1:
        return nn.Identity(), 3
    embed_kwargs = {
        "include_input": True, "input_dims": 1, "max_freq_log2": multires - 1, "num_freqs": multires,
        "sampling": "gaussian", "periodic_fns": [torch.sin, torch.cos], "is_search": is_search,
    }
    if selected_periods is None and selected_angles is None:
        embed_kwargs["input_dims"] = 2 if is_search else 1
        embedder = Embedder(res, **embed_kwargs)
    else:
        if is_search:
            embed_kwargs["include_input"] = False
        embedder = Embedder_periodic(res, selected_angles, selected_periods, freq_scales, freq_offsets, angle_offsets,
                                     **embed_kwargs)
    return embedder, embedder.out_dim


# ---------------------------------------------------------------------------------------------
# network modules (models/networks.py)
class _NetFunction(torch.autograd.Function):
    """forward = fused MLP on a materialised embedding; backward = dgrad chain + grouped wgrad, the weight
    gradient lands in the blob-shaped grad buffer whose views are the parameters' .grad."""

    @staticmethod
    def forward(ctx, emb, blob, net, out_act, train):
        n = emb.shape[0]
        bp = ops.pad_rows(n)
        e = _as_f32(emb)
        if bp != n:
            e = torch.cat([e, e.new_zeros((bp - n, e.shape[1]))], 0)
        e = e.contiguous()
        net._sync_pack()
        # `train` is decided by the caller: inside forward() grad mode is always off, and needs_input_grad
        # ignores torch.no_grad().  The stash belongs to THIS call until its backward has run.
        ws = net._take_workspace(bp) if train else None
        out = ops.mlp_fwd_emb(e, net.K, net._wf, blob.detach(), None, ws["actT"] if train else None, out_act, net.W)
        ctx.net, ctx.n, ctx.bp, ctx.out_act, ctx.ws = net, n, bp, out_act, ws
        ctx.save_for_backward(out)
        return out[:n]

    @staticmethod
    def backward(ctx, gout):
        net, n, bp = ctx.net, ctx.n, ctx.bp
        (out,) = ctx.saved_tensors
        ws = ctx.ws
        if ws is None:
            raise RuntimeError("backward through a forward that ran without gradient tracking")
        g = gout.contiguous().float()
        if bp != n:
            g = torch.cat([g, g.new_zeros((bp - n, 3))], 0)
        ops.mlp_bwd_act(g, out, net.K, net._wb, net._blob.detach(), ws["actT"], ws["dzT"], ctx.out_act, net.W)
        ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, net.K, net._ksplit, ws["gslabs"], net.W)
        gb = torch.empty_like(net._blob)
        ops.grad_reduce(ws["gslabs"], net._ksplit, net._n_params, gb)
        net._give_workspace(bp, ws)
        ctx.ws = None
        return None, gb, None, None, None    # no gradient to the embedding (train.py never asks for one)


class _NetBase(nn.Module):
    """The trainable state is ONE flat nn.Parameter in the layout the kernels read (the reference's tensors
    back to back); state_dict()/load_state_dict() speak the reference's names and shapes
    (networks.py:40-49).  torch.optim.Adam over the blob is elementwise identical to Adam over the
    reference's per-layer tensors."""

    def __init__(self, K, D, W, skips, activation, device, tail_order):
        super().__init__()
        if D != 8 or W not in FUSED_WIDTHS or list(skips) != [4]:
            raise NotImplementedError(f"kernels are specialised for D=8, W in {FUSED_WIDTHS}, skips=[4] (got D={D}, W={W}, skips={skips})")
        if activation != "snake":
            raise NotImplementedError("kernels implement the 'snake' activation (the reference's setting, configs/*.txt)")
        self.K, self.D, self.W, self.skips = K, D, W, list(skips)
        dev = torch.device(device)
        self._layout, self._n_params = param_layout(K, W)
        index = {name: (off, r, c) for name, off, r, c in self._layout}
        blob = torch.zeros(self._n_params, dtype=torch.float32)
        self._unused = {}       # tensors the reference constructs but never uses (alpha_linear; top1: feature_linear2)
        # default nn.Linear initialisation drawn from the global generator in the reference's construction order
        order = [("periodic_linears.0", W, NPP_E)]
        order += [(f"periodic_linears.{i + 1}", W, W + NPP_E if i in self.skips else W) for i in range(D - 1)]
        for lin_name, rows, cols in order + tail_order:
            lin = nn.Linear(cols, rows)
            for suffix, t in (("weight", lin.weight), ("bias", lin.bias)):
                key = f"{lin_name}.{suffix}"
                if key in index:
                    off, r, c = index[key]
                    blob[off:off + r * c] = t.detach().reshape(-1)
                else:
                    self._unused[key] = t.detach().clone()
        self._blob = nn.Parameter(blob.to(dev))
        self._wf = torch.empty(ops.pack_bytes(K, 0, W), dtype=torch.uint8, device=dev)
        self._wb = torch.empty(ops.pack_bytes(K, 1, W), dtype=torch.uint8, device=dev)
        self._packed_version = -1
        self._ksplit = 4
        self._ws = {}

    # ---- reference-named views -----------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        out = {}
        for name, off, r, c in self._layout:
            v = self._blob.detach()[off:off + r * c]
            out[name] = (v.view(r, c) if name.endswith("weight") else v).clone()
        out.update({k: v.clone() for k, v in self._unused.items()})
        return out

    def load_state_dict(self, sd, strict=True):
        missing = [n for n, *_ in self._layout if n not in sd]
        if missing and strict:
            raise KeyError(f"missing keys: {missing}")
        with torch.no_grad():
            for name, off, r, c in self._layout:
                if name in sd:
                    t = torch.as_tensor(sd[name], dtype=torch.float32).reshape(-1)
                    if t.numel() != r * c:
                        raise ValueError(f"{name}: expected {r}x{c}, got {tuple(torch.as_tensor(sd[name]).shape)}")
                    self._blob[off:off + r * c] = t.to(self._blob.device)
            for k in self._unused:
                if k in sd:
                    self._unused[k] = torch.as_tensor(sd[k], dtype=torch.float32).clone()
        self._packed_version = -1

    # ---- kernel-side state ---------------------------------------------------------------------
    def _sync_pack(self):
        v = self._blob._version
        if v != self._packed_version:
            ops.pack_weights(self._blob.detach(), self.K, self._wf, self._wb, self.W)
            self._packed_version = v

    def _take_workspace(self, bp):
        """Stash + gradient workspace for one forward/backward pair (several forwards may be pending)."""
        free = self._ws.get(bp)
        if free:
            return free.pop()
        s = ops.train_workspace(self.K, bp, self._ksplit, self.W)
        dev = self._blob.device
        return {"actT": torch.empty(s[1], dtype=torch.uint8, device=dev),
                "dzT": torch.empty(s[2], dtype=torch.uint8, device=dev),
                "gslabs": torch.empty(s[3] // 4, dtype=torch.float32, device=dev)}

    def _give_workspace(self, bp, ws):
        if bp not in self._ws:
            self._ws = {bp: []}              # keep the pool of one batch shape only
        if len(self._ws[bp]) < 2:
            self._ws[bp].append(ws)

    out_act = 0     # the module returns the raw network output; render() applies sigmoid / tanh

    def forward(self, x, x_periodic):
        """(None, (B, K*462)) -> (B, 3) (networks.py:56-95 / :134-173)."""
        if x_periodic.shape[1] != self.K * NPP_E:
            raise ValueError(f"x_periodic has {x_periodic.shape[1]} columns, this net takes {self.K * NPP_E}")
        train = torch.is_grad_enabled() and self._blob.requires_grad
        return _NetFunction.apply(x_periodic, self._blob, self, self.out_act, train)


def _fused_config(D, W, skips, activation, *widths):
    """The fused chain kernels serve D = 8, W = 256 or 512 (one library each), skips = [4], snake and 462-wide proposals;
    everything else goes to the generic dense-layer path (dense.py)."""
    return D == 8 and W in FUSED_WIDTHS and list(skips) == [4] and activation == "snake" and all(w % NPP_E == 0 and w > 0 for w in widths)


class NPP_Net(_NetBase):
    """models/networks.py:8-95 (K > 1: top-1 proposal + auxiliary proposals).  Configurations outside the fused kernels'
    specialisation (e.g. netwidth 128, activation='relu', other multires) are constructed as
    dense.DenseNPPNet: same arguments, parameter names and forward, one launch per layer."""

    def __new__(cls, input_ch_periodic, input_ch_periodic_aux, freq_scales, freq_offsets, angle_offsets, D=8, W=256, freq_nerf=3,
                output_ch=3, skips=[4], activation="relu", device="cuda"):
        E1, Ea = int(input_ch_periodic) * int(freq_nerf), int(input_ch_periodic_aux) * int(freq_nerf)
        if cls is NPP_Net and not (_fused_config(D, W, skips, activation, Ea) and E1 == NPP_E and output_ch == 3):
            from .dense import DenseNPPNet
            return DenseNPPNet(input_ch_periodic, input_ch_periodic_aux, freq_scales, freq_offsets, angle_offsets, D=D, W=W,
                               freq_nerf=freq_nerf, output_ch=output_ch, skips=skips, activation=activation, device=device)
        return super().__new__(cls)

    def __init__(self, input_ch_periodic, input_ch_periodic_aux, freq_scales, freq_offsets, angle_offsets, D=8, W=256,
                 freq_nerf=3, output_ch=3, skips=[4], activation="relu", device="cuda"):
        E1, Ea = int(input_ch_periodic) * int(freq_nerf), int(input_ch_periodic_aux) * int(freq_nerf)
        if E1 != NPP_E or Ea % NPP_E or Ea == 0 or output_ch != 3:
            raise NotImplementedError(f"kernels take 462-wide proposals (got {E1} + {Ea}) and 3 outputs")
        tail = [("scale_linears.0", W, Ea + W), ("pos_linears.0", W // 2, 2 * W), ("feature_linear1", W, W),
                ("feature_linear2", W, W), ("alpha_linear", 1, W), ("rgb_linear", 3, W // 2)]
        super().__init__(1 + Ea // NPP_E, D, W, skips, activation, device, tail)
        self.input_ch_periodic, self.input_ch_periodic_aux = E1, Ea


class NPP_Net_top1(_NetBase):
    """models/networks.py:100-173 (K == 1); other configurations -> dense.DenseNPPNetTop1 (see NPP_Net)."""

    def __new__(cls, input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=8, W=256, freq_nerf=3, output_ch=3, skips=[4],
                activation="relu", device="cuda"):
        E1 = int(input_ch_periodic) * int(freq_nerf)
        if cls is NPP_Net_top1 and not (_fused_config(D, W, skips, activation) and E1 == NPP_E and output_ch == 3):
            from .dense import DenseNPPNetTop1
            return DenseNPPNetTop1(input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=D, W=W, freq_nerf=freq_nerf,
                                   output_ch=output_ch, skips=skips, activation=activation, device=device)
        return super().__new__(cls)

    def __init__(self, input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=8, W=256, freq_nerf=3, output_ch=3,
                 skips=[4], activation="relu", device="cuda"):
        E1 = int(input_ch_periodic) * int(freq_nerf)
        if E1 != NPP_E or output_ch != 3:
            raise NotImplementedError(f"kernels take a 462-wide proposal (got {E1}) and 3 outputs")
        tail = [("pos_linears.0", W // 2, W), ("feature_linear1", W, W), ("feature_linear2", W, W),
                ("alpha_linear", 1, W), ("rgb_linear", 3, W // 2)]
        super().__init__(1, D, W, skips, activation, device, tail)
        self.input_ch_periodic = E1


def NPP_Net_light(input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=8, W=256, input_ch=3, output_ch=3, skips=[4],
                  activation="relu", device="cuda"):
    """models/networks.py:176-263 -- the network of the proposal-ranking fits; one dense-layer launch per layer
    (dense.DenseNPPNetLight).  The fits themselves have a dedicated host loop in light.ProposalRanker."""
    from .dense import DenseNPPNetLight
    return DenseNPPNetLight(input_ch_periodic, freq_scales, freq_offsets, angle_offsets, D=D, W=W, input_ch=input_ch,
                            output_ch=output_ch, skips=skips, activation=activation, device=device)


# ---------------------------------------------------------------------------------------------
# render plumbing (models/helpers.py:14-62)
def batchify(fn, chunk):
    if chunk is None:
        return fn

    def ret(inputs, inputs_periodic):
        return torch.cat([fn(None if inputs is None else inputs[i:i + chunk], inputs_periodic[i:i + chunk])
                          for i in range(0, inputs_periodic.shape[0], chunk)], 0)
    return ret


def run_network(inputs, inputs_periodic, fn, netchunk=1024 * 64):
    outputs_flat = batchify(fn, netchunk)(inputs, inputs_periodic)
    return torch.reshape(outputs_flat, list(inputs_periodic.shape[:-1]) + [outputs_flat.shape[-1]])


def render(select_coords_emb, select_coords_emb_periodic, args, network_query_fn, network_fn):
    raw = network_query_fn(select_coords_emb, select_coords_emb_periodic, network_fn)
    if args.normalize_type == 1:
        return torch.sigmoid(raw)
    if args.normalize_type == 2:
        return torch.tanh(raw)
    assert False, "Wrong normalize type"


# ---------------------------------------------------------------------------------------------
# adaptive robust pixel loss (robust_loss_pytorch/adaptive.py; models/mse_calculator.py)
class _PixelLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mask, latent_alpha, latent_scale, owner):
        n = x.shape[0]
        lat = torch.cat([latent_alpha.reshape(-1), latent_scale.reshape(-1)]).detach().contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dlat = torch.zeros(6, dtype=torch.float32, device=x.device)
        m = None if mask is None else _as_f32(mask).reshape(n).contiguous()
        ops.pixel_loss(_as_f32(x).contiguous(), _as_f32(y).contiguous(), m, lat, owner._spline, owner._n_knots,
                       owner._x_scale, 1.0, loss, dx, dlat)
        ctx.save_for_backward(dx, dlat)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dx, dlat = ctx.saved_tensors
        return dx * g, -dx * g, None, (dlat[:3] * g).reshape(1, 3), (dlat[3:] * g).reshape(1, 3), None


class AdaptiveLossFunction(nn.Module):
    """The reference's `adaptive_pix` (helpers.py:8-9: AdaptiveLossFunction(num_dims=3, float_dtype=float32,
    device=0)): learnable latent_alpha / latent_scale of shape (1,3) (adaptive.py:146-181)."""

    def __init__(self, num_dims=3, float_dtype=np.float32, device="cuda", alpha_lo=0.001, alpha_hi=1.999,
                 scale_lo=1e-5, scale_init=1.0):
        super().__init__()
        if num_dims != 3 or (alpha_lo, alpha_hi, scale_lo, scale_init) != (0.001, 1.999, 1e-5, 1.0):
            raise NotImplementedError("kernels implement the reference's pixel-loss configuration (3 channels, default ranges)")
        dev = torch.device("cuda" if isinstance(device, int) else device)
        from .model import LATENT_ALPHA_INIT
        self.latent_alpha = nn.Parameter(torch.full((1, 3), LATENT_ALPHA_INIT, dtype=torch.float32, device=dev))
        self.latent_scale = nn.Parameter(torch.zeros((1, 3), dtype=torch.float32, device=dev))
        self._spline, self._n_knots, self._x_scale = ops.load_spline(dev)

    def alpha(self):
        return torch.sigmoid(self.latent_alpha) * (1.999 - 0.001) + 0.001

    def scale(self):
        return (1 - 1e-5) * torch.nn.functional.softplus(self.latent_scale + 0.54132485) + 1e-5


class _QuadLossFunction(torch.autograd.Function):
    """img2mse with loss_type 'l2' / 'robust_loss' (mse_calculator.py:19-23): coef * mean(x^2) on npp_pixel_loss_quad."""

    @staticmethod
    def forward(ctx, x, y, mask, coef):
        x2 = x.detach().reshape(-1, 3).contiguous()
        y2 = y.detach().reshape(-1, 3).contiguous()
        m = None if mask is None else mask.detach().reshape(-1).contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=x2.device)
        dpred = torch.empty_like(x2)
        ops.pixel_loss_quad(x2, y2, m, coef, 1.0, loss, dpred)
        ctx.save_for_backward(dpred)
        ctx.shape = x.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (g * dpred).reshape(ctx.shape), (-g * dpred).reshape(ctx.shape), None, None


def img2mse(x, y, loss_type, adaptive, mask=None):
    """models/mse_calculator.py:13-27: 'robust_loss_adaptive' (the reference's setting, train.py:195), 'l2', 'robust_loss'."""
    if loss_type != "robust_loss_adaptive":
        if loss_type not in ops.QUAD_COEF:
            raise NotImplementedError(f"loss_type {loss_type!r}: 'robust_loss_adaptive', 'l2' or 'robust_loss' (models/mse_calculator.py:19-23)")
        return _QuadLossFunction.apply(x, y, mask, ops.quad_coef(loss_type))
    return _PixelLossFunction.apply(x, y, mask, adaptive.latent_alpha, adaptive.latent_scale, adaptive)


# ---------------------------------------------------------------------------------------------
# create_npp_net (models/helpers.py:75-175)
_adaptive_pix = None


def adaptive_pix(device="cuda"):
    """The module-level adaptive pixel loss of models/helpers.py:8-9 (created on first use, not at import)."""
    global _adaptive_pix
    if _adaptive_pix is None:
        _adaptive_pix = AdaptiveLossFunction(num_dims=3, float_dtype=np.float32, device=device)
    return _adaptive_pix


def create_npp_net(args, selected_angles, selected_periods, res, percep_net, is_search=False, style_net=None):
    embedder, freq_nerf = get_embedder(args.multires, args.i_embed, res, is_search=is_search)
    if is_search:
        # helpers.py:92-105: ONE candidate (angles, periods of its two orientations), no top-K in the model; the periodic
        # embedder is returned as a single object, not a list
        embedder_periodics, input_ch_periodic = get_embedder(args.multires, args.i_embed, res, selected_angles=selected_angles,
                                                             selected_periods=selected_periods, freq_scales=args.freq_scales,
                                                             freq_offsets=args.freq_offsets, angle_offsets=args.angle_offsets,
                                                             is_search=True)
        model = NPP_Net_light(D=args.netdepth, W=args.netwidth, input_ch=freq_nerf, input_ch_periodic=input_ch_periodic,
                              freq_scales=args.freq_scales, freq_offsets=args.freq_offsets, angle_offsets=args.angle_offsets,
                              output_ch=3, skips=[4], activation=args.activation)
    else:
        embedder_periodics, input_ch_periodics = [], []
        for i in range(args.p_topk):
            ep, ch = get_embedder(args.multires, args.i_embed, res, selected_angles=selected_angles[i],
                                  selected_periods=selected_periods[i], freq_scales=args.freq_scales,
                                  freq_offsets=args.freq_offsets, angle_offsets=args.angle_offsets)
            embedder_periodics.append(ep)
            input_ch_periodics.append(ch)
        input_ch_periodics = np.array(input_ch_periodics)
        common = dict(freq_scales=args.freq_scales, freq_offsets=args.freq_offsets, angle_offsets=args.angle_offsets,
                      D=args.netdepth, W=args.netwidth, freq_nerf=freq_nerf, output_ch=3, skips=[4], activation=args.activation)
        if args.p_topk > 1:
            model = NPP_Net(input_ch_periodic=input_ch_periodics[:1].sum(), input_ch_periodic_aux=input_ch_periodics[1:].sum(),
                            **common)
        else:
            model = NPP_Net_top1(input_ch_periodic=input_ch_periodics[:1].sum(), **common)
    grad_vars = list(model.parameters()) + list(adaptive_pix().parameters())
    if percep_net is not None and getattr(args, "use_adaptive_perceptual_loss", False):
        for adaptive in percep_net.adaptive_perceps:
            grad_vars += list(adaptive.parameters())
    if style_net is not None and getattr(args, "use_adaptive_style_loss", False):
        for adaptive in style_net.adaptives:
            grad_vars += list(adaptive.parameters())
    network_query_fn = lambda inputs, inputs_periodic, network_fn: run_network(inputs, inputs_periodic, network_fn,  # noqa: E731
                                                                                 netchunk=args.netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    render_kwargs_train = {"network_query_fn": network_query_fn, "network_fn": model}
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, embedder, embedder_periodics
