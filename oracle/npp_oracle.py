"""NumPy fp32 restatement of the reference's per-image NPP-Net optimisation path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line (relative to /root/reference) whose behaviour it restates.
Nothing here is copied: the reference is PyTorch/autograd, this is explicit
NumPy with hand-derived backward passes, which is also the derivation the HIP
kernels implement.

All arithmetic is float32 unless stated.  ``emulate_bf16=True`` options round
MFMA operands to bfloat16 at the same points the HIP kernels do, for tight
kernel-vs-oracle comparisons; the default (False) is the reference's fp32 maths.
"""
from __future__ import annotations

import math
import os

import numpy as np

F32 = np.float32

__all__ = [
    "F32", "FREQ_OFFSETS", "SEED0_FREQS", "E_PER_PROPOSAL", "bf16_round",
    "periodic_warp", "fourier_features", "embed", "snake", "snake_grad",
    "param_shapes", "init_params", "mlp_forward", "mlp_backward", "render",
    "sigmoid", "load_partition_spline", "adaptive_params", "robust_nll",
    "robust_nll_grads", "img2mse", "img2mse_grads", "img2mse_quad_grads", "adam_init", "adam_step",
    "lr_schedule", "psnr", "synthetic_image", "synthetic_periodicity",
    "patch_size_from_period", "mlp_macs_per_pixel",
]

# options/arg_config.py:20 -- default fine-level period offsets, in this order
FREQ_OFFSETS = (0.0, -1.0, 1.0, 0.5, -0.5)
# models/embedder.py:26 under torch.manual_seed(0) on CPU
SEED0_FREQS = (15.409960746765137, -2.93428897857666, -21.787893295288086,
               5.68431282043457, -10.845223426818848, -13.985954284667969,
               4.033468246459961, 8.380263328552246, -7.192575931549072,
               -4.033435344696045)
E_PER_PROPOSAL = 462  # 22 warped coords x (1 + 2*10 Fourier blocks)


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (what v_cvt_pk_bf16_f32 does)."""
    a = np.ascontiguousarray(x, dtype=F32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(F32).reshape(a.shape)


def minifloat_round(x, mbits, emin, maxv):
    """Round-to-nearest-even fp32 -> an OCP 8-bit float -> fp32, saturating, subnormals kept: what v_cvt_pk_fp8_f32 (e4m3:
    mbits 3, emin -6, max 448) / v_cvt_pk_bf8_f32 (e5m2: mbits 2, emin -14, max 57344) do with MODE.FP16_OVFL set
    (tools/micro/fp8_probe.hip).  Used only to EMULATE the build's 8-bit training stash (npp_tune "stash8"): the reference
    has no such rounding."""
    a = np.abs(np.asarray(x, dtype=np.float64))
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, emin)
    q = np.exp2(e - mbits)
    r = np.minimum(np.rint(a / q) * q, maxv)
    return (np.sign(x) * r).astype(F32)


def fp8_round(x):
    return minifloat_round(x, 3, -6, 448.0)


def bf8_round(x):
    return minifloat_round(x, 2, -14, 57344.0)


DZ8_LIFT = 11      # csrc/npp_layout.h kDz8Lift


def dz8_quantise(dz, draw, tile=64):
    """The build's bf8 gradient stash: per 64-row workgroup tile the chain runs on dL/draw * 2^(DZ8_LIFT - e), e =
    floor(log2 max |dL/draw| of the tile) clipped to [-100, 100] (csrc/npp_mlp_bwd.hip); the scaled values are rounded to
    e5m2 and the power of two is undone by the matrix instruction's block scale (exact)."""
    out = np.empty_like(np.asarray(dz, dtype=F32))
    for r0 in range(0, dz.shape[0], tile):
        m = float(np.abs(draw[r0:r0 + tile]).max()) if draw[r0:r0 + tile].size else 0.0
        e = int(np.floor(np.log2(m))) if m >= 2.0 ** -126 else -127
        e = min(max(e, -100), 100)
        sc = np.float64(2.0) ** (DZ8_LIFT - e)
        out[r0:r0 + tile] = (bf8_round((dz[r0:r0 + tile].astype(np.float64) * sc).astype(F32)).astype(np.float64) / sc).astype(F32)
    return out


# --------------------------------------------------------------------------
# a1: periodicity-aware warp  (models/embedder.py:102-148)
# --------------------------------------------------------------------------
def periodic_warp(coords_yx, angles_deg, periods, res, freq_offsets=FREQ_OFFSETS,
                  freq_scales=(1.0,), angle_offsets=(0.0,)):
    """(N,2) (row=y, col=x) -> (N,22) fp32.

    Column order (embedder.py:110-133,140-148): [x/W*2-1, {sin,cos}(phi_0(o)) for o
    in offsets] ++ [y/H*2-1, {sin,cos}(phi_1(o))...]; phi_i(o) = ((y cos th_i +
    x sin th_i) mod p) / p * 2 * pi, p = (period_i + o) * scale, th in degrees
    (deg2rad, :121-127); mod = torch.remainder (result has the divisor's sign).
    """
    c = np.asarray(coords_yx, dtype=F32)
    y = c[:, 0:1]
    x = c[:, 1:2]
    H, W = res
    cols_x = [(x / F32(W) - F32(0.5)) * F32(2)]
    cols_y = [(y / F32(H) - F32(0.5)) * F32(2)]
    ang = np.asarray(angles_deg, dtype=F32)
    per = np.asarray(periods, dtype=F32)
    for sc in freq_scales:
        for off in freq_offsets:
            for idx in range(2):
                for aoff in angle_offsets:
                    freq = F32(F32(per[idx] + F32(off)) * F32(sc))
                    th = F32(np.deg2rad(F32(ang[idx] + F32(aoff))))
                    t = y * F32(np.cos(th)) + x * F32(np.sin(th))
                    r = np.remainder(t, freq).astype(F32)
                    phi = ((r / freq) * F32(2)) * F32(np.pi)
                    tgt = cols_x if idx == 0 else cols_y
                    tgt.append(np.sin(phi).astype(F32))
                    tgt.append(np.cos(phi).astype(F32))
    return np.concatenate(cols_x + cols_y, axis=1).astype(F32)


# --------------------------------------------------------------------------
# a2: Gaussian Fourier features  (models/embedder.py:11-56)
# --------------------------------------------------------------------------
def fourier_features(v, freqs):
    """(N,D) -> (N, D*(1+2*len(freqs))): [v, sin(f0 v), cos(f0 v), sin(f1 v) ...]
    (embedder.py:15-17 include_input, :41-44 per-frequency sin/cos, :56 cat)."""
    v = np.asarray(v, dtype=F32)
    out = [v]
    for f in np.asarray(freqs, dtype=F32).reshape(-1):
        a = v * F32(f)
        out.append(np.sin(a).astype(F32))
        out.append(np.cos(a).astype(F32))
    return np.concatenate(out, axis=1).astype(F32)


def embed(coords_yx, angles_deg, periods, freqs, res, freq_offsets=FREQ_OFFSETS):
    """K proposals, proposal-major concat (NPP_completion/train.py:93-105).
    angles_deg, periods: (K,2).  Returns (N, K*22*(1+2*n_freq))."""
    angles_deg = np.asarray(angles_deg, dtype=F32).reshape(-1, 2)
    periods = np.asarray(periods, dtype=F32).reshape(-1, 2)
    parts = []
    for k in range(angles_deg.shape[0]):
        v = periodic_warp(coords_yx, angles_deg[k], periods[k], res, freq_offsets)
        parts.append(fourier_features(v, freqs))
    return np.concatenate(parts, axis=1)


# --------------------------------------------------------------------------
# a6: snake activation  (models/activations.py:29-35), a = 1
# --------------------------------------------------------------------------
def snake(z):
    s = np.sin(z)
    return (z + s * s).astype(F32)


def snake_grad(z):
    return (F32(1) + np.sin(F32(2) * z)).astype(F32)


def sigmoid(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


# --------------------------------------------------------------------------
# a5: NPP_Net / NPP_Net_top1  (models/networks.py:9-95, :100-173)
# --------------------------------------------------------------------------
def param_shapes(K, W=256, E=E_PER_PROPOSAL, D=8, skips=(4,)):
    """Parameter tensors that take part in forward, torch state_dict names and
    (out,in) shapes (networks.py:40-49, :128-140).  alpha_linear (and, for K==1,
    feature_linear2) exist in the reference but never receive gradients
    (SURVEY.md A.15); they are not part of the path and are omitted."""
    sh = {}
    for i in range(D):
        if i == 0:
            cin = E
        elif (i - 1) in skips:
            cin = W + E
        else:
            cin = W
        sh[f"periodic_linears.{i}.weight"] = (W, cin)
        sh[f"periodic_linears.{i}.bias"] = (W,)
    sh["feature_linear1.weight"] = (W, W)
    sh["feature_linear1.bias"] = (W,)
    if K > 1:
        sh["scale_linears.0.weight"] = (W, W + (K - 1) * E)
        sh["scale_linears.0.bias"] = (W,)
        sh["feature_linear2.weight"] = (W, W)
        sh["feature_linear2.bias"] = (W,)
        sh["pos_linears.0.weight"] = (W // 2, 2 * W)
    else:
        sh["pos_linears.0.weight"] = (W // 2, W)
    sh["pos_linears.0.bias"] = (W // 2,)
    sh["rgb_linear.weight"] = (3, W // 2)
    sh["rgb_linear.bias"] = (3,)
    return sh


def init_params(K, W=256, E=E_PER_PROPOSAL, D=8, seed=0):
    """nn.Linear default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in))
    for weight and bias).  Values are our own NumPy stream: initial weights are an
    explicit input of the path."""
    rng = np.random.RandomState(seed)
    P = {}
    for name, shp in param_shapes(K, W, E, D).items():
        if name.endswith("weight"):
            bound = 1.0 / math.sqrt(shp[1])
            last_bound = bound
        else:
            bound = last_bound
        P[name] = rng.uniform(-bound, bound, size=shp).astype(F32)
    return P


def _lin(x, P, name, rb):
    w = P[name + ".weight"]
    if rb:
        x = bf16_round(x)
        w = bf16_round(w)
    return (x @ w.T + P[name + ".bias"]).astype(F32)


def mlp_forward(P, emb, K, E=E_PER_PROPOSAL, D=8, skips=(4,), emulate_bf16=False):
    """(B, K*E) -> raw (B,3) and the cache the backward needs.
    Topology: networks.py:56-95 (K>1) / :145-173 (K==1); concat orders
    [input, h] (:71), [f1, aux] (:76), [f1, f2] (:85)."""
    rb = emulate_bf16
    emb = np.asarray(emb, dtype=F32)
    x0 = emb[:, :E]
    aux = emb[:, E:]
    assert aux.shape[1] == (K - 1) * E
    cache = {"x0": x0, "aux": aux, "K": K}
    h = x0
    for i in range(D):
        cache[f"in{i}"] = h
        z = _lin(h, P, f"periodic_linears.{i}", rb)
        cache[f"z{i}"] = z
        h = snake(z)
        if i in skips:
            h = np.concatenate([x0, h], axis=1)
    cache["in_f1"] = h
    f1 = _lin(h, P, "feature_linear1", rb)
    cache["f1"] = f1
    if K > 1:
        s_in = np.concatenate([f1, aux], axis=1)
        cache["in_s"] = s_in
        zs = _lin(s_in, P, "scale_linears.0", rb)
        cache["z_s"] = zs
        a_s = snake(zs)
        cache["in_f2"] = a_s
        f2 = _lin(a_s, P, "feature_linear2", rb)
        cache["f2"] = f2
        p_in = np.concatenate([f1, f2], axis=1)
    else:
        p_in = f1
    cache["in_p"] = p_in
    zp = _lin(p_in, P, "pos_linears.0", rb)
    cache["z_p"] = zp
    ap = snake(zp)
    cache["in_rgb"] = ap
    raw = (ap @ P["rgb_linear.weight"].T + P["rgb_linear.bias"]).astype(F32)
    return raw, cache


def render(P, emb, K, **kw):
    """models/helpers.py:41-62 with normalize_type == 1 (sigmoid)."""
    raw, cache = mlp_forward(P, emb, K, **kw)
    return sigmoid(raw), cache


def mlp_backward(P, cache, draw, E=E_PER_PROPOSAL, D=8, skips=(4,), emulate_bf16=False, emulate_stash8=False):
    """Gradients of sum(raw * draw) w.r.t. every parameter (what autograd produces
    for networks.py:56-95).  No gradient flows to the embedding inputs.
    emulate_stash8 (with emulate_bf16): the operand roundings of the build's 8-bit stash in the WEIGHT-gradient products --
    gradients bf8 with the per-tile scale (dz8_quantise), layer inputs fp8 (embedding columns via their bf16 fragments) -- the
    data-gradient chain keeps its bf16 operands."""
    rb = emulate_bf16
    s8 = emulate_stash8
    K = cache["K"]
    G = {}
    W = P["feature_linear1.weight"].shape[0]

    def q(x):
        return bf16_round(x) if rb else x

    def sgrad(z):
        """snake'(z); stash8: as the unsigned byte the forward leaves for the backward chain (round(127.5 s') / 127.5)"""
        g = snake_grad(z)
        return (np.clip(np.rint(g.astype(np.float64) * 127.5), 0, 255) / 127.5).astype(F32) if s8 else g

    def qin(name, inp):
        if not s8:
            return q(inp)
        x = np.asarray(inp, dtype=F32).copy()
        emb_cols = {"periodic_linears.0": slice(0, x.shape[1]), "periodic_linears.5": slice(0, E),
                    "scale_linears.0": slice(W, x.shape[1])}.get(name)
        if name == "periodic_linears.5" and 5 - 1 not in skips:
            emb_cols = None
        if emb_cols is not None:
            x[:, emb_cols] = bf16_round(x[:, emb_cols])          # the embedding leaves the forward as bf16 fragments
        return fp8_round(x)

    def wg(name, dz, inp):
        if s8:
            dq = dz8_quantise(dz, draw)
            G[name + ".weight"] = (dq.astype(np.float64).T @ qin(name, inp).astype(np.float64)).astype(F32)
            G[name + ".bias"] = dq.sum(axis=0).astype(F32)
            return
        G[name + ".weight"] = (q(dz).T @ q(inp)).astype(F32)
        G[name + ".bias"] = dz.sum(axis=0).astype(F32)

    def dg(name, dz, cols=None):
        w = P[name + ".weight"]
        if cols is not None:
            w = w[:, cols]
        return (q(dz) @ q(w)).astype(F32)

    draw = np.asarray(draw, dtype=F32)
    if s8:
        dq = dz8_quantise(draw, draw)
        G["rgb_linear.weight"] = (dq.astype(np.float64).T @ fp8_round(cache["in_rgb"]).astype(np.float64)).astype(F32)
        G["rgb_linear.bias"] = dq.sum(axis=0).astype(F32)
    else:
        G["rgb_linear.weight"] = (draw.T @ cache["in_rgb"]).astype(F32)
        G["rgb_linear.bias"] = draw.sum(axis=0).astype(F32)
    d_ap = (draw @ P["rgb_linear.weight"]).astype(F32)
    dzp = d_ap * sgrad(cache["z_p"])
    wg("pos_linears.0", dzp, cache["in_p"])
    d_pin = dg("pos_linears.0", dzp)
    if K > 1:
        df1 = d_pin[:, :W]
        df2 = d_pin[:, W:]
        wg("feature_linear2", df2, cache["in_f2"])
        d_as = dg("feature_linear2", df2)
        dzs = d_as * sgrad(cache["z_s"])
        wg("scale_linears.0", dzs, cache["in_s"])
        df1 = df1 + dg("scale_linears.0", dzs, slice(0, W))
    else:
        df1 = d_pin
    wg("feature_linear1", df1, cache["in_f1"])
    dh = dg("feature_linear1", df1)
    for i in reversed(range(D)):
        if i in skips:
            dh = dh[:, E:]  # drop the part that would flow to the raw embedding
        dz = dh * sgrad(cache[f"z{i}"])
        wg(f"periodic_linears.{i}", dz, cache[f"in{i}"])
        if i > 0:
            dh = dg(f"periodic_linears.{i}", dz)
    return G


def mlp_macs_per_pixel(K, W=256, E=E_PER_PROPOSAL):
    """forward and train (fwd + wgrad + dgrad) MACs per pixel."""
    if K > 1:
        fwd = (K + 1) * E * W + 11 * W * W + 1.5 * W
        emb_part = (K + 1) * E * W
    else:
        fwd = 2 * E * W + 8.5 * W * W + 1.5 * W
        emb_part = 2 * E * W
    return fwd, 3 * fwd - emb_part


# --------------------------------------------------------------------------
# a8: adaptive robust pixel loss
#   models/mse_calculator.py:13-27, robust_loss_pytorch/adaptive.py:146-204,
#   distribution.py:90-114,143-210, general.py:85-118, cubic_spline.py:65-97,
#   util.py:64-95
# --------------------------------------------------------------------------
_SPLINE_CACHE = {}


def load_partition_spline(path=None):
    """(x_scale, values f32, tangents f32).  The table is this repo's own numerical
    re-derivation of log Z(alpha) (tools/gen_partition_spline.py), checked against
    the reference's resources/partition_spline.npz in tests/golden (distribution.py:
    129-141 loads the same three arrays)."""
    if path is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(os.path.dirname(here),
                            "learning-continuous-implicit-representation-for-near-periodic-patterns_amd",
                            "resources", "partition_spline.npz")
    if path not in _SPLINE_CACHE:
        with np.load(path) as f:
            _SPLINE_CACHE[path] = (float(f["x_scale"]), f["values"].astype(F32),
                                   f["tangents"].astype(F32))
    return _SPLINE_CACHE[path]


ALPHA_LO, ALPHA_HI = 0.001, 1.999
SCALE_LO, SCALE_INIT = 1e-5, 1.0
_SP_SHIFT = F32(np.log(np.expm1(F32(1.0))))  # util.py:90 inv_softplus(1)


def adaptive_params(latent_alpha, latent_scale):
    """adaptive.py:146-181: alpha = sigmoid(l)*(hi-lo)+lo ; scale = (ref-lo)*
    softplus(l + log(e-1)) + lo.  Also returns d alpha/d latent, d scale/d latent."""
    la = np.asarray(latent_alpha, dtype=F32)
    ls = np.asarray(latent_scale, dtype=F32)
    sg = sigmoid(la)
    alpha = sg * F32(ALPHA_HI - ALPHA_LO) + F32(ALPHA_LO)
    dalpha = sg * (F32(1) - sg) * F32(ALPHA_HI - ALPHA_LO)
    xs = ls + _SP_SHIFT
    sp = np.where(xs > 20, xs, np.log1p(np.exp(np.minimum(xs, F32(20))))).astype(F32)
    scale = F32(SCALE_INIT - SCALE_LO) * sp + F32(SCALE_LO)
    dscale = F32(SCALE_INIT - SCALE_LO) * sigmoid(xs)
    return alpha.astype(F32), scale.astype(F32), dalpha.astype(F32), dscale.astype(F32)


def _log_partition(alpha, spline):
    """distribution.py:90-114 (curve, alpha < 4 branch) + :143-169 + cubic_spline.py:65-97.
    Returns (logZ, dlogZ/dalpha)."""
    x_scale, vals, tans = spline
    a = alpha.astype(F32)
    den = np.abs(a - F32(2)) + F32(0.25)
    x = (F32(2.25) * a - F32(4.5)) / den + a + F32(2)
    # d/da for a < 2: 0.5625/(2.25-a)^2 + 1 ; for a > 2: (2.25*(a-1.75) - (2.25a-4.5))/(a-1.75)^2 + 1
    dx = np.where(a < 2, F32(0.5625) / (den * den), F32(0.5625) / (den * den)) + F32(1)
    xq = x * F32(x_scale)
    n = vals.shape[0]
    lo = np.floor(np.clip(xq, 0, n - 2)).astype(np.int64)
    t = (xq - lo.astype(F32)).astype(F32)
    t2 = t * t
    t3 = t * t2
    h01 = F32(-2) * t3 + F32(3) * t2
    h00 = F32(1) - h01
    h11 = t3 - t2
    h10 = h11 - t2 + t
    mid = vals[lo] * h00 + vals[lo + 1] * h01 + tans[lo] * h10 + tans[lo + 1] * h11
    dh01 = F32(-6) * t2 + F32(6) * t
    dh11 = F32(3) * t2 - F32(2) * t
    dh10 = dh11 - F32(2) * t + F32(1)
    dmid = (vals[lo + 1] - vals[lo]) * dh01 + tans[lo] * dh10 + tans[lo + 1] * dh11
    before = tans[0] * t + vals[0]
    after = tans[-1] * (t - F32(1)) + vals[-1]
    val = np.where(t < 0, before, np.where(t > 1, after, mid))
    dval = np.where(t < 0, tans[0], np.where(t > 1, tans[-1], dmid))
    return val.astype(F32), (dval * F32(x_scale) * dx).astype(F32)


_EPS32 = F32(np.finfo(np.float32).eps)


def robust_nll(x, alpha, scale, spline=None):
    """distribution.py:171-210 with general.py:85-118 ('otherwise' branch: alpha is
    confined to (0.001, 1.999) by adaptive.py so 0, 2, +-inf never occur).
    x (N,C); alpha, scale (1,C)."""
    spline = spline or load_partition_spline()
    x = np.asarray(x, dtype=F32)
    ssx = np.square(x / scale)
    beta = np.maximum(_EPS32, np.abs(alpha - F32(2)))
    asafe = np.where(alpha >= 0, F32(1), F32(-1)) * np.maximum(_EPS32, np.abs(alpha))
    rho = (beta / asafe) * (np.power(ssx / beta + F32(1), F32(0.5) * alpha) - F32(1))
    logz, _ = _log_partition(alpha, spline)
    return (rho + np.log(scale) + logz).astype(F32)


def robust_nll_grads(x, alpha, scale, spline=None):
    """Closed-form d nll / d(x, alpha, scale), elementwise (N,C).  Derivation (beta =
    2 - alpha, u = (x/c)^2/beta + 1, e = alpha/2, rho = beta/alpha (u^e - 1)):
      d rho/dx = x/c^2 u^(e-1);  d rho/dc = -x^2/c^3 u^(e-1)
      d rho/da = -(2/a^2)(u^e - 1) + (beta/a) u^e (ln(u)/2 + (a/2)(x/c)^2/(beta^2 u))
    plus d logZ/da from the spline and d log c/dc = 1/c."""
    spline = spline or load_partition_spline()
    x = np.asarray(x, dtype=F32)
    c = scale
    ssx = np.square(x / c)
    beta = np.maximum(_EPS32, np.abs(alpha - F32(2)))
    u = ssx / beta + F32(1)
    e = F32(0.5) * alpha
    ue = np.power(u, e)
    ue1 = ue / u
    dx = (x / (c * c)) * ue1
    dc = -(x * x) / (c * c * c) * ue1 + F32(1) / c
    _, dlogz = _log_partition(alpha, spline)
    da = (-(F32(2) / (alpha * alpha)) * (ue - F32(1))
          + (beta / alpha) * ue * (F32(0.5) * np.log(u) + e * ssx / (beta * beta * u))
          + dlogz)
    return dx.astype(F32), da.astype(F32), dc.astype(F32)


def img2mse(pred, gt, latent_alpha, latent_scale, mask=None, spline=None):
    """mse_calculator.py:13-27 with loss_type 'robust_loss_adaptive':
    d = pred - gt; d = d*m + (1-m)*d*0.3; mean(nll(d))."""
    d = np.asarray(pred, dtype=F32) - np.asarray(gt, dtype=F32)
    if mask is not None:
        m = np.asarray(mask, dtype=F32)
        d = d * m + (F32(1) - m) * d * F32(0.3)
    alpha, scale, _, _ = adaptive_params(latent_alpha, latent_scale)
    return F32(np.mean(robust_nll(d, alpha, scale, spline), dtype=np.float64))


def img2mse_grads(pred, gt, latent_alpha, latent_scale, mask=None, spline=None):
    """Returns (loss, dL/dpred (N,C), dL/dlatent_alpha (1,C), dL/dlatent_scale (1,C))."""
    pred = np.asarray(pred, dtype=F32)
    d = pred - np.asarray(gt, dtype=F32)
    w = np.ones_like(d)
    if mask is not None:
        m = np.asarray(mask, dtype=F32)
        w = (m + (F32(1) - m) * F32(0.3)) * np.ones_like(d)
        d = d * m + (F32(1) - m) * d * F32(0.3)
    alpha, scale, dalpha, dscale = adaptive_params(latent_alpha, latent_scale)
    nll = robust_nll(d, alpha, scale, spline)
    dx, da, dc = robust_nll_grads(d, alpha, scale, spline)
    inv = F32(1.0 / d.size)
    loss = F32(np.mean(nll, dtype=np.float64))
    dpred = (dx * w * inv).astype(F32)
    dla = (da.sum(axis=0, keepdims=True, dtype=np.float64) * dalpha * inv).astype(F32)
    dls = (dc.sum(axis=0, keepdims=True, dtype=np.float64) * dscale * inv).astype(F32)
    return loss, dpred, dla, dls


# --------------------------------------------------------------------------
# a14: Adam + LR schedule (models/helpers.py:164, NPP_completion/train.py:253-263,337)
# --------------------------------------------------------------------------
def img2mse_quad_grads(pred, gt, loss_type, mask=None):
    """img2mse for the non-adaptive --loss_type switches (models/mse_calculator.py:13-27): 'l2' -> mean(diff^2);
    'robust_loss' -> mean(lossfun(diff, alpha = 2, scale = 0.1)) with robust_loss_pytorch/general.py:88-91,116: for alpha == 2 the general
    loss IS 0.5 (x / scale)^2.  diff = x * mask + (1 - mask) * x * 0.3 (:16-17).  -> (loss, d loss / d pred)."""
    coef = {"l2": 1.0, "robust_loss": 0.5 / (0.1 * 0.1)}[loss_type]
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    d0 = pred - gt
    w = 1.0 if mask is None else np.asarray(mask, np.float64) + (1.0 - np.asarray(mask, np.float64)) * 0.3
    x = d0 * w
    return float(coef * np.mean(x * x)), (2.0 * coef * x * w / x.size).astype(np.float32)


def adam_init(P):
    return {"step": 0, "m": {k: np.zeros_like(v) for k, v in P.items()},
            "v": {k: np.zeros_like(v) for k, v in P.items()}}


def adam_step(P, G, st, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad / weight decay): p -= lr/(1-b1^t) * m / (sqrt(v)/
    sqrt(1-b2^t) + eps).  Parameters without a gradient are skipped."""
    st["step"] += 1
    t = st["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = F32(lr / bc1)
    for k, g in G.items():
        g = np.asarray(g, dtype=F32)
        m = st["m"][k] = (F32(b1) * st["m"][k] + F32(1 - b1) * g).astype(F32)
        v = st["v"][k] = (F32(b2) * st["v"][k] + F32(1 - b2) * g * g).astype(F32)
        denom = np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)
        P[k] = (P[k] - step_size * (m / denom)).astype(F32)
    return P


def lr_schedule(global_step, lrate=5e-4, lrate_decay=500, decay_rate=0.1):
    """train.py:256-262: lr <- lrate * 0.1^(global_step / (lrate_decay*100)), applied
    AFTER optimizer.step() and BEFORE global_step += 1 (train.py:337)."""
    return lrate * (decay_rate ** (global_step / (lrate_decay * 100)))


def psnr(pred, gt, mask=None):
    """-10 log10 mean((pred-gt)^2), range [0,1]; defined by the
    build (the reference's mse2psnr, mse_calculator.py:29, is unused)."""
    d = (np.asarray(pred, dtype=np.float64) - np.asarray(gt, dtype=np.float64)) ** 2
    if mask is not None:
        m = np.broadcast_to(np.asarray(mask, dtype=np.float64), d.shape)
        mse = (d * m).sum() / max(m.sum(), 1.0)
    else:
        mse = d.mean()
    return float(-10.0 * math.log10(max(mse, 1e-20)))


# --------------------------------------------------------------------------
# Synthetic workload -- shared by tests, smoke and bench
# --------------------------------------------------------------------------
def synthetic_image(H, W=None, seed=0, noise=0.03):
    """Lattice image with shifts d1=(dx,dy)=(40,8)s, d2=(-6,36)s, s=H/256; unknown
    centre rectangle rows [0.375H,0.625H) x cols [0.3125W,0.6875W).
    Returns img (H,W,3) f32 in [0,1], mask (H,W,1) f32 (1 = known)."""
    W = W or H
    s = H / 256.0
    d1 = np.array([40.0, 8.0]) * s
    d2 = np.array([-6.0, 36.0]) * s
    A = np.stack([d1, d2], axis=1)  # columns are the shifts in (x,y)
    Ainv = np.linalg.inv(A)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = np.einsum("ij,jhw->ihw", Ainv, np.stack([xx, yy]).astype(np.float64))
    u, v = uv[0], uv[1]
    rgb = np.stack([0.5 + 0.4 * np.cos(2 * np.pi * u) * np.cos(2 * np.pi * v),
                    0.5 + 0.4 * np.sin(2 * np.pi * u),
                    0.5 + 0.3 * np.cos(4 * np.pi * v)], axis=-1)
    rng = np.random.RandomState(seed)
    rgb = np.clip(rgb + rng.normal(0.0, noise, rgb.shape), 0.0, 1.0).astype(F32)
    mask = np.ones((H, W, 1), dtype=F32)
    mask[int(0.375 * H):int(0.625 * H), int(0.3125 * W):int(0.6875 * W)] = 0
    return rgb, mask


def synthetic_periodicity(H, K):
    """Top-K (angles, periods, shifts) for synthetic_image.  angle = 180 - atan2(dy,dx)
    of the OTHER shift, period = |d| sin(angle(d1,d2)) (NPP_proposal/feature_searching.py:
    144,309-327); proposals 2..K reuse the lattice with periods x {2, .5, 3, 1/3}."""
    s = H / 256.0
    d1 = np.array([40.0, 8.0]) * s
    d2 = np.array([-6.0, 36.0]) * s
    cross = abs(d1[0] * d2[1] - d1[1] * d2[0])
    p1 = cross / np.linalg.norm(d2)  # spacing of lines parallel to d2
    p2 = cross / np.linalg.norm(d1)
    a1 = 180.0 - math.degrees(math.atan2(d2[1], d2[0]))
    a2 = 180.0 - math.degrees(math.atan2(d1[1], d1[0]))
    mult = [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0]
    angles = np.array([[a1, a2]] * K, dtype=F32)
    periods = np.array([[p1 * m, p2 * m] for m in mult[:K]], dtype=F32)
    shifts = [[[float(d1[0]), float(d1[1])], [float(d2[0]), float(d2[1])]]] * K
    return angles, periods, shifts


def patch_size_from_period(periods_top1):
    """loaders/loaders.py:133-134."""
    mp = float(max(periods_top1))
    return int(np.clip(mp + (32 - mp % 32), 64, 160))
