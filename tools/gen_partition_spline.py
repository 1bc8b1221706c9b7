#!/usr/bin/env python3
"""Numerically re-derive the cubic-Hermite spline of log Z(alpha) used by the adaptive
robust loss (Barron, "A General and Adaptive Robust Loss Function", CVPR 2019, eq. 16-18).

The reference ships this table as externel_lib/robust_loss_pytorch/resources/
partition_spline.npz and loads it at distribution.py:129-141.  We do not copy that file:
this script recomputes it from the definition

    Z(alpha) = integral_{-inf}^{inf} exp(-rho(x, alpha, 1)) dx

on the same knot grid the reference's interpolation expects (knot k at curve value
x_k = k / x_scale, alpha_k = inv_partition_spline_curve(x_k), distribution.py:117-126;
tangents are per knot spacing, cubic_spline.py:24-40).  Only x in [0, 8] (alpha in [0, 4])
is generated: adaptive.py confines alpha to (0.001, 1.999), i.e. x < 4.

Run:  python tools/gen_partition_spline.py [--check /root/reference/...npz]
Writes <package>/resources/partition_spline.npz  (x_scale, values f64, tangents f64).
"""
import argparse
import os
import sys

import numpy as np
from scipy import integrate

X_SCALE = 1024
X_MAX = 8.0


def inv_curve(x):
    x = np.asarray(x, dtype=np.float64)
    lo = 0.5 * x + 1.25 - np.sqrt(np.maximum(1.5625 - x + 0.25 * x * x, 0.0))
    hi = 0.5 * x - 1.25 + np.sqrt(np.maximum(9.5625 - 3.0 * x + 0.25 * x * x, 0.0))
    return np.where(x <= 4.0, lo, hi)


def d_inv_curve(x):
    x = np.asarray(x, dtype=np.float64)
    s_lo = np.sqrt(np.maximum(1.5625 - x + 0.25 * x * x, 1e-300))
    s_hi = np.sqrt(np.maximum(9.5625 - 3.0 * x + 0.25 * x * x, 1e-300))
    lo = 0.5 - (-1.0 + 0.5 * x) / (2.0 * s_lo)
    hi = 0.5 + (-3.0 + 0.5 * x) / (2.0 * s_hi)
    return np.where(x <= 4.0, lo, hi)


def rho(x, a):
    if a == 0.0:
        return np.log1p(0.5 * x * x)
    if a == 2.0:
        return 0.5 * x * x
    b = abs(a - 2.0)
    return (b / a) * (np.power(x * x / b + 1.0, 0.5 * a) - 1.0)


def drho_da(x, a):
    b = abs(a - 2.0)
    db = 1.0 if a > 2.0 else -1.0
    u = x * x / b + 1.0
    e = 0.5 * a
    ue = np.power(u, e)
    return ((db * a - b) / (a * a)) * (ue - 1.0) + (b / a) * ue * (
        0.5 * np.log(u) + e * (-(x * x) / (b * b)) * db / u)


def _quad(f):
    # even integrand: 2 * int_0^inf, split for accuracy of the heavy tails
    v = 0.0
    for lo, hi in ((0.0, 1.0), (1.0, 10.0), (10.0, 1e3), (1e3, np.inf)):
        r, _ = integrate.quad(f, lo, hi, epsabs=1e-13, epsrel=1e-12, limit=400)
        v += r
    return 2.0 * v


def log_z(a):
    return np.log(_quad(lambda x: np.exp(-rho(x, a))))


def dlog_z(a):
    z = _quad(lambda x: np.exp(-rho(x, a)))
    dz = _quad(lambda x: -drho_da(x, a) * np.exp(-rho(x, a)))
    return dz / z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", default=None, help="reference npz to compare against")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    n = int(X_MAX * X_SCALE) + 1
    xs = np.arange(n, dtype=np.float64) / X_SCALE
    alphas = inv_curve(xs)
    alphas[0] = 0.0
    vals = np.empty(n)
    tans = np.empty(n)
    for k in range(n):
        a = float(alphas[k])
        vals[k] = log_z(a)
        if a < 1e-3 or abs(a - 2.0) < 1e-3:
            # closed form is singular at alpha in {0, 2}: 4th-order central difference in x
            h = 0.25 / X_SCALE
            xk = xs[k]
            pts = [xk - 2 * h, xk - h, xk + h, xk + 2 * h]
            if pts[0] < 0:  # one-sided at x = 0
                f = [log_z(float(inv_curve(xk + i * h))) for i in range(5)]
                d = (-25 * f[0] + 48 * f[1] - 36 * f[2] + 16 * f[3] - 3 * f[4]) / (12 * h)
            else:
                f = [log_z(float(inv_curve(p))) for p in pts]
                d = (f[0] - 8 * f[1] + 8 * f[2] - f[3]) / (12 * h)
            tans[k] = d / X_SCALE
        else:
            tans[k] = dlog_z(a) * float(d_inv_curve(xs[k])) / X_SCALE
        if k % 1024 == 0:
            print(f"knot {k}/{n} alpha={a:.5f} logZ={vals[k]:.9f} tan={tans[k]:.3e}", flush=True)
    out = args.out or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
        "learning-continuous-implicit-representation-for-near-periodic-patterns_amd",
        "resources", "partition_spline.npz")
    np.savez(out, x_scale=np.int64(X_SCALE), values=vals, tangents=tans)
    print("wrote", out)
    if args.check:
        ref = np.load(args.check)
        rv, rt = ref["values"][:n], ref["tangents"][:n]
        print("max |values - ref|   =", np.abs(vals - rv).max())
        print("max |tangents - ref| =", np.abs(tans - rt).max(), "(ref tangent scale",
              np.abs(rt).max(), ")")
    return 0


if __name__ == "__main__":
    sys.exit(main())
