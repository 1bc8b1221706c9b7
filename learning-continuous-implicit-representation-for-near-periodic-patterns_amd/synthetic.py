"""The synthetic workload of SURVEY.md 8d (what bench.py, the smoke test and the drivers' self-tests fit): a noisy colour
lattice with a known periodicity and a rectangular hole, its top-K (angles, periods, shifts), the seed-0 Fourier
frequencies, nn.Linear-default initial weights and the algorithmic MAC counts.  Product-side definition: the oracle keeps its
own copy for the tests (tests/test_synthetic.py checks the two agree), and bench.py touches oracle/ only inside cpu_baseline."""
import math

import numpy as np

from ._lib import param_layout

E_PER_PROPOSAL = 462
# torch.manual_seed(0); torch.normal(0, 1, (10, 1)) * 10 on the CPU generator: what models/embedder.py:26 draws first in a fresh
# seeded process (tests/golden/g1_embed.npz 'freqs')
SEED0_FREQS = (15.409960746765137, -2.93428897857666, -21.787893295288086, 5.68431282043457, -10.845223426818848,
               -13.985954284667969, 4.033468246459961, 8.380263328552246, -7.192575931549072, -4.033435344696045)


def synthetic_image(H, W=None, seed=0, noise=0.03):
    """Lattice with shifts d1 = (dx, dy) = (40, 8) s, d2 = (-6, 36) s, s = H / 256; rgb = smooth functions of the lattice
    coordinates + N(0, noise^2), clipped to [0, 1]; unknown centre rectangle rows [0.375 H, 0.625 H) x cols [0.3125 W, 0.6875 W).
    -> img (H, W, 3) f32, mask (H, W, 1) f32 (1 = known)."""
    W = W or H
    s = H / 256.0
    A = np.stack([np.array([40.0, 8.0]) * s, np.array([-6.0, 36.0]) * s], axis=1)      # columns: the shifts in (x, y)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = np.einsum("ij,jhw->ihw", np.linalg.inv(A), np.stack([xx, yy]).astype(np.float64))
    u, v = uv[0], uv[1]
    rgb = np.stack([0.5 + 0.4 * np.cos(2 * np.pi * u) * np.cos(2 * np.pi * v), 0.5 + 0.4 * np.sin(2 * np.pi * u),
                    0.5 + 0.3 * np.cos(4 * np.pi * v)], axis=-1)
    rng = np.random.RandomState(seed)
    rgb = np.clip(rgb + rng.normal(0.0, noise, rgb.shape), 0.0, 1.0).astype(np.float32)
    mask = np.ones((H, W, 1), dtype=np.float32)
    mask[int(0.375 * H):int(0.625 * H), int(0.3125 * W):int(0.6875 * W)] = 0
    return rgb, mask


def synthetic_periodicity(H, K):
    """Top-K (angles, periods, shifts) of synthetic_image: angle = 180 - atan2(dy, dx) of the OTHER shift, period = |d| sin(angle
    between the shifts) (NPP_proposal/feature_searching.py:144,309-327); proposals 2..K reuse the lattice with periods x {2, .5, 3, 1/3}."""
    s = H / 256.0
    d1, d2 = np.array([40.0, 8.0]) * s, np.array([-6.0, 36.0]) * s
    cross = abs(d1[0] * d2[1] - d1[1] * d2[0])
    p1, p2 = cross / np.linalg.norm(d2), cross / np.linalg.norm(d1)
    a1 = 180.0 - math.degrees(math.atan2(d2[1], d2[0]))
    a2 = 180.0 - math.degrees(math.atan2(d1[1], d1[0]))
    mult = [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0]
    angles = np.array([[a1, a2]] * K, dtype=np.float32)
    periods = np.array([[p1 * m, p2 * m] for m in mult[:K]], dtype=np.float32)
    shifts = [[[float(d1[0]), float(d1[1])], [float(d2[0]), float(d2[1])]]] * K
    return angles, periods, shifts


def init_params(K, seed=0, width=256):
    """nn.Linear default init (U(-1/sqrt(in), 1/sqrt(in)) for weight and bias) for every tensor of the parameter blob, from a
    NumPy stream: initial weights are an explicit input of the path."""
    layout, _ = param_layout(K, width)
    rng = np.random.RandomState(seed)
    P, bound = {}, 1.0
    for name, _, rows, cols in layout:
        if name.endswith("weight"):
            bound = 1.0 / math.sqrt(cols)
            P[name] = rng.uniform(-bound, bound, size=(rows, cols)).astype(np.float32)
        else:
            P[name] = rng.uniform(-bound, bound, size=(rows * cols,)).astype(np.float32)
    return P


def mlp_macs_per_pixel(K, W=256, E=E_PER_PROPOSAL):
    """forward and training (forward + weight-gradient + data-gradient) MACs per pixel; no gradient flows to the
    embedding inputs."""
    if K > 1:
        fwd, emb_part = (K + 1) * E * W + 11 * W * W + 1.5 * W, (K + 1) * E * W
    else:
        fwd, emb_part = 2 * E * W + 8.5 * W * W + 1.5 * W, 2 * E * W
    return fwd, 3 * fwd - emb_part
