#!/usr/bin/env python3
"""g10c_light_init.npz: the weights a candidate fit of NPP_proposal/search.py STARTS from, produced by the reference's own code path
(build container only): torch.manual_seed(0) (search.py:91) -> get_embedder(multires, i_embed, res, is_search=True) (helpers.py:84: draws
the 10 Gaussian Fourier frequencies from the global generator, embedder.py:26) -> the is_search periodic embedder -> NPP_Net_light(D = 4,
W = 256) (helpers.py:100-103).  Stored: the Fourier frequencies and, per parameter tensor, its first 16 values, sum and sum of squares.
    python tests/golden/make_golden_light_init.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS  # noqa: E402


def main():
    R = import_reference()
    emb, nets = R["emb"], R["nets"]
    res = (211, 325)
    out = {}
    torch.manual_seed(0)
    embedder, freq_nerf = emb.get_embedder(10, 0, res, is_search=True)
    ep, in_ch_p = emb.get_embedder(10, 0, res, selected_angles=torch.Tensor([80.54, 168.69]), selected_periods=torch.Tensor([40.77, 36.48]),
                                   freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, is_search=True)
    net = nets.NPP_Net_light(D=4, W=256, input_ch=int(freq_nerf), input_ch_periodic=int(in_ch_p), freq_scales=FREQ_SCALES,
                             freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, output_ch=3, skips=[4], activation="snake")
    out["freqs"] = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    for k, v in net.state_dict().items():
        a = v.numpy().astype(np.float64).reshape(-1)
        out["head." + k] = v.numpy().reshape(-1)[:16].copy()
        out["sum." + k] = np.float64(a.sum())
        out["sq." + k] = np.float64((a * a).sum())
        out["shape." + k] = np.array(v.shape, np.int64)
    np.savez(os.path.join(OUT, "g10c_light_init.npz"), **out)
    print("g10c_light_init.npz:", len(out), "arrays; freqs[:3]", out["freqs"][:3])


if __name__ == "__main__":
    main()
