#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8 f1 (proposal-ranking fits), from the REFERENCE's own modules (build container only):
  g10_light.npz  models/embedder.py is_search embedders (2-D Fourier 42-dim, periodic 20-dim; :52-54,:76-86),
                 models/networks.py:176-263 NPP_Net_light (D=4; W=64 stored in full, forward + every parameter gradient
                 through sigmoid + a quadratic loss), externel_lib/lpips/lpips.py:92-133 LPIPS.forward(use_robust=False)
                 with a stand-in trunk and the vendored lin weights.
    python tests/golden/make_golden_light.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, REF, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS  # noqa: E402


def main():
    R = import_reference()
    emb, nets = R["emb"], R["nets"]
    out = {}
    # ---- is_search embedders --------------------------------------------------------------
    torch.manual_seed(0)
    res = (211, 325)
    embedder, freq_nerf = emb.get_embedder(10, 0, res, is_search=True)
    out["freq_nerf"] = np.int64(freq_nerf)
    out["freqs"] = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
    angles, periods = np.array([80.54, 168.69], np.float32), np.array([40.77, 36.48], np.float32)
    ep, in_ch_p = emb.get_embedder(10, 0, res, selected_angles=torch.Tensor(angles), selected_periods=torch.Tensor(periods),
                                   freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, is_search=True)
    out["input_ch_periodic"] = np.int64(in_ch_p)
    rng = np.random.RandomState(2)
    coords = np.stack([rng.randint(0, res[0], 60), rng.randint(0, res[1], 60)], 1)
    coords = np.concatenate([coords, [[0, 0], [0, res[1] - 1], [res[0] - 1, 0], [res[0] - 1, res[1] - 1]]], 0)
    out["coords"], out["res"], out["angles"], out["periods"] = coords, np.array(res), angles, periods
    out["pos_emb"] = embedder.embed(torch.Tensor(coords).clone()).numpy()          # search.py:104 (clone: embed normalises in place)
    out["per_emb"] = ep.embed(torch.Tensor(coords)).numpy()                        # search.py:107
    # ---- NPP_Net_light, D = 4 (searching_config netdepth), W = 64, snake --------------------
    torch.manual_seed(1)
    W = 64
    net = nets.NPP_Net_light(D=4, W=W, input_ch=int(freq_nerf), input_ch_periodic=int(in_ch_p), freq_scales=FREQ_SCALES,
                             freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, output_ch=3, skips=[4], activation="snake")
    for k, v in net.state_dict().items():
        out["sd." + k] = v.numpy().copy()
    x = torch.from_numpy(out["pos_emb"]).clone()
    xp = torch.from_numpy(out["per_emb"]).clone()
    raw = net(x, xp)
    pred = torch.sigmoid(raw)                                                      # helpers.py:55-56
    tgt = torch.from_numpy(rng.rand(coords.shape[0], 3).astype(np.float32))
    loss = ((pred - tgt) ** 2).mean()
    loss.backward()
    out["raw"], out["pred"], out["tgt"], out["loss"] = raw.detach().numpy(), pred.detach().numpy(), tgt.numpy(), loss.detach().numpy()
    for k, p in net.named_parameters():
        out["grad." + k] = (p.grad.numpy().copy() if p.grad is not None else np.zeros(0, np.float32))
    # ---- LPIPS.forward(use_robust=False) with a stand-in trunk (search.py:193) ----------------
    import lpips.lpips as LL
    chns, sizes = [64, 128, 256, 512, 512], [16, 8, 4, 2, 1]
    g = torch.Generator().manual_seed(9)
    obj = LL.LPIPS.__new__(LL.LPIPS)
    torch.nn.Module.__init__(obj)
    obj.pnet_type, obj.pnet_tune, obj.pnet_rand, obj.spatial, obj.lpips, obj.version = "vgg", False, False, False, True, "0.1"
    obj.scaling_layer = LL.ScalingLayer()
    obj.chns, obj.L = chns, 5
    obj.lins = torch.nn.ModuleList([LL.NetLinLayer(c, use_dropout=True) for c in chns])
    for i, l in enumerate(obj.lins):
        setattr(obj, f"lin{i}", l)
    obj.load_state_dict(torch.load(os.path.join(REF, "externel_lib/lpips/weights/v0.1/vgg.pth"), map_location="cpu"), strict=False)
    obj.eval()
    N = 1
    f0 = [torch.relu(torch.randn(N, c, s, s, generator=g)) for c, s in zip(chns, sizes)]
    f1 = [torch.relu(torch.randn(N, c, s, s, generator=g)) for c, s in zip(chns, sizes)]
    calls = {"n": 0}

    class Net:
        def forward(self, x):
            calls["n"] += 1
            return f0 if calls["n"] == 1 else f1
    obj.net = Net()
    val = obj.forward(torch.rand(N, 3, 16, 16, generator=g), torch.rand(N, 3, 16, 16, generator=g), False)
    out["lp_val"] = val.detach().numpy()
    for kk in range(5):
        out[f"lp_f0_{kk}"], out[f"lp_f1_{kk}"] = f0[kk].numpy(), f1[kk].numpy()
        out[f"lp_lin{kk}"] = obj.lins[kk].model[1].weight.detach().numpy().reshape(-1)
    np.savez_compressed(os.path.join(OUT, "g10_light.npz"), **out)
    print("g10_light.npz:", {k: v.shape for k, v in out.items() if k.startswith(("pos", "per", "raw", "lp_val"))}, "freq_nerf", freq_nerf, "in_ch_p", in_ch_p)
    print([k for k in out if k.startswith("sd.")])


if __name__ == "__main__":
    main()
