#!/usr/bin/env python3
"""Golden G14: the segmentation criterion's LPIPS(net='alex', spatial=True) distance maps from the REFERENCE's own
LPIPS.forward (externel_lib/lpips/lpips.py:92-133, retPerLayer=True, use_robust=False, normalize=True) fed GRAYSCALE images
like NPP_segmentation/train.py:361-362.  torchvision's pretrained AlexNet is not available offline: the trunk
is an AlexNet-`features`-shaped torch stack with weights drawn from torch.Generator().manual_seed(SEED) in the order the test
re-draws them; the `lin` layers carry the vendored weights/v0.1/alex.pth.      python tests/golden/make_golden_segment.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, OUT  # noqa: E402

SEED = 20240
SHAPES = [(64, 3, 11), (192, 64, 5), (384, 192, 3), (256, 384, 3), (256, 256, 3)]


def alex_weights(seed=SEED):
    """[(weight (co,ci,k,k), bias (co,))] in torchvision's alexnet.features order (indices 0, 3, 6, 8, 10)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for co, ci, k in SHAPES:
        w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        b = torch.randn(co, generator=g) * 0.05
        out.append((w, b))
    return out


def main():
    R = import_reference()
    import lpips.lpips as LL
    chns = [64, 192, 384, 256, 256]
    obj = LL.LPIPS.__new__(LL.LPIPS)
    torch.nn.Module.__init__(obj)
    obj.pnet_type, obj.pnet_tune, obj.pnet_rand, obj.spatial, obj.lpips, obj.version = "alex", False, False, True, True, "0.1"
    obj.scaling_layer = LL.ScalingLayer()
    obj.chns, obj.L = chns, 5
    obj.adaptive_perceps = []
    obj.lins = torch.nn.ModuleList([LL.NetLinLayer(c, use_dropout=True) for c in chns])
    for i, l in enumerate(obj.lins):
        setattr(obj, f"lin{i}", l)
    obj.load_state_dict(torch.load("/root/reference/externel_lib/lpips/weights/v0.1/alex.pth", map_location="cpu"), strict=False)
    obj.eval()
    W = alex_weights()
    F = torch.nn.functional

    class Net:                                                   # pretrained_networks.py:56-94 on torchvision's layer list
        def forward(self, x):
            outs = []
            x = F.relu(F.conv2d(x, W[0][0], W[0][1], stride=4, padding=2)); outs.append(x)
            x = F.relu(F.conv2d(F.max_pool2d(x, 3, 2), W[1][0], W[1][1], padding=2)); outs.append(x)
            x = F.relu(F.conv2d(F.max_pool2d(x, 3, 2), W[2][0], W[2][1], padding=1)); outs.append(x)
            x = F.relu(F.conv2d(x, W[3][0], W[3][1], padding=1)); outs.append(x)
            x = F.relu(F.conv2d(x, W[4][0], W[4][1], padding=1)); outs.append(x)
            return outs
    obj.net = Net()
    g = torch.Generator().manual_seed(3)
    H = 128
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    base = 0.5 + 0.35 * torch.cos(2 * np.pi * xx / 16) * torch.cos(2 * np.pi * yy / 20)
    in0 = (base + 0.02 * torch.randn(H, H, generator=g)).clamp(0, 1)[None, None]
    in1 = in0.clone()
    in1[:, :, 40:90, 30:100] = torch.rand(50, 70, generator=g)          # a region the pattern does not explain
    with torch.no_grad():
        val, res = obj.forward(in0, in1, False, retPerLayer=True, normalize=True)
    out = {"in0": in0.numpy(), "in1": in1.numpy(), "val": val.numpy(), "seed": np.int64(SEED)}
    for k in range(5):
        out[f"map{k}"] = res[k].numpy()
        out[f"lin{k}"] = obj.lins[k].model[1].weight.detach().numpy().reshape(-1)
    np.savez_compressed(os.path.join(OUT, "g14_segment.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
