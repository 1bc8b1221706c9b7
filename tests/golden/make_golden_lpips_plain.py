#!/usr/bin/env python3
"""g7b_lpips_plain.npz: the reference's LPIPS.forward(in0, in1, use_robust=False, normalize=True) (externel_lib/lpips/lpips.py:92-133,
the loop's head under --use_adaptive_perceptual_loss off) with its autograd gradient w.r.t. the first input's features, on the features
and vendored lin weights of g7_lpips.npz (same stand-in trunk: everything after the trunk is the reference's code).
Runs only where /root/reference exists; the committed .npz is what the tests read.

    python tests/golden/make_golden_lpips_plain.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, OUT, REF          # noqa: E402


def main():
    R = import_reference()
    import lpips.lpips as LL
    g7 = np.load(os.path.join(OUT, "g7_lpips.npz"))
    chns = [64, 128, 256, 512, 512]
    obj = LL.LPIPS.__new__(LL.LPIPS)
    torch.nn.Module.__init__(obj)
    obj.pnet_type, obj.pnet_tune, obj.pnet_rand, obj.spatial, obj.lpips, obj.version = "vgg", False, False, False, True, "0.1"
    obj.scaling_layer = LL.ScalingLayer()
    obj.chns, obj.L = chns, 5
    obj.adaptive_perceps = [R["adaptive"].AdaptiveLossFunction(num_dims=c, float_dtype=np.float32, device="cpu") for c in chns]
    obj.lins = torch.nn.ModuleList([LL.NetLinLayer(c, use_dropout=True) for c in chns])
    for i, l in enumerate(obj.lins):
        setattr(obj, f"lin{i}", l)
    obj.load_state_dict(torch.load(os.path.join(REF, "externel_lib/lpips/weights/v0.1/vgg.pth"), map_location="cpu"), strict=False)
    obj.eval()
    f0 = [torch.from_numpy(g7[f"f0_{k}"]).clone().requires_grad_(True) for k in range(5)]
    f1 = [torch.from_numpy(g7[f"f1_{k}"]) for k in range(5)]
    calls = {"n": 0}

    class Net:
        def forward(self, x):
            calls["n"] += 1
            return f0 if calls["n"] == 1 else f1
    obj.net = Net()
    in0 = torch.from_numpy(g7["in0"])
    val = obj.forward(in0, in0, use_robust=False, normalize=True)
    loss = torch.mean(val)                                     # train.py:249
    loss.backward()
    out = {"val": val.detach().numpy(), "loss": loss.detach().numpy()}
    for k in range(5):
        out[f"df0_{k}"] = f0[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g7b_lpips_plain.npz"), **out)
    print("loss", float(loss), "val", val.detach().numpy().ravel())


if __name__ == "__main__":
    main()
