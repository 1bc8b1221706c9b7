#!/usr/bin/env python3
"""g4b_quad.npz: the reference's img2mse (models/mse_calculator.py:13-27) with the two non-adaptive --loss_type switches, 'l2' and
'robust_loss' (robust_loss_pytorch.general.lossfun, alpha = 2, scale = 0.1), with and without a mask: loss and d loss / d pred by the
reference's own autograd.  Runs only where /root/reference exists (the build container); the committed .npz is what the tests read.

    python tests/golden/make_golden_quad.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, OUT          # noqa: E402


def main():
    R = import_reference()
    msec = R["msec"]
    g = torch.Generator().manual_seed(5)
    out = {}
    pred = torch.rand(300, 3, generator=g)
    gt = torch.rand(300, 3, generator=g)
    mask = (torch.rand(300, 1, generator=g) > 0.3).float()
    out["pred"], out["gt"], out["mask"] = pred.numpy().copy(), gt.numpy(), mask.numpy()
    for lt in ("l2", "robust_loss"):
        for mtag, m in (("nomask", None), ("mask", mask)):
            p = pred.clone().requires_grad_(True)
            loss = msec.img2mse(p, gt, lt, None, m)
            loss.backward()
            out[f"{lt}_{mtag}_loss"] = loss.detach().numpy()
            out[f"{lt}_{mtag}_dpred"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4b_quad.npz"), **out)
    print({k: (v.shape, float(np.asarray(v).ravel()[0])) for k, v in out.items() if k.endswith("loss")})


if __name__ == "__main__":
    main()
