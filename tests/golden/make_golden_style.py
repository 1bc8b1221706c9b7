#!/usr/bin/env python3
"""Golden G12: models/style_loss.py:37-74 VGG16FeatureExtractor.style_loss (use_adaptive=True) from the feature tensors on:
the class is instantiated without its torchvision trunk (pretrained weights absent) and fed prepared features,
so Gram matrices, the per-element AdaptiveLossFunction(num_dims = C^2), the 1 / (c w h) scaling and the means are the
reference's code.  Also the `weight` variant (:66-69).      python tests/golden/make_golden_style.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, OUT  # noqa: E402


def main():
    R = import_reference()
    import models.style_loss as SL
    chns, sizes, N = [16, 32, 48], [8, 4, 2], 3            # reduced channel counts (the kernels are generic in C); real: 64,128,256
    g = torch.Generator().manual_seed(21)
    obj = SL.VGG16FeatureExtractor.__new__(SL.VGG16FeatureExtractor)
    torch.nn.Module.__init__(obj)
    obj.use_adaptive = True
    obj.adaptives = [R["adaptive"].AdaptiveLossFunction(num_dims=c * c, float_dtype=np.float32, device="cpu") for c in chns]
    out = {"chns": np.array(chns), "sizes": np.array(sizes)}
    for i, ad in enumerate(obj.adaptives):
        with torch.no_grad():
            ad.latent_alpha.add_(0.4 * torch.randn(1, chns[i] ** 2, generator=g))
            ad.latent_scale.add_(0.4 * torch.randn(1, chns[i] ** 2, generator=g))
        out[f"la{i}"], out[f"ls{i}"] = ad.latent_alpha.detach().numpy(), ad.latent_scale.detach().numpy()
    A = [torch.relu(torch.randn(N, c, s, s, generator=g)).requires_grad_(True) for c, s in zip(chns, sizes)]
    B = [torch.relu(torch.randn(N, c, s, s, generator=g)) for c, s in zip(chns, sizes)]
    calls = {"n": 0}

    def fake_forward(image):
        calls["n"] += 1
        return A if calls["n"] % 2 == 1 else B
    obj.forward = fake_forward
    dummy = torch.zeros(N, 3, 16, 16)
    for tag, w in (("mean", None), ("weighted", torch.rand(N, generator=g))):
        for t in A:
            t.grad = None
        for ad in obj.adaptives:
            ad.latent_alpha.grad = None
            ad.latent_scale.grad = None
        loss = obj.style_loss(dummy, dummy, w)
        loss.backward()
        out[f"{tag}_loss"] = loss.detach().numpy()
        if w is not None:
            out[f"{tag}_w"] = w.numpy()
        for i in range(3):
            out[f"{tag}_dA{i}"] = A[i].grad.numpy().copy()
            out[f"{tag}_dla{i}"] = obj.adaptives[i].latent_alpha.grad.numpy().copy()
            out[f"{tag}_dls{i}"] = obj.adaptives[i].latent_scale.grad.numpy().copy()
    for i in range(3):
        out[f"A{i}"], out[f"B{i}"] = A[i].detach().numpy(), B[i].numpy()
    np.savez_compressed(os.path.join(OUT, "g12_style.npz"), **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if "loss" in k or "dA0" in k})


if __name__ == "__main__":
    main()
