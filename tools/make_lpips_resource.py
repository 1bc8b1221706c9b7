"""resources/lpips_lin_v0_1.npz: the five 1x1 'lin' layers of LPIPS v0.1 for the VGG16 and AlexNet trunks -- the weight files the
reference carries in its tree (externel_lib/lpips/weights/v0.1/{vgg,alex}.pth, 7 KB each; lpips.py:60-75 loads them by default),
re-saved as plain arrays {vgg,alex}_lin{0..4} of shape (C,).  Data, not code; run here where /root/reference is mounted."""
import os, sys
import numpy as np
import torch

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/externel_lib/lpips/weights/v0.1"
out = {}
for net in ("vgg", "alex"):
    sd = torch.load(os.path.join(src, f"{net}.pth"), map_location="cpu")
    for i in range(5):
        out[f"{net}_lin{i}"] = sd[f"lin{i}.model.1.weight"].reshape(-1).numpy().astype(np.float32)
pkg = [d for d in os.listdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) if d.endswith("_amd") and d != "npp_amd"][0]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pkg, "resources", "lpips_lin_v0_1.npz")
np.savez(dst, **out)
print(dst, {k: v.shape for k, v in out.items()})
