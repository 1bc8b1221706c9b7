"""Round-4 debug probe: npp_mlp_wgrad of an A/B library against the in-tree one ON THE SAME STASH, tensor by tensor and slab by slab:
   python tools/r4_wgrad_diff.py build_ab/libnpp_x.so [rows] [ksplit]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops, synthetic as syn          # noqa: E402
from npp_amd.model import NPPNet                   # noqa: E402
from npp_amd import _lib                            # noqa: E402

if sys.argv[1] in ("tile128", "w16"):       # the in-tree library's 256 x 128-tile kernel (NPP_WGRAD_TILE_N=128, read at a library instance's first call):
    import shutil                   # a second instance of the same file, first called after the variable is set below
    shutil.copy(_lib.LIB_PATH, "/tmp/libnpp_tile128.so")
    other = C.CDLL("/tmp/libnpp_tile128.so")
else:
    other = C.CDLL(os.path.abspath(sys.argv[1]))
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 704
ks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
other.npp_mlp_wgrad.restype = C.c_int
other.npp_mlp_wgrad.argtypes = _lib.SYMBOLS["npp_mlp_wgrad"][1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 256, 3
angles, periods, _ = syn.synthetic_periodicity(H, K)
net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0), device=dev, ksplit=ks)
g = torch.Generator().manual_seed(1)
c = torch.randint(0, H, (rows, 2), generator=g, dtype=torch.int32).to(dev)
gt = torch.rand(rows, 3, generator=g).to(dev)
net.zero_grad()
net.forward_train(c)
ws = net.workspace(rows)
ws["dpred"].zero_()
net.pixel_loss(rows, rows, gt)
net.backward(rows)
torch.cuda.synchronize()
ref = ws["gslabs"].clone()
if sys.argv[1] in ("tile128", "w16"):
    os.environ["NPP_WGRAD_TILE_N"] = "128" if sys.argv[1] == "tile128" else "16"
out = torch.full_like(ref, float("nan"))
rc = other.npp_mlp_wgrad(ws["dzT"].data_ptr(), ws["actT"].data_ptr(), rows, K, 256, ks, out.data_ptr(), None)
torch.cuda.synchronize()
print("rc", rc, "rows", rows, "ksplit", ks)
stride = ref.numel() // ks
R, O = ref.view(ks, stride).cpu().numpy(), out.view(ks, stride).cpu().numpy()
bad = 0
for name, off, r, cdim in net.layout:
    for s in range(ks):
        a, b = R[s, off:off + r * cdim].reshape(r, cdim), O[s, off:off + r * cdim].reshape(r, cdim)
        d = np.abs(a - b)
        if not np.array_equal(a, b) and not (len(sys.argv) > 4 and np.allclose(a, b, rtol=float(sys.argv[4]), atol=1e-12)):
            bad += 1
            rr, cc = np.nonzero(d > 1e-6 * (np.abs(a).max() + 1e-30))
            print(f"{name:28s} slab {s}: max|d| {np.nanmax(d):.3e} of {np.abs(a).max():.3e}  nan {int(np.isnan(b).sum())}  "
                  f"rows {rr.min() if rr.size else -1}..{rr.max() if rr.size else -1} cols {cc.min() if cc.size else -1}..{cc.max() if cc.size else -1}  "
                  f"n_bad {rr.size} / {a.size}")
print("differing (tensor, slab) pairs:", bad)
