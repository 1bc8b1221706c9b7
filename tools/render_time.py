#!/usr/bin/env python3
"""Time the 1024^2 K=3 bf16 render (npp_mlp_fwd<render>) of the library named by NPP_LIB_PATH ('' = in-tree).
   python tools/render_time.py [reps]      -> one line: lib, ms, TFLOP/s, max|diff| vs the first call (determinism)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npp_amd  # noqa: E402
from npp_amd import EmbedCfg, synthetic as syn  # noqa: E402
from npp_amd.fit import CompletionFit  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
yy, xx = np.meshgrid(np.arange(1024, dtype=np.int32), np.arange(1024, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
a4, p4, _ = syn.synthetic_periodicity(1024, 3)
net = CompletionFit(*syn.synthetic_image(64), a4, p4, syn.SEED0_FREQS, syn.init_params(3, seed=0), device=dev).net
net.cfg = EmbedCfg.make(a4, p4, syn.SEED0_FREQS, (1024, 1024))
ref = net.render(grid).clone()
for _ in range(3):
    net.render(grid)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = net.render(grid)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps
macs = 3 * 480 * 256 + 4 * 256 * 256 + 2 * 256 * 256 + 3 * 256 * 256 + 3 * 256 * 128 + 128 * 3   # K=3 forward, padded k-steps
print(os.environ.get("NPP_LIB_PATH", "in-tree"), f"{ms:.4f} ms", f"{2 * macs * grid.shape[0] / ms / 1e9:.1f} TFLOP/s(approx)",
      "checksum", float(ref.double().sum()), "repeat-diff", float((out - ref).abs().max()))
