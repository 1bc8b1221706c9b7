#!/usr/bin/env python3
"""Is the stacked remapping fit bit-reproducible with the style terms on the side stream (StackedFit.style_side_stream = True)?
Runs the same 8-iteration stack of two remapping fits (LPIPS on) R times per setting and prints, per setting, the largest difference
of any network parameter / latent between run 0 and each later run, and the mean time per stacked iteration.

    python tools/r6_side_stream_repro.py [R]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402,F401  (aliases the package as npp_amd)
from test_gpu_stack import _remap_fits  # noqa: E402
from npp_amd.stack import StackedFit  # noqa: E402


def run(dev, side, lpips, iters=8, M=2):
    st = StackedFit(_remap_fits(dev, M, 256, 1, 4, lpips), ksplit=4)
    st.style_side_stream = side
    st.step_full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters - 1):
        st.step_full()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (iters - 1)
    out = []
    for f in st.fits:
        out += [f.net.params.clone(), f.net.latents.clone()] + [l.clone() for l in f.style.latents]
        if lpips:
            out.append(f.percepLoss._lat.clone())
    return out, ms


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    for lpips in (False, True):
        for side in (False, True):
            base, ms0 = run(dev, side, lpips)
            worst, mss = 0.0, [ms0]
            for _ in range(R - 1):
                o, ms = run(dev, side, lpips)
                mss.append(ms)
                worst = max(worst, max(float((a - b).abs().max()) for a, b in zip(base, o)))
            print(f"lpips={lpips!s:5} style_side_stream={side!s:5} runs={R} max|run_i - run_0|={worst:.3e} "
                  f"ms/iteration min={min(mss):.3f} median={sorted(mss)[len(mss) // 2]:.3f}", flush=True)


if __name__ == "__main__":
    main()
