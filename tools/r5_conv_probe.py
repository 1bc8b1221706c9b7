"""Round 5: per-layer A/B of the trunk convolution forms inside ONE process (ops.tune flips the launcher's choice):
results of form B against form A (max |diff| in units of the output's largest value, share of differing elements) and the
launch time of each, (a) in a tight loop and (b) with the operands evicted between launches (a 600-MB fill: cold L2 / MALL, like
inside the iteration where a layer's operands were last touched ~0.5 ms / 1 GB of traffic ago).
usage: r5_conv_probe.py KEY A B [N P]      e.g.  r5_conv_probe.py conv_wink 0 1 12 96"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops  # noqa: E402

SHAPES = [("c1_1", 16, 64, 1), ("c1_2", 64, 64, 1), ("c2_1", 64, 128, 2), ("c2_2", 128, 128, 2), ("c3_1", 128, 256, 4),
          ("c3_x", 256, 256, 4), ("c4_1", 256, 512, 8), ("c4_x", 512, 512, 8)]


def main():
    key, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    N, P = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (12, 96)
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    trash = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    print(f"{key}: A = {va}, B = {vb}; N = {N} (forward) / {N // 2} (data gradient), P = {P}")
    for name, cin, cout, div in SHAPES:
        H = P // div
        w = (torch.randn(cout, 3 if cin == 16 else cin, 3, 3, generator=g) * (2.0 / (9 * max(cin, 3))) ** 0.5).to(dev)
        pf, pb = ops.conv_pack(w.contiguous(), in_natural=(cin == 16))
        bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
        img = torch.rand(N, cin if cin > 16 else 16, H, H, generator=g).to(dev)
        x = ops.trunk_alloc(N, cin, H, H, dev)
        ops.trunk_grad_in(img - 0.3, None, N, N, cin, H, H, x, as_f16=True)          # a flat fp16 tensor with a zero border
        for mode in ("fwd", "dgrad"):
            n_run = N if mode == "fwd" else N // 2
            if mode == "fwd":
                outs = [ops.trunk_alloc(N, cout, H, H, dev) for _ in range(2)]
                args = lambda y: (x, N, n_run, H, H, cin, cout, pf, bias, 0, None, y)      # noqa: E731
            else:
                if cin == 16:
                    continue
                gy = ops.trunk_alloc(N, cout, H, H, dev)
                ops.trunk_grad_in(torch.randn(N, cout, H, H, generator=g).to(dev), None, N, N, cout, H, H, gy)
                outs = [ops.trunk_alloc(N, cin, H, H, dev) for _ in range(2)]
                args = lambda y: (gy, N, n_run, H, H, cout, cin, pb, None, 1, x, y)       # noqa: E731
            res = []
            for v, y in zip((va, vb), outs):
                ops.tune(key, v)
                y.zero_()
                ops.conv3x3(*args(y))
                torch.cuda.synchronize()
                ts = []
                for cold in (False, True):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 12)]
                    for i in range(12):
                        if cold:
                            trash.fill_(i)
                        e[2 * i].record()
                        ops.conv3x3(*args(y))
                        e[2 * i + 1].record()
                    torch.cuda.synchronize()
                    ts.append(sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(2, 12))[5])
                res.append(ts)
            ya = outs[0].view(torch.float16 if mode == "fwd" else torch.bfloat16).float()
            yb = outs[1].view(torch.float16 if mode == "fwd" else torch.bfloat16).float()
            d = (ya - yb).abs()
            flops = 2 * 9 * (3 if cin == 16 else cin) * cout * n_run * H * H
            print(f"  {name:5s} {mode:5s}  A {res[0][0]:6.1f} / {res[0][1]:6.1f} us   B {res[1][0]:6.1f} / {res[1][1]:6.1f} us (tight / cold)   "
                  f"B cold = {flops / res[1][1] / 1e6 / 2500:.3f} of peak   max|B-A| / max|A| = {float(d.max() / ya.abs().max()):.2e}, "
                  f"differing {float((d > 0).float().mean()):.2e}", flush=True)
    ops.tune(key, 1)


if __name__ == "__main__":
    main()
