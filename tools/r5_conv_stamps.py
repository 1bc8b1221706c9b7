"""Round 5: in-kernel timeline of the trunk convolution kernels (diagnostic build: tools/build_variant.sh diag "-DNPP_DIAG -I tools",
run with NPP_LIB_PATH=build_ab/libnpp_diag.so).  Per layer shape and kernel form: when the workgroups START relative to the first
one (s_memrealtime, 10-ns ticks), and how many shader cycles each spends before its main loop (operand prologue), in it, in the
epilogue and waiting for its stores -- against the launch's duration between two HIP events.
usage: r5_conv_stamps.py [N P]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npp_amd  # noqa: E402
from npp_amd import ops  # noqa: E402

SHAPES = [("c1_1", 16, 64, 1), ("c1_2", 64, 64, 1), ("c2_1", 64, 128, 2), ("c2_2", 128, 128, 2), ("c3_1", 128, 256, 4), ("c3_x", 256, 256, 4)]


def main():
    N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 96)
    dev = torch.device("cuda")
    L = npp_amd.lib()
    L.npp_diag_set_stamps.argtypes = [C.c_void_p, C.c_longlong]
    L.npp_diag_set_stamps.restype = None
    g = torch.Generator(device="cpu").manual_seed(0)
    trash = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    nwords = 8 * 8 * 8192
    stamps = torch.zeros(nwords, dtype=torch.int64, device=dev)
    for name, cin, cout, div in SHAPES:
        H = P // div
        w = (torch.randn(cout, 3 if cin == 16 else cin, 3, 3, generator=g) * (2.0 / (9 * max(cin, 3))) ** 0.5).to(dev)
        pf, pb = ops.conv_pack(w.contiguous(), in_natural=(cin == 16))
        bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
        x = ops.trunk_alloc(N, cin, H, H, dev)
        ops.trunk_grad_in(torch.rand(N, max(cin, 16), H, H, generator=g).to(dev) - 0.3, None, N, N, max(cin, 16), H, H, x, as_f16=True)
        y = ops.trunk_alloc(N, cout, H, H, dev)
        for wink in (0, 1):
            if wink and cin < 128:
                continue
            ops.tune("conv_wink", wink)
            for cold in (False, True):
                for _ in range(3):
                    ops.conv3x3(x, N, N, H, H, cin, cout, pf, bias, 0, None, y)
                torch.cuda.synchronize()
                if cold:
                    trash.fill_(1)
                stamps.zero_()
                torch.cuda.synchronize()
                L.npp_diag_set_stamps(stamps.data_ptr(), nwords)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv3x3(x, N, N, H, H, cin, cout, pf, bias, 0, None, y)
                e1.record()
                torch.cuda.synchronize()
                L.npp_diag_set_stamps(None, 0)
                s = stamps.cpu().numpy().reshape(-1, 8)
                s = s[s[:, 1] != 0]
                rt0 = s[:, 0].astype(np.float64)
                start_us = (rt0 - rt0.min()) / 100.0
                ph = np.stack([s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4], s[:, 5] - s[:, 1]], 1).astype(np.float64)
                xcc = (s[:, 7] >> 32) & 0xf
                hw = s[:, 7] & 0xffffffff
                cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)              # (cu_id, se_id, xcc)
                n_cu = len(set(cu.tolist()))
                waves_per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
                # end of each wave in us after the first start, with the clock estimated from the longest wave: unknown -> report cycles
                q = lambda v: f"{np.percentile(v, 50):7.0f} / {np.percentile(v, 95):7.0f} / {v.max():7.0f}"   # noqa: E731
                print(f"{name} wink={wink} {'cold' if cold else 'warm'}: events {e0.elapsed_time(e1) * 1e3:6.1f} us | {len(s)} waves on {n_cu} CUs "
                      f"(waves per CU {waves_per_cu.min()}..{waves_per_cu.max()}) | wave start after the first: p50 {np.percentile(start_us, 50):.2f} "
                      f"p95 {np.percentile(start_us, 95):.2f} max {start_us.max():.2f} us")
                print(f"      cycles p50 / p95 / max:  prologue {q(ph[:, 0])} | main {q(ph[:, 1])} | epilogue {q(ph[:, 2])} | store drain {q(ph[:, 3])} | "
                      f"whole wave {q(ph[:, 4])}", flush=True)
    ops.tune("conv_wink", 1)
    # ---- the fused pair (conv1_1 -> conv1_2 -> pool1): phases = operand prologue | conv a | stage conv b's first weights | conv b | epilogue | store drain
    H = P
    wa = (torch.randn(64, 3, 3, 3, generator=g) * 0.3).to(dev)
    wb = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).to(dev)
    pfa, _ = ops.conv_pack(wa.contiguous(), in_natural=True)
    pfb, _ = ops.conv_pack(wb.contiguous())
    ba, bb = (torch.randn(64, generator=g) * 0.1).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
    x = ops.trunk_alloc(N, 16, H, H, dev)
    ops.trunk_grad_in(torch.rand(N, 16, H, H, generator=g).to(dev) - 0.3, None, N, N, 16, H, H, x, as_f16=True)
    ya, yb, yp = ops.trunk_alloc(N, 64, H, H, dev), ops.trunk_alloc(N, 64, H, H, dev), ops.trunk_alloc(N, 64, H // 2, H // 2, dev)
    for keep in (N, N // 2):
        for cold in (False, True):
            for _ in range(3):
                ops.conv_pair_fwd(x, N, N, keep, H, H, 16, 64, 64, pfa, ba, pfb, bb, ya, yb, yp)
            torch.cuda.synchronize()
            if cold:
                trash.fill_(1)
            stamps.zero_()
            torch.cuda.synchronize()
            L.npp_diag_set_stamps(stamps.data_ptr(), nwords)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_pair_fwd(x, N, N, keep, H, H, 16, 64, 64, pfa, ba, pfb, bb, ya, yb, yp)
            e1.record()
            torch.cuda.synchronize()
            L.npp_diag_set_stamps(None, 0)
            s = stamps.cpu().numpy().reshape(-1, 8)
            s = s[s[:, 1] != 0]
            start_us = (s[:, 0] - s[:, 0].min()) / 100.0
            ph = np.stack([s[:, k + 1] - s[:, k] for k in range(1, 6)] + [s[:, 6] - s[:, 1]], 1).astype(np.float64)
            q = lambda v: f"{np.percentile(v, 50):6.0f}/{v.max():6.0f}"   # noqa: E731
            print(f"pair c1 keep={keep} {'cold' if cold else 'warm'}: events {e0.elapsed_time(e1) * 1e3:6.1f} us | {len(s)} waves | wave start after the first: "
                  f"p50 {np.percentile(start_us, 50):.2f} max {start_us.max():.2f} us | cycles p50/max: prologue {q(ph[:, 0])} | conv a {q(ph[:, 1])} | "
                  f"conv b {q(ph[:, 2])} (incl. first weight stage) | -- {q(ph[:, 3])} epilogue | drain {q(ph[:, 4])} | whole {q(ph[:, 5])}", flush=True)


if __name__ == "__main__":
    main()
