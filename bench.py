#!/usr/bin/env python3
"""bench.py -- fitted pixels/s of the NPP-Net optimisation step on MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "c2"): one 512x512 completion image, top-3
periodicity proposals, 256-wide x 8-layer NPP_Net, bf16 MFMA operands / fp32 accumulate.
A step = ONE optimisation iteration of NPP_completion/train.py:133-264 over
N_rand = 8192 pixel rows + patch_num * P^2 = 2 * 96^2 patch rows (P from loaders.py:133-134):
fused embedder + MLP forward (with stashes) -> adaptive robust loss -> backward chain ->
grouped wgrad -> Adam (+ weight re-pack).  Inputs (coordinates, ground truth) are resident
in HBM before the timed region; nothing is cached between steps.
`value` = rows fitted per second, summed over ranks (each rank fits its own image: weak
scaling, no data-path collective; one all_gather of the fitted images after the loop).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512, help="image side (512 = BASELINE c2)")
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the c4 embedder and full-loop extras")
    ap.add_argument("--ksplit", type=int, default=12)
    ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per iteration (measured: no gain, the eager launches already run ahead of the GPU)")
    return ap.parse_args()


def cpu_baseline(K, H, seconds_target=15.0):
    """The oracle (NumPy fp32 port of the reference path) timed on this host's cores on a
    bounded sample of the same workload: full train steps (embed + fwd + robust loss + bwd +
    Adam) over 2048-row batches of the same image."""
    import oracle
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    img, mask = oracle.synthetic_image(H)
    P = oracle.init_params(K, seed=0)
    st = oracle.adam_init(P)
    rng = np.random.RandomState(0)
    rows = 2048
    la = np.full((1, 3), 2.3841858e-07, np.float32)
    ls = np.zeros((1, 3), np.float32)
    done, t0 = 0, time.time()
    while True:
        c = np.stack([rng.randint(0, H, rows), rng.randint(0, H, rows)], 1)
        emb = oracle.embed(c, angles, periods, oracle.SEED0_FREQS, (H, H))
        raw, cache = oracle.mlp_forward(P, emb, K)
        pr = oracle.sigmoid(raw)
        _, dpred, _, _ = oracle.img2mse_grads(pr, img[c[:, 0], c[:, 1]], la, ls)
        G = oracle.mlp_backward(P, cache, dpred * pr * (1 - pr))
        oracle.adam_step(P, G, st, 5e-4)
        done += rows
        if time.time() - t0 > seconds_target:
            break
    dt = time.time() - t0
    return {"value": done / dt, "unit": "rows/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{done} rows ({done // rows} train steps of {rows} rows, same image/net) in {dt:.1f}s, "
                      f"NumPy fp32 oracle, BLAS threads = all host cores"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:      # launched by torch.distributed.run (any N >= 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if dist is not None else 0)

    import oracle                      # only for the synthetic workload + cpu_baseline leg
    from npp_amd import ops
    from npp_amd.fit import CompletionFit

    H, K = args.size, args.K
    img, mask = oracle.synthetic_image(H, seed=rank)           # each rank fits its own image
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    P = oracle.init_params(K, seed=rank)
    fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, P, device=dev, N_rand=8192,
                        ksplit=args.ksplit, seed=rank)
    net = fit.net
    patch = oracle.patch_size_from_period(periods[0])          # loaders.py:133-134 -> 96 at 512^2
    n_pix, n_patch = fit.N_rand, 2 * patch * patch             # patch_num = 2 (arg_config.py:63)
    n_rows = n_pix + n_patch
    bp = ops.pad_rows(n_rows)

    # ---- synthetic inputs, resident in HBM before timing: per-step coordinate batches ----
    rng = np.random.RandomState(1000 + rank)
    n_batches = 8
    batches = []
    for _ in range(n_batches):
        sel = rng.choice(fit.i_train.shape[0], n_pix, replace=False)
        pix = fit.i_train[sel]
        cy = rng.randint(patch // 2 + 1, H - patch // 2 - 1, 2)
        cx = rng.randint(patch // 2 + 1, H - patch // 2 - 1, 2)
        pc = []
        for y0, x0 in zip(cy, cx):                             # sampler.py:269-279 patch coordinate grids
            yy, xx = np.meshgrid(np.arange(y0 - patch // 2, y0 + patch // 2),
                                 np.arange(x0 - patch // 2, x0 + patch // 2), indexing="ij")
            pc.append(np.stack([yy, xx], -1).reshape(-1, 2))
        allc = np.concatenate([pix] + pc + [np.zeros((bp - n_rows, 2), np.int64)], 0).astype(np.int32)
        c = torch.from_numpy(allc).to(dev)
        gt = fit.img[c[:n_rows, 0].long(), c[:n_rows, 1].long()].contiguous()
        batches.append((c, gt))
    ws = net.workspace(bp)
    ws["dpred"].zero_()

    def step(i):
        c, gt = batches[i % n_batches]
        net.zero_grad()
        net.forward_train(c)
        # Round 1: the patch rows carry the adaptive robust loss against the image as well
        # (the contextual / LPIPS kernels are not in the loop yet); every row
        # goes through exactly the forward / backward / wgrad work of the reference step.
        net.pixel_loss(bp, n_rows, gt)
        net.backward(bp)
        net.optimizer_step(bp)

    graphs = None
    if args.graph:
        graphs = [net.capture_step(c, n_rows, gt) for c, gt in batches]
        eager_step = step

        def step(i):                      # noqa: F811  -- graph replay of the same iteration
            net.replay_step(graphs[i % n_batches])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_rows * args.steps / dt

    # ---- per-kernel device time (HIP events on the launch stream), same region ----------
    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    c0, gt0 = batches[0]
    kt = {
        "mlp_fwd_train": timed(lambda: net.forward_train(c0)),
        "mlp_bwd_chain": timed(lambda: ops.mlp_bwd(ws["dpred"], ws["pred"], K, net.wb, net.params, ws["actT"], ws["dzT"])),
        "mlp_wgrad": timed(lambda: ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, K, net.ksplit, ws["gslabs"])),
        "pixel_loss": timed(lambda: net.pixel_loss(bp, n_rows, gt0)),
        "adam+repack": timed(lambda: (ops.adam_step(net.params, net.m, net.v, ws["gslabs"], net.ksplit, net.n_params, 0.0, 1),
                                      net.repack())),
        "render_fwd_512sq": timed(lambda: net.render(fit.i_all_dev), reps=5),
    }
    fwd_macs, train_macs = oracle.mlp_macs_per_pixel(K)
    flops = {"mlp_fwd_train": 2 * fwd_macs * n_rows,
             "mlp_bwd_chain": 2 * (train_macs - 2 * fwd_macs) * n_rows,   # dgrad = fwd - embedding part
             "mlp_wgrad": 2 * fwd_macs * n_rows}
    # Algorithmic HBM bytes per launch: 16-bit stash fragments, every array once.
    E, W2 = 462, 256
    emb_cols = K * 480
    z_cols = 10 * W2 + W2 // 2                      # fp16 z of L0..L7, S, P
    lin_cols = 2 * W2                               # bf16 f1, f2
    dz_cols = 11 * W2 + W2 // 2 + 32               # dz of every layer (+ the padded rgb rows)
    wjob_rows = (W2 + 480) * 2 + (W2 + W2) * 10 + (W2 + 480) * (K - 1) + (W2 // 2 + W2) * 2 + (32 + W2 // 2)
    n_par = net.n_params
    hbm_bytes = {"mlp_fwd_train": bp * (8 + 12 + 2 * (z_cols + lin_cols + emb_cols)) + 2.4e6,
                 "mlp_bwd_chain": bp * (24 + 2 * z_cols + 2 * dz_cols) + 1.5e6,
                 "mlp_wgrad": bp * 2 * wjob_rows + 4 * n_par * args.ksplit}
    dom = max(flops, key=lambda k: kt[k])
    tf = {k: flops[k] / kt[k] / 1e12 for k in flops}
    gbs = {k: hbm_bytes[k] / kt[k] / 1e9 for k in flops}
    # the roof that binds the dominant kernel is the one it sits closer to
    hbm_bound = gbs[dom] / PEAK_HBM_GBS > tf[dom] / PEAK_BF16_TFLOPS
    measured = None
    try:     # PMC-measured HBM traffic of the same kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
             # passes; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md), committed under profiles/
        import glob
        pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_summary.json")))[-1]))
        key = {"mlp_fwd_train": "npp::mlp_fwd_kernel<true, true>", "mlp_bwd_chain": "npp::mlp_bwd_kernel<true>",
               "mlp_wgrad": "npp::wgrad_kernel"}[dom]
        measured = pm["kernels"][key]["hbm_bytes"]
    except Exception:
        pass
    roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom,
                "achieved": gbs[dom] if hbm_bound else tf[dom], "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": gbs[dom] / PEAK_HBM_GBS if hbm_bound else tf[dom] / PEAK_BF16_TFLOPS,
                "traffic": measured, "algorithmic_bytes_per_launch": hbm_bytes[dom],
                "algorithmic_flops_per_launch": flops[dom], "avg_launch_us": kt[dom] * 1e6,
                "all_kernels_us": {k: round(v * 1e6, 2) for k, v in kt.items()},
                "all_kernels_tflops": {k: round(v, 1) for k, v in tf.items()},
                "all_kernels_hbm_GBs": {k: round(v, 0) for k, v in gbs.items()},
                "all_kernels_mfma_frac": {k: round(v / PEAK_BF16_TFLOPS, 4) for k, v in tf.items()},
                "step_mfma_frac": 2 * train_macs * n_rows / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
    render_px_s = H * H / kt["render_fwd_512sq"]

    # ---- c4: stand-alone embedder on the full 1024^2 grid, fp32 (HBM-write-bound kernel K1) ----
    c4 = None
    if rank == 0 and not args.no_extras:
        yy, xx = np.meshgrid(np.arange(1024, dtype=np.int32), np.arange(1024, dtype=np.int32), indexing="ij")
        grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
        a4, p4, _ = oracle.synthetic_periodicity(1024, 3)
        from npp_amd import EmbedCfg
        cfg4 = EmbedCfg.make(a4, p4, oracle.SEED0_FREQS, (1024, 1024))
        out = {}
        for name, dt, prec, bpe in (("fp32_precise", torch.float32, True, 4), ("fp32", torch.float32, False, 4), ("bf16", torch.bfloat16, False, 2)):
            t_emb = timed(lambda: ops.embed_fwd(grid, cfg4, dt, precise=prec), reps=5)
            nbytes = grid.shape[0] * (8 + bpe * 3 * 462)
            out[name] = {"ms": t_emb * 1e3, "pixels_per_s": grid.shape[0] / t_emb, "GB_per_s": nbytes / t_emb / 1e9,
                         "frac_of_hbm_peak": nbytes / t_emb / 1e9 / PEAK_HBM_GBS}
        net4 = CompletionFit(*oracle.synthetic_image(64), a4, p4, oracle.SEED0_FREQS, oracle.init_params(3, seed=0), device=dev).net
        net4.cfg = cfg4
        t_r = timed(lambda: net4.render(grid), reps=3)
        out["render_1024sq_bf16"] = {"ms": t_r * 1e3, "pixels_per_s": grid.shape[0] / t_r,
                                     "TFLOP_per_s": 2 * fwd_macs * grid.shape[0] / t_r / 1e12}
        c4 = out
        del grid

    # ---- the complete loop body of train.py:133-264 (patch sampler inputs pre-drawn, VGG trunks via
    #      PyTorch/MIOpen glue with fixed-seed weights, CX core / LPIPS head / patch gather in HIP) ----
    full_loop = None
    if rank == 0 and not args.no_extras:
        _, _, shifts = oracle.synthetic_periodicity(H, K)
        f3 = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), device=dev,
                           N_rand=8192, ksplit=args.ksplit, seed=0, shifts=shifts)
        pre = []
        while len(pre) < 10:
            b = f3.sample_batch()
            if b is not None:
                pre.append(b)
        for b in pre:
            f3.step_from(b)
        per = {}
        for src in ("val", "train", "same"):
            bs = [b for b in pre if b["source"] == src]
            if bs:
                per[src] = timed(lambda: [f3.step_from(b) for b in bs], reps=5) / len(bs) * 1e3
        mix = 0.5 * per.get("val", 0) + 0.3 * per.get("train", per.get("val", 0)) + 0.2 * per.get("same", per.get("val", 0))
        full_loop = {"ms_per_iter_by_patch_source": per, "ms_per_iter_mix_50_30_20": mix,
                     "rows_per_s": (n_pix + 2 * f3.patch_size ** 2) / (mix * 1e-3),
                     "note": "device side only (sampling pre-drawn); VGG19/VGG16 trunks run through PyTorch/MIOpen"}

    # ---- iterations to 28 dB on a fresh fit of the same image (not timed) -----------------
    iters_to_target, final_psnr = None, None
    if not args.no_psnr and rank == 0:
        f2 = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), device=dev,
                           N_rand=8192, ksplit=args.ksplit, seed=0)
        for it in range(1, 301):
            f2.step()
            if iters_to_target is None and it % 5 == 0 and f2.psnr() >= 28.0:
                iters_to_target = it
        final_psnr = f2.psnr()

    # ---- the one collective of the job: gather the fitted images -------------------------
    gather_ms = None
    if dist is not None:
        out = fit.render_image().contiguous()
        bufs = [torch.empty_like(out) for _ in range(world)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_gather(bufs, out)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(K, H)

    if rank == 0:
        line = {
            "metric": "fitted pixels/sec/GPU (512^2 grid, 256-wide MLP) + iters-to-target-PSNR",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"c2: {H}x{H} completion image, top-{K} proposals, 256-wide x 8-layer NPP_Net; "
                                   f"step = 1 optimisation iteration over {n_pix} pixel rows + 2x{patch}^2 patch rows "
                                   f"(fused embed+MLP fwd, adaptive robust loss, bwd chain, grouped wgrad, Adam)",
                       "rows_per_step": n_rows, "image": [H, H], "K": K, "width": 256, "ksplit": args.ksplit,
                       "images_per_gpu": 1, "hip_graph": bool(args.graph)},
            "value_per_gpu": value / world,
            "render_pixels_per_s_per_gpu": render_px_s,
            "iters_to_28dB": iters_to_target, "psnr_known_after_300_iters": final_psnr,
            "final_gather_ms": gather_ms,
            "c4_embedder_1024sq": c4, "full_loop_with_patch_losses": full_loop,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
