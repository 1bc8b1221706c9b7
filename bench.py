#!/usr/bin/env python3
"""bench.py -- fitted pixels/s of the NPP-Net optimisation step on MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "c2"): one 512x512 completion image, top-3
periodicity proposals, 256-wide x 8-layer NPP_Net, bf16 MFMA operands / fp32 accumulate.
A step = ONE COMPLETE optimisation iteration of NPP_completion/train.py:133-264 over
N_rand = 8192 pixel rows + patch_num * P^2 = 2 * 96^2 patch rows (P from loaders.py:133-134):
fused embedder + MLP forward (with stashes) -> adaptive robust pixel loss -> patch plumbing ->
VGG19[0:18] trunk on the 2 * n_p * k patches -> contextual loss -> trunk data-gradient
(-> VGG16 trunk + LPIPS head and back when the patch source is 'same', ~20 % of iterations)
-> MLP backward chain -> grouped wgrad -> Adam (+ weight re-pack).  Every kernel of the step is
libnpp_hip.so.  `value` is the loop THAT SAMPLES (round 6): every timed step is CompletionFit.step_full() on the
reference's own random stream (models/sampler.py:297-354, train.py:164-181: the native MT19937 generator on a
producer thread, the sampler's device launches one iteration ahead on their own stream), K = --steps iterations
between barrier + synchronize, repeated for WINDOWS consecutive windows; the line reports the MEDIAN window
(`ms_per_step` x `steps` = that window; min / max of the windows in `config`).  The image, mask and lattice are
resident in HBM; nothing is cached between steps.  `config.device_only_rows_per_s` keeps the round-2..5 definition
(sampler outputs of 40 iterations pre-drawn, timed steps cycle through that pool with the exact 50 / 30 / 20 source
mix).  Rows fitted per second, summed over ranks (each rank fits its own image: weak scaling, no data-path
collective; one all_gather of the fitted images after the loop).  `mlp_only_step` is the same iteration without
the patch losses (the round-1 line).
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
PEAK_FP8_TFLOPS = 5000.0      # dense fp8 (block-scaled 32x32x64 / 16x16x128 forms), same table: what the 8-bit weight-gradient launch runs on
WINDOWS = 5                   # consecutive --steps windows of the headline loop; the line reports their median
PEAK_HBM_GBS = 8000.0
# the sampler's 50 / 30 / 20 source mix (sampler.py:297-354) as a period-10 pattern: any 10 consecutive pool entries hold 5 / 3 / 2
POOL_PATTERN = ("val", "train", "val", "same", "val", "train", "val", "same", "train", "val")


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
    ap.add_argument("--ksplit", type=int, default=0, help="split-K of the weight-gradient launch; 0 = auto (CUs / tiles)")
    ap.add_argument("--windows", type=int, default=WINDOWS, help="consecutive --steps windows of the sampling loop whose median is `value` "
                    "(0: profiling runs -- only the pre-drawn-pool loop is run and reported, marked as such)")
    ap.add_argument("--pool", type=int, default=40, help="pre-drawn sampler outputs the timed steps cycle through (a multiple of 10: the "
                    "sampler's 50 / 30 / 20 % source mix is then held exactly)")
    return ap.parse_args()


def _dig(d, *keys):
    """d[k0][k1]... or None when any level is missing (extras are optional)."""
    for k_ in keys:
        if not isinstance(d, dict) or k_ not in d:
            return None
        d = d[k_]
    return d


def _e2e_rate(e2e, key, n_rows):
    ms = _dig(e2e, key, "ms_per_iter")
    return (n_rows / (ms * 1e-3)) if ms else None


def host_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box of this pool
    shows 256 logical CPUs but grants 16 of them per GPU; 256 threads on that share run 200x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(K, H, patch, n_pix, n_p, k, seconds_target=20.0):
    """The CPU restatement of the same workload timed on this host's cores, on a bounded sample:
    leg A -- the MLP half (embed + NPP_Net forward + robust pixel loss + backward + Adam) of WHOLE iterations
    (all n_pix + n_p * patch^2 rows at once) in PyTorch-CPU fp32 with autograd, torch.set_num_threads(all cores)
    (oracle/npp_torch_oracle.py, pinned to the NumPy oracle by tests/test_oracle_torch.py);
    leg B -- the patch-loss half per iteration in PyTorch-CPU as well (VGG19[0:18] trunk on the 2 * n_p * k patches by
    F.conv2d, the contextual loss as torch ops, autograd back to the prediction patches; every 5th iteration also the
    VGG16 trunk + LPIPS head on 2 * n_p patches, the 20 % 'same' iterations).
    value = rows of one iteration / (leg A time + leg B time per iteration)."""
    import oracle
    from oracle import npp_torch_oracle as T
    cores = host_cores()
    torch.set_num_threads(cores)
    try:                                   # the NumPy oracle's SGEMMs: same thread budget
        from threadpoolctl import threadpool_limits
        blas_limit = threadpool_limits(limits=cores)
    except ImportError:
        blas_limit = None
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    img, mask = oracle.synthetic_image(H)
    rng = np.random.RandomState(0)
    total = n_pix + n_p * patch * patch
    la = np.full((1, 3), 2.3841858e-07, np.float32)
    ls = np.zeros((1, 3), np.float32)
    # ---- leg A: MLP half, torch, whole batch
    Pt = T.params_t(oracle.init_params(K, seed=0))
    opt = torch.optim.Adam(list(Pt.values()), lr=5e-4, betas=(0.9, 0.999))
    c = np.stack([rng.randint(0, H, total), rng.randint(0, H, total)], 1)
    ct, gt = torch.from_numpy(c), img[c[:, 0], c[:, 1]]
    T.train_step_t(Pt, opt, ct, gt, angles, periods, oracle.SEED0_FREQS, (H, H), K, la, ls)        # warm-up
    n_a, t0 = 0, time.time()
    while time.time() - t0 < seconds_target * 0.5:
        T.train_step_t(Pt, opt, ct, gt, angles, periods, oracle.SEED0_FREQS, (H, H), K, la, ls)
        n_a += 1
    t_a = (time.time() - t0) / n_a

    # ---- leg B: patch-loss half on PyTorch-CPU like the reference runs it (F.conv2d trunks on the host's oneDNN + autograd, torch
    #      contextual core: oracle/npp_torch_oracle.py, pinned to the NumPy oracle by tests/test_oracle_torch.py); the LPIPS head's
    #      per-channel robust NLL on the five taps stays in the NumPy oracle (elementwise work between two torch trunk passes)
    def weights(cfg):
        ws_, cin = [], 3
        for v in cfg:
            if v != "M":
                ws_.append(((rng.randn(v, cin, 3, 3) * np.sqrt(2.0 / (9 * cin))).astype(np.float32), np.zeros(v, np.float32)))
                cin = v
        return T.trunk_weights_t(ws_)
    w19, w16 = weights(oracle.VGG19_CX_CFG), weights(oracle.VGG16_LPIPS_CFG)
    chns = [64, 128, 256, 512, 512]
    lins = [np.abs(rng.randn(c_)).astype(np.float32) * 0.05 for c_ in chns]
    lat_a = [np.full((1, c_), 2.3841858e-07, np.float32) for c_ in chns]
    lat_s = [np.zeros((1, c_), np.float32) for c_ in chns]

    def head(f0, f1):
        loss_, dfs_, _, _ = oracle.lpips_head_grads(f0, f1, lins, lat_a, lat_s)
        return loss_, dfs_
    nk = n_p * k
    xy = torch.from_numpy(rng.rand(2 * nk, 3, patch, patch).astype(np.float32))
    T.contextual_step_t(xy, nk, oracle.VGG19_CX_CFG, w19, oracle.VGG19_CX_TAPS)                   # warm-up (oneDNN primitives)
    n_b, t1 = 0, time.time()
    while True:
        T.contextual_step_t(xy, nk, oracle.VGG19_CX_CFG, w19, oracle.VGG19_CX_TAPS)
        if n_b % 5 == 2:                                                                              # the 20 % 'same' iterations
            T.lpips_step_t(xy[:2 * n_p], n_p, oracle.VGG16_LPIPS_CFG, w16, oracle.VGG16_LPIPS_TAPS, head)
        n_b += 1
        if time.time() - t1 > seconds_target * 0.5 and n_b >= 5:
            break
    t_b = (time.time() - t1) / n_b
    if blas_limit is not None:
        blas_limit.restore_original_limits()
    return {"value": total / (t_a + t_b), "unit": "rows/s", "cores": cores, "kind": "port",
            "mlp_half_rows_per_s": total / t_a, "mlp_half_s_per_iteration": t_a, "patch_half_s_per_iteration": t_b,
            "sample": f"{n_a} MLP-half steps of {total} rows (PyTorch-CPU fp32 autograd, {cores} threads = this process's CPU share of {os.cpu_count()} logical CPUs, whole batch) in {n_a * t_a:.1f}s + "
                      f"{n_b} patch-loss halves (PyTorch-CPU F.conv2d + autograd: VGG19 trunk + contextual loss on {2 * n_p * k} {patch}x{patch} patches, "
                      f"VGG16 + LPIPS head every 5th) in {n_b * t_b:.1f}s"}


def dry_run_dist(args, rank, world):
    """NPP_BENCH_DRYRUN=1: the launch / rendezvous / gather skeleton of the N-rank run without a GPU (gloo), for the CPU
    test of the `--gpus N` spawn path (tests/test_bench_spawn.py).  No kernel runs and nothing is measured."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    assert dist.get_world_size() == args.gpus == world
    mine = torch.full((4, 4, 3), float(rank))
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": "gloo", "collective_ranks": dist.get_world_size(),
                          "gathered": [float(b[0, 0, 0]) for b in bufs], "max_over_ranks": float(t.item())}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torch.distributed.run, rendezvous on
        # 127.0.0.1) BEFORE anything in this process touches the GPU, and exit with their code.
        # torch.cuda.device_count() does not initialise HIP on this image.
        from npp_amd.parallel import launch_ranks
        dry = os.environ.get("NPP_BENCH_DRYRUN") == "1"
        shared_card = "NPP_BENCH_DEVICE" in os.environ           # rehearsal: all ranks on one card, gloo (see below)
        n_dev = torch.cuda.device_count()
        if not dry and not shared_card and n_dev < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but this node exposes {n_dev} GPU(s); refusing to report an "
                     f"N={args.gpus} line from fewer devices")
        sys.exit(launch_ranks(os.path.abspath(__file__), args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or run `python bench.py --gpus {args.gpus}`, which starts the ranks itself)")
    if os.environ.get("NPP_BENCH_DRYRUN") == "1":
        return dry_run_dist(args, rank, world)
    dist = None
    backend = None
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:      # launched by torch.distributed.run (any N >= 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # Rehearsal knobs (one-GPU box): NPP_BENCH_BACKEND=gloo NPP_BENCH_DEVICE=0 runs N ranks on one card without RCCL.
        backend = os.environ.get("NPP_BENCH_BACKEND", "nccl")
        if "NPP_BENCH_DEVICE" in os.environ:
            local = int(os.environ["NPP_BENCH_DEVICE"])
        elif local >= torch.cuda.device_count():
            sys.exit(f"bench.py: rank {rank} has LOCAL_RANK {local} but the node exposes {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if dist is not None else 0)
    # Rehearsal of the 8-rank node's HOST side on a smaller box: NPP_BENCH_CPUS_PER_RANK=c confines this rank (its enqueueing
    # thread, its sampler's producer thread, the native generator) to c logical CPUs of its own, like cpu_count / 8 on the node.
    if os.environ.get("NPP_BENCH_CPUS_PER_RANK"):
        ncpu = int(os.environ["NPP_BENCH_CPUS_PER_RANK"])
        avail = sorted(os.sched_getaffinity(0))
        mine = [avail[(rank * ncpu + i) % len(avail)] for i in range(ncpu)]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, ncpu))

    from npp_amd import ops, synthetic as syn     # (oracle/ is imported inside cpu_baseline() only)
    from npp_amd.io import patch_size_from_period
    from npp_amd.fit import CompletionFit

    H, K = args.size, args.K
    img, mask = syn.synthetic_image(H, seed=rank)           # each rank fits its own image
    angles, periods, shifts = syn.synthetic_periodicity(H, K)
    P = syn.init_params(K, seed=rank)
    # Own image and own initial weights per rank, but the SAME sampler stream (seed 0) everywhere: the patch-source mix of
    # the pool ('same' iterations cost 1.5x a 'val' one) is then identical on every GPU, i.e. per-GPU work is fixed as N
    # grows (weak scaling) instead of the slowest random mix setting the max-over-ranks time.
    fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, P, device=dev, N_rand=8192,
                        ksplit=args.ksplit, seed=0, shifts=shifts)
    if world > 1:
        args.no_extras = True          # the extras (c4 table, end-to-end, ranking, throughput mode) are N = 1 reports
    net = fit.net
    patch = fit.patch_size                                     # loaders.py:133-134 -> 96 at 512^2
    assert patch == patch_size_from_period(periods[0])
    n_pix, n_patch = fit.N_rand, fit.patch_num * patch * patch # patch_num = 2 (arg_config.py:63)
    n_rows = n_pix + n_patch
    bp = ops.pad_rows(n_rows)

    # ---- synthetic inputs, resident in HBM before timing: the sampler's output for `pool` iterations
    #      (train.py:152-181: sample_patches -> pixel draw), in the reference's RNG order ----
    # The pool holds the sampler's expected patch-source mix EXACTLY (50 / 30 / 20 % val / train / same, sampler.py:297-354) AND
    # in an order in which EVERY window of 10 consecutive entries (cyclically) holds 5 val / 3 train / 2 same: draws arrive in the
    # reference's RNG order, fill per-source quotas and are then dealt into the period-10 pattern below -- so any --steps that is a
    # multiple of 10 (the driver's 20 included) times the sampler's own mix ('same' iterations cost ~1.2x a 'val' one).
    quota = {"val": args.pool // 2, "train": (args.pool * 3) // 10, "same": 0}
    quota["same"] = args.pool - quota["val"] - quota["train"]
    by_src = {"val": [], "train": [], "same": []}
    while sum(len(v_) for v_ in by_src.values()) < args.pool:
        b = fit.sample_batch()
        if b is not None and len(by_src[b["source"]]) < quota[b["source"]]:   # k == 0 -> the reference skips the iteration (train.py:160-161)
            assert b["n"] == n_rows and b["bp"] == bp
            by_src[b["source"]].append(b)
    pool = []
    if args.pool % 10 == 0:
        taken = {s_: 0 for s_ in by_src}
        for i in range(args.pool):
            s_ = POOL_PATTERN[i % 10]
            pool.append(by_src[s_][taken[s_]])
            taken[s_] += 1
    else:
        pool = by_src["val"] + by_src["train"] + by_src["same"]
    mix = {s_: sum(b["source"] == s_ for b in pool) for s_ in ("val", "train", "same")}
    ws = net.workspace(bp)

    def step(i):
        fit.step_from(pool[i % len(pool)])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, len(pool))):               # every pool entry at least once (allocations, first-call setup)
        step(i)
    barrier()
    # (no cyclic garbage collection inside the timed steps: a generation-2 pass over a process that holds torch + the pool is a
    #  multi-millisecond host stall, visible in a 20-step region of ~12 ms)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [n_rows * args.steps / float(x.item()) for x in allt]       # rows/s of every rank's own loop
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dev_only_ms_per_step = dt / args.steps * 1e3
    dev_only_value = world * n_rows * args.steps / dt
    dev_only_per_rank = per_rank
    mix_timed_dev = {s_: sum(pool[i % len(pool)]["source"] == s_ for i in range(args.steps)) for s_ in ("val", "train", "same")}

    # ---- THE HEADLINE: the loop that samples.  Own image / weights per rank, the SAME reference stream (seed 0)
    #      on every rank so that the per-window patch-source mix -- 'same' iterations cost ~1.2x a 'val' one -- is identical on
    #      every GPU (weak scaling: per-GPU work fixed as N grows).  WINDOWS windows of exactly --steps iterations, each between
    #      barrier + synchronize on both sides, max over ranks per window; reported: the median window. ----
    NW = max(0, int(args.windows))
    fe = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=rank), device=dev, N_rand=8192,
                       ksplit=args.ksplit, seed=0, shifts=shifts, rng_mode="reference", prefetch=8)
    for _ in range(max(args.warmup, 20) if NW else 0):
        fe.step_full()
    barrier()
    gc.collect()
    gc.disable()
    win_dt, win_ok, win_mix = [], [], []
    for w_ in range(NW):
        n_ok, mix_w = 0, {"val": 0, "train": 0, "same": 0}
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if fe.step_full():                                   # False: the reference `continue`s (no valid real patch), nothing fitted
                n_ok += 1
                mix_w[fe.last_draw["source"]] += 1
        barrier()
        dtw = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dtw], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtw = float(t.item())
        win_dt.append(dtw)
        win_ok.append(n_ok)
        win_mix.append(mix_w)
    gc.enable()
    if NW == 0:                                  # --windows 0 (profiling): the line reports the pre-drawn-pool loop and says so
        win_dt, win_ok, win_mix = [dev_only_ms_per_step * args.steps * 1e-3], [args.steps], [dict(mix_timed_dev)]
        NW = 1
    order = sorted(range(NW), key=lambda i_: win_dt[i_] / max(win_ok[i_], 1))
    med = order[NW // 2]
    dt = win_dt[med]
    ms_per_step = dt / args.steps * 1e3
    value = world * n_rows * win_ok[med] / dt
    win_rates = [world * n_rows * win_ok[i_] / win_dt[i_] for i_ in range(NW)]
    per_rank = None
    if dist is not None:
        t = torch.tensor([n_rows * win_ok[med] / dt], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]               # (every rank's window is bounded by the same barriers)
    fe.close()
    del fe
    # the timed steps walk the pool in order: a step count that is not a multiple of the pool weighs the sources by the pool's
    # ORDER instead of its 50 / 30 / 20 mix ('same' iterations cost 1.3x): say so in the line instead of hiding it
    mix_timed = {s_: sum(pool[i % len(pool)]["source"] == s_ for i in range(args.steps)) for s_ in ("val", "train", "same")}   # (of the device-only loop)
    mix_ok = (mix_timed["val"] * 10 == args.steps * 5 and mix_timed["train"] * 10 == args.steps * 3 and mix_timed["same"] * 10 == args.steps * 2)
    if not mix_ok and rank == 0:
        print(f"bench.py: --steps {args.steps} is not a multiple of 10 (or --pool is not): timed source mix {mix_timed} is not 50/30/20",
              file=sys.stderr)

    # ---- host time to ENQUEUE one iteration (34 C-ABI calls through ctypes), measured on a drained queue over 8 iterations (well
    #      inside the HIP queue's depth, so no launch call blocks on the device): the loop is device-bound while this stays below
    #      ms_per_step -- the number to watch when 8 ranks share a host ----
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for i in range(8):
        step(i)
    host_enqueue_ms = (time.perf_counter() - t_h) / 8 * 1e3
    torch.cuda.synchronize()
    # ---- the same loop INCLUDING each rank's host-side sampling (reference random stream from the native generator on a producer
    #      thread + the sampler's device launches), on every rank at once: N producer threads + N enqueueing threads on one host ----
    e2e_ranks = None
    if dist is not None and world > 1:
        te = torch.tensor([host_enqueue_ms], dtype=torch.float64, device=dev)
        alle = [torch.empty_like(te) for _ in range(world)]
        dist.all_gather(alle, te)
        e2e_ranks = {"host_enqueue_ms_per_iter": [float(x[0]) for x in alle]}

    # ---- the one collective of the job: gather the fitted images -- directly behind the timed loop, before any rank-0-only
    #      extra (the other ranks would sit in the all_gather meanwhile) ----
    gather_ms = None
    if dist is not None:
        out = fit.render_image().contiguous()
        bufs = [torch.empty_like(out) for _ in range(world)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_gather(bufs, out)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del bufs
    if rank != 0:                          # everything below is rank 0's report
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- extra: the same iteration without the patch losses (round-1 definition of the step) ----
    batches = [(b["coords"], fit.masked_img[b["coords"][:n_rows, 0].long(), b["coords"][:n_rows, 1].long()].contiguous()) for b in pool[:8]]
    n_batches = len(batches)
    ws["dpred"].zero_()

    def mlp_step(i):
        c, gt = batches[i % n_batches]
        net.zero_grad()
        net.forward_train(c)
        net.pixel_loss(bp, n_rows, gt)                          # every row carries the pixel loss here
        net.backward(bp)
        net.optimizer_step(bp)

    for i in range(10):
        mlp_step(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(100):
        mlp_step(i)
    torch.cuda.synchronize()
    mlp_ms = (time.perf_counter() - t1) / 100 * 1e3
    ws["dpred"].zero_()
    ws["n_rows"] = None

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
    nk = fit.patch_num * fit.topk
    xy = torch.rand((2 * nk, 3, patch, patch), device=dev)
    cxl, lpl = fit.contextualLoss, fit.percepLoss
    sc19, sh19 = [1.0 / s_ for s_ in cxl._STD], [-m_ / s_ for m_, s_ in zip(cxl._MEAN, cxl._STD)]
    f19 = cxl.hip_trunk._forward(xy, sc19, sh19)[0]
    _, dfx = ops.cx_fwd_bwd(f19[:nk], f19[nk:], 0.5, None, 1e-3, None, True)
    lbuf = torch.zeros(1, device=dev)
    patch_kt = {
        "vgg19_fwd_12img": timed(lambda: cxl.hip_trunk._forward(xy, sc19, sh19)),
        "cx_core_fwd_bwd": timed(lambda: ops.cx_fwd_bwd(f19[:nk], f19[nk:], 0.5, None, 1e-3, lbuf, True)),
        "vgg19_dgrad_6img": timed(lambda: cxl.hip_trunk._backward([dfx], nk, sc19, tuple(xy.shape), zero_rest=False)),
        "contextual_total": timed(lambda: cxl.fused(xy, nk, 1e-3, lbuf)),
        "lpips_total_4img": timed(lambda: lpl.fused(xy[:4 * fit.patch_num // 2].contiguous(), fit.patch_num, 1e-3, lbuf)),
    }
    lpl.zero_latent_grads()
    kt = {
        "mlp_fwd_train": timed(lambda: net.forward_train(c0)),
        "mlp_bwd_chain": timed(lambda: ops.mlp_bwd(ws["dpred"], ws["pred"], K, net.wb, net.params, ws["actT"], ws["dzT"])),
        "mlp_wgrad": timed(lambda: ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, K, net.ksplit, ws["gslabs"])),
        "pixel_loss": timed(lambda: net.pixel_loss(bp, n_rows, gt0)),
        "adam+repack": timed(lambda: (ops.adam_step(net.params, net.m, net.v, ws["gslabs"], net.ksplit, ws["gslabs"].numel() // net.ksplit, 0.0, 1),
                                      net.repack())),
        "render_fwd_512sq": timed(lambda: net.render(fit.i_all_dev), reps=5),
    }
    # ---- the same kernels IN SEQUENCE: HIP events between the launches of complete MLP-only steps (median over 30 steps).
    #      A kernel timed back-to-back with itself has its operands hot in L2 / the Infinity Cache; inside the step the stash it
    #      reads was written one or two launches (hundreds of MB) earlier.  `roofline` below uses THESE durations. ----
    def seq_times(reps=30):
        names = ["mlp_fwd_train", "pixel_loss", "mlp_bwd_chain", "mlp_wgrad", "adam+repack"]
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(reps)]
        for r_ in range(reps + 3):
            e = evs[r_ - 3] if r_ >= 3 else None
            net.zero_grad()
            if e: e[0].record()
            net.forward_train(c0)
            if e: e[1].record()
            net.pixel_loss(bp, n_rows, gt0)
            if e: e[2].record()
            ops.mlp_bwd(ws["dpred"], ws["pred"], K, net.wb, net.params, ws["actT"], ws["dzT"])
            if e: e[3].record()
            ops.mlp_wgrad(ws["dzT"], ws["actT"], bp, K, net.ksplit, ws["gslabs"])
            if e: e[4].record()
            net.optimizer_step(bp)
            if e: e[5].record()
        torch.cuda.synchronize()
        t = np.array([[e[i].elapsed_time(e[i + 1]) * 1e-3 for i in range(len(names))] for e in evs])
        return dict(zip(names, np.median(t, 0)))
    kt_seq = seq_times()
    ws["dpred"].zero_()
    ws["n_rows"] = None

    # ---- ... and inside the COMPLETE iteration (the timed region itself): HIP events around the three MLP launches of 2 x pool
    #      complete iterations, recorded on the launch stream by wrapping the host layer's calls for the length of this pass ----
    def in_iteration_times():
        names = {"mlp_fwd": "mlp_fwd_train", "mlp_bwd_patch": "mlp_bwd_chain", "mlp_wgrad": "mlp_wgrad"}
        rec = {v: [] for v in names.values()}
        saved = {}

        def wrap(fn_name, key):
            orig = getattr(ops, fn_name)
            saved[fn_name] = orig

            def timed_call(*a_, **k_):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r_ = orig(*a_, **k_)
                e1.record()
                rec[key].append((e0, e1))
                return r_
            setattr(ops, fn_name, timed_call)
        for fn_name, key in names.items():
            wrap(fn_name, key)
        try:
            for i in range(2 * len(pool)):
                step(i)
            torch.cuda.synchronize()
        finally:
            for fn_name, orig in saved.items():
                setattr(ops, fn_name, orig)
        return {k_: float(np.median([a_.elapsed_time(b_) for a_, b_ in v_])) * 1e-3 for k_, v_ in rec.items() if v_}
    kt_iter = in_iteration_times()
    fwd_macs, train_macs = syn.mlp_macs_per_pixel(K)
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
    stash8 = bool(ops.tune("stash8"))      # the 8-bit training stash (default): 2 one-byte arrays per snake layer (fp8 output, u8 snake'),
                                           # fp8 f1 / f2 / embedding slots, bf8 gradients; the weight-gradient launch contracts them on the fp8 MFMA
    if stash8:
        hbm_bytes = {"mlp_fwd_train": bp * (8 + 12 + 2 * z_cols + lin_cols + emb_cols) + 2.4e6,
                     "mlp_bwd_chain": bp * (24 + z_cols + dz_cols) + 1.5e6,
                     "mlp_wgrad": bp * wjob_rows + 4 * n_par * net.ksplit}
    else:
        hbm_bytes = {"mlp_fwd_train": bp * (8 + 12 + 2 * (z_cols + lin_cols + emb_cols)) + 2.4e6,
                     "mlp_bwd_chain": bp * (24 + 2 * z_cols + 2 * dz_cols) + 1.5e6,
                     "mlp_wgrad": bp * 2 * wjob_rows + 4 * n_par * net.ksplit}
    # `roofline` is priced on the launch durations INSIDE the complete iteration (kt_iter: what the timed region runs); the MLP-only
    # step's in-sequence durations (kt_seq, every row a pixel row: the wgrad launch follows the backward chain directly) and the
    # tight-loop ones are printed beside them
    kt_roof = {k: kt_iter.get(k, kt_seq[k]) for k in flops}
    dom = max(flops, key=lambda k: kt_roof[k])
    tf = {k: flops[k] / kt_roof[k] / 1e12 for k in flops}                # in-iteration durations
    tf_seq = {k: flops[k] / kt_seq[k] / 1e12 for k in flops}             # MLP-only step, in sequence
    tf_tight = {k: flops[k] / kt[k] / 1e12 for k in flops}               # back-to-back with itself (flattering: hot operands)
    gbs = {k: hbm_bytes[k] / kt_roof[k] / 1e9 for k in flops}
    # SURVEY.md 8(d) declares the MLP forward / backward / weight-gradient kernels MFMA-bound: `roofline` is the FLOP view
    # (algorithmic FLOPs of 8(d) x rows of one launch / the launch's average duration, against the dense bf16 MFMA peak).
    # The kernels also stream this design's 16-bit activation / gradient stash through HBM (far more than 8(d)'s
    # algorithmic 32 B/row): that byte model and the rate it implies are reported beside it as `design_traffic`.
    measured, mfma_pmc, traffic_source = None, None, None
    pmc_key = ({"mlp_fwd_train": "void npp::mlp_fwd_kernel<2, true, false, false", "mlp_bwd_chain": "void npp::mlp_bwd_kernel<true, true>",
                "mlp_wgrad": "npp::wgrad8_kernel"} if stash8 else
               {"mlp_fwd_train": "void npp::mlp_fwd_kernel<1, true, false, false", "mlp_bwd_chain": "void npp::mlp_bwd_kernel<true, false>",
                "mlp_wgrad": "npp::wgrad_kernel"})          # (prefixes of the profiler's kernel names)
    peak_of = {"mlp_fwd_train": PEAK_BF16_TFLOPS, "mlp_bwd_chain": PEAK_BF16_TFLOPS,
               "mlp_wgrad": PEAK_FP8_TFLOPS if stash8 else PEAK_BF16_TFLOPS}    # the dense peak of the operand type each launch multiplies
    try:     # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE
             # doubled per the gfx950 note in MI355X_MICROARCH.md): the newest summary committed under profiles/
        import glob
        pm_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_hbm_summary.json")))[-1]
        pm = json.load(open(pm_path))
        measured = next((v_["hbm_bytes"] for k_, v_ in pm["kernels"].items() if k_.startswith(pmc_key[dom]) or k_.startswith(pmc_key[dom].replace("void ", ""))), None)
        if measured is None:
            raise KeyError(pmc_key[dom])
        traffic_source = ("profiles/" + os.path.basename(pm_path) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, committed; "
                          "NOT collected in this run)")
    except (OSError, IndexError, KeyError, ValueError):
        measured = None
    try:     # matrix-pipe busy fraction of the same kernel from SQ_VALU_MFMA_BUSY_CYCLES (tools/pmc_sq.sh), same source
        import glob
        pq = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_sq_summary.json")))[-1]))
        mfma_pmc = {k_: next((x_.get("mfma_pipe_busy_frac") for n_, x_ in pq["kernels"].items() if n_.startswith(v_) or n_.startswith(v_.replace("void ", ""))), None)
                    for k_, v_ in pmc_key.items()}
        # the inference render (one 0.7-ms dispatch, long enough for GRBM_GUI_ACTIVE / 8 / wall to be the clock the chip held:
        # MI355X_MICROARCH.md 'DVFS give-back'): pipe-busy fraction AT that clock, next to the FLOP fraction of the 2.4-GHz peak
        rk = next((x_ for n_, x_ in pq["kernels"].items() if "npp::mlp_fwd_kernel<0, true, false, false" in n_ or "npp::mlp_fwd_kernel<false, true, false, false" in n_), None)
        if rk:
            mfma_pmc["render_fwd_512sq"] = rk.get("mfma_pipe_busy_frac")
            mfma_pmc["render_effective_clock_GHz"] = rk.get("effective_clock_GHz")
    except (OSError, IndexError, KeyError, ValueError):
        mfma_pmc = None
    roofline = {"bound": "mfma", "kernel": dom, "achieved": tf[dom], "peak": peak_of[dom], "unit": "TFLOP/s",
                "frac": tf[dom] / peak_of[dom], "traffic": measured, "traffic_source": traffic_source,
                "peak_is": "dense fp8 MFMA (the launch multiplies bf8 x fp8 operands)" if peak_of[dom] == PEAK_FP8_TFLOPS else "dense bf16 MFMA",
                "training_stash": "8-bit (npp_tune stash8: fp8 layer outputs + u8 snake' + bf8 gradients with a per-tile power-of-two scale)" if stash8 else "16-bit",
                "algorithmic_flops_per_launch": flops[dom], "avg_launch_us": kt_roof[dom] * 1e6,
                "timing": "HIP events around the launch inside 2 x pool COMPLETE iterations (the timed region's own launches, cold "
                          "operands), median; all_kernels_us_in_sequence = the same between the launches of MLP-only steps",
                "design_traffic": {"note": "HBM view of the same launch: bytes of this design's stash arrays (every array once "
                                           "per job), NOT SURVEY 8(d)'s algorithmic bytes (32 B/row + weights)",
                                   "bytes_per_launch": hbm_bytes[dom], "GB_per_s": gbs[dom], "frac_of_8TBs": gbs[dom] / PEAK_HBM_GBS,
                                   "survey_8d_algorithmic_bytes_per_step": bp * 32 + 2.4e6 + 28 * n_par},
                # the same launch under the HBM roof: measured bytes (PMC) against its duration, and where that traffic puts the ridge
                "arithmetic_intensity_flop_per_byte": (flops[dom] / measured) if measured else None,
                "ridge_flop_per_byte": peak_of[dom] * 1e12 / (PEAK_HBM_GBS * 1e9),
                "hbm_roof_frac": (measured / kt_roof[dom] / 1e9 / PEAK_HBM_GBS) if measured else None,
                "mfma_frac_ceiling_at_this_traffic": (flops[dom] / (measured / (PEAK_HBM_GBS * 1e9)) / 1e12 / peak_of[dom]) if measured else None,
                "fwd_mfma_frac": tf["mlp_fwd_train"] / PEAK_BF16_TFLOPS, "bwd_mfma_frac": tf["mlp_bwd_chain"] / PEAK_BF16_TFLOPS,
                # the weight-gradient launch against BOTH peaks: the bf16 one BASELINE.json's 40 % target is stated on (the launch does
                # the same algorithmic FLOPs as its bf16 form) and the fp8 one of the operands it actually multiplies in stash8 mode
                "wgrad_mfma_frac": tf["mlp_wgrad"] / PEAK_BF16_TFLOPS, "wgrad_frac_of_its_own_peak": tf["mlp_wgrad"] / peak_of["mlp_wgrad"],
                "trio_us": (kt_roof["mlp_fwd_train"] + kt_roof["mlp_bwd_chain"] + kt_roof["mlp_wgrad"]) * 1e6,
                "trio_frac_of_bf16_peak": sum(flops.values()) / (kt_roof["mlp_fwd_train"] + kt_roof["mlp_bwd_chain"] + kt_roof["mlp_wgrad"]) / 1e12 / PEAK_BF16_TFLOPS,
                "fwd_us": kt_roof["mlp_fwd_train"] * 1e6, "bwd_us": kt_roof["mlp_bwd_chain"] * 1e6, "wgrad_us": kt_roof["mlp_wgrad"] * 1e6,
                "mfma_pipe_busy_frac_pmc": mfma_pmc,
                "stacked_M8_fwd_mfma_frac": None, "stacked_M8_bwd_mfma_frac": None, "stacked_M8_wgrad_mfma_frac": None,
                "all_kernels_us_in_iteration": {k: round(v * 1e6, 2) for k, v in kt_iter.items()},
                "all_kernels_us_in_sequence": {k: round(v * 1e6, 2) for k, v in kt_seq.items()},
                "all_kernels_mfma_frac_mlp_only_step": {k: round(v / PEAK_BF16_TFLOPS, 4) for k, v in tf_seq.items()},
                "all_kernels_us_tight_loop": {k: round(v * 1e6, 2) for k, v in kt.items()},
                "all_kernels_tflops": {k: round(v, 1) for k, v in tf.items()},
                "all_kernels_tflops_tight_loop": {k: round(v, 1) for k, v in tf_tight.items()},
                "all_kernels_design_traffic_GBs": {k: round(v, 0) for k, v in gbs.items()},
                "all_kernels_mfma_frac": {k: round(v / PEAK_BF16_TFLOPS, 4) for k, v in tf.items()},
                "render_mfma_frac": 2 * fwd_macs * H * H / kt["render_fwd_512sq"] / 1e12 / PEAK_BF16_TFLOPS,
                "patch_source_mix_in_timed_steps": mix_timed,
                "mlp_flops_over_full_step_frac": 2 * train_macs * n_rows / (dev_only_ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
    render_px_s = H * H / kt["render_fwd_512sq"]

    # ---- c4: stand-alone embedder on the full 1024^2 grid, fp32 (HBM-write-bound kernel K1) ----
    c4 = None
    if rank == 0 and not args.no_extras:
        yy, xx = np.meshgrid(np.arange(1024, dtype=np.int32), np.arange(1024, dtype=np.int32), indexing="ij")
        grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
        a4, p4, _ = syn.synthetic_periodicity(1024, 3)
        from npp_amd import EmbedCfg
        cfg4 = EmbedCfg.make(a4, p4, syn.SEED0_FREQS, (1024, 1024))
        out = {}
        for name, dt, prec, bpe in (("fp32_precise", torch.float32, True, 4), ("fp32", torch.float32, False, 4), ("bf16", torch.bfloat16, False, 2)):
            t_emb = timed(lambda: ops.embed_fwd(grid, cfg4, dt, precise=prec), reps=5)
            nbytes = grid.shape[0] * (8 + bpe * 3 * 462)
            out[name] = {"ms": t_emb * 1e3, "pixels_per_s": grid.shape[0] / t_emb, "GB_per_s": nbytes / t_emb / 1e9,
                         "frac_of_hbm_peak": nbytes / t_emb / 1e9 / PEAK_HBM_GBS}
        net4 = CompletionFit(*syn.synthetic_image(64), a4, p4, syn.SEED0_FREQS, syn.init_params(3, seed=0), device=dev).net
        net4.cfg = cfg4
        t_r = timed(lambda: net4.render(grid), reps=3)
        out["render_1024sq_bf16"] = {"ms": t_r * 1e3, "pixels_per_s": grid.shape[0] / t_r,
                                     "TFLOP_per_s": 2 * fwd_macs * grid.shape[0] / t_r / 1e12}
        # the same render in exact fp32 (config c4's arithmetic): materialised fp32 table -> generic dense-layer kernels
        # (v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s), in chunks of 131072 rows to bound the 5.5 KB/row table
        from npp_amd.dense import DenseNPPNet
        dn = DenseNPPNet(22, 44, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=256, freq_nerf=21, activation="snake", device=dev)
        dn.load_state_dict({k_: torch.from_numpy(v_) for k_, v_ in syn.init_params(3, seed=0).items()}, strict=False)

        def render_fp32():
            outs = []
            with torch.no_grad():
                for a_ in range(0, grid.shape[0], 131072):
                    emb = ops.embed_fwd(grid[a_:a_ + 131072], cfg4, torch.float32, precise=False)
                    outs.append(torch.sigmoid(dn(None, emb)))
            return torch.cat(outs, 0)
        ref32 = render_fp32()
        t_r32 = timed(render_fp32, reps=2)
        # ... and FUSED in exact fp32 (npp_mlp_fwd32: one launch, v_mfma_f32_32x32x2_f32, nothing materialised): c4's line
        fused32 = net4.render_fp32(grid)
        t_f32 = timed(lambda: net4.render_fp32(grid), reps=3)
        out["render_1024sq_fp32_fused"] = {"dtype": "fp32", "ms": t_f32 * 1e3, "pixels_per_s": grid.shape[0] / t_f32,
                                           "TFLOP_per_s": 2 * fwd_macs * grid.shape[0] / t_f32 / 1e12,
                                           "frac_of_fp32_mfma_peak": 2 * fwd_macs * grid.shape[0] / t_f32 / 1e12 / 157.3,
                                           "max_abs_diff_vs_fp32_dense_chain": float((fused32 - ref32).abs().max())}
        del fused32
        d32 = (net4.render(grid) - ref32).abs()
        out["render_1024sq_fp32_dense"] = {"ms": t_r32 * 1e3, "pixels_per_s": grid.shape[0] / t_r32,
                                           "TFLOP_per_s": 2 * fwd_macs * grid.shape[0] / t_r32 / 1e12,
                                           "frac_of_fp32_mfma_peak": 2 * fwd_macs * grid.shape[0] / t_r32 / 1e12 / 157.3,
                                           "max_abs_diff_vs_bf16_fused": float(d32.max()),
                                           "psnr_vs_bf16_fused_dB": float(-10.0 * torch.log10((d32.double() ** 2).mean()))}
        c4 = out
        del grid, ref32, d32

    # ---- the reference's DEFAULT width (options/arg_config.py:57 --netwidth 512): the same fused chain, second build ----
    w512 = None
    if rank == 0 and not args.no_extras:
        from npp_amd.model import NPPNet
        W5 = 512
        net5 = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), params=syn.init_params(K, seed=0, width=W5), device=dev, width=W5)
        f5, t5 = syn.mlp_macs_per_pixel(K, W5)
        yy5, xx5 = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="ij")
        g5 = torch.from_numpy(np.stack([yy5, xx5], -1).reshape(-1, 2)).to(dev)
        c5 = g5[torch.randint(0, H * H, (n_rows,), device=dev)].contiguous()
        gt5 = torch.rand(n_rows, 3, device=dev)
        net5.workspace(n_rows)["dpred"].zero_()

        def step5():
            net5.zero_grad()
            net5.forward_train(c5)
            net5.pixel_loss(n_rows, n_rows, gt5)
            net5.backward(n_rows)
            net5.optimizer_step(n_rows)
        t_r5 = timed(lambda: net5.render(g5), reps=10)
        t_s5 = timed(step5, reps=30)
        w512 = {"workload": f"{H}x{H} K={K} netwidth 512 (libnpp_hip_w512.so), MLP half only", "ksplit": net5.ksplit,
                "render_ms": t_r5 * 1e3, "render_pixels_per_s": H * H / t_r5, "render_mfma_frac": 2 * f5 * H * H / t_r5 / 1e12 / PEAK_BF16_TFLOPS,
                "mlp_only_step_ms": t_s5 * 1e3, "mlp_only_rows_per_s": n_rows / t_s5,
                "mlp_mfma_frac": 2 * t5 * n_rows / t_s5 / 1e12 / PEAK_BF16_TFLOPS}
        del net5, g5, c5

    # ---- per patch-source cost of the complete iteration (device + host enqueue, same pool) ----
    per_source = None
    if rank == 0 and not args.no_extras:
        per_source = {}
        for src in ("val", "train", "same"):
            bs = [b for b in pool if b["source"] == src]
            if bs:
                per_source[src] = timed(lambda: [fit.step_from(b) for b in bs], reps=5) / len(bs) * 1e3

    # ---- iterations to 28 dB on a fresh fit of the same image (not timed) -----------------
    iters_to_target, final_psnr, e2e = None, None, None
    if not args.no_psnr and rank == 0:
        f2 = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev,
                           N_rand=8192, ksplit=args.ksplit, seed=0, shifts=shifts)
        for it in range(1, 301):                               # the complete loop incl. host-side sampling
            f2.step_full()
            if iters_to_target is None and it % 5 == 0 and f2.psnr() >= 28.0:
                iters_to_target = it
        final_psnr = f2.psnr()
        # wall time per iteration of the complete loop INCLUDING the host-side sampler (not part of `value`, whose inputs
        # are resident before timing): with the reference's exact NumPy stream and with rng_mode='fast'
        e2e = {}
        for mode, pf in ((("numpy", 0), ("reference", 8), ("fast", 8)) if world == 1 else ()):
            f4 = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev,
                               N_rand=8192, ksplit=args.ksplit, seed=0, shifts=shifts, rng_mode=mode, prefetch=pf)
            for _ in range(20):
                f4.step_full()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(200):
                f4.step_full()
            torch.cuda.synchronize()
            e2e[{"numpy": "numpy_stream_serial", "reference": "same_stream_native_rng_producer_thread", "fast": "fast_mode"}[mode]] = {
                "ms_per_iter": (time.perf_counter() - t2) / 200 * 1e3, "psnr_known_dB": f4.psnr()}
            f4.close()

    # ---- one proposal-ranking candidate fit (search.py:85-205: NPP_Net_light, 300 iterations x 2048 rows,
    #      then the LPIPS + contextual score on the pseudo-mask region), wall time incl. host sampling ----
    ranking = None
    if rank == 0 and not args.no_extras:
        from npp_amd.light import ProposalRanker
        pseudo = np.ones((H, H), np.float32)
        pseudo[H // 4:H // 4 + 128, H // 4:H // 4 + 160] = 0
        rk = ProposalRanker(img * mask, np.stack(np.nonzero(pseudo * mask[..., 0]), 1), np.stack(np.nonzero((1 - pseudo) * mask[..., 0]), 1),
                            device=dev, rng_mode="fast")
        rk.fit_candidate(angles[0], periods[0])                 # warm-up (allocations, first-call setup)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        net_l = rk.fit_candidate(angles[0], periods[0])
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t3
        sc = rk.score(net_l)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t3
        # search.py fits every candidate of an image (top-k of the detector: 9 by default) from the same weights on the same pixel
        # rows: the product path (ProposalRanker.fit_candidates / rank) carries all of them in every launch
        cands9 = [(angles[i % K] + 3.0 * (i // K), periods[i % K] * (1.0 + 0.11 * (i // K))) for i in range(9)]
        rk.fit_candidates(cands9)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        nets9 = rk.fit_candidates(cands9)
        torch.cuda.synchronize()
        t_set = time.perf_counter() - t4
        sc9 = [rk.score(n_) for n_ in nets9]
        torch.cuda.synchronize()
        t_set_all = time.perf_counter() - t4
        # the same set on the 16-bit matrix pipe (csrc/npp_light16.hip: bf16 operands, fp32 accumulation / master weights / Adam)
        rk16 = ProposalRanker(img * mask, np.stack(np.nonzero(pseudo * mask[..., 0]), 1), np.stack(np.nonzero((1 - pseudo) * mask[..., 0]), 1),
                              device=dev, rng_mode="fast", precision="bf16")
        rk16.fit_candidates(cands9)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        nets16 = rk16.fit_candidates(cands9)
        torch.cuda.synchronize()
        t_set16 = time.perf_counter() - t5
        sc16 = [rk16.score(n_) for n_ in nets16]
        torch.cuda.synchronize()
        t_set16_all = time.perf_counter() - t5
        set16 = {"n_candidates": len(cands9), "fit_s": t_set16, "fit_s_per_candidate": t_set16 / len(cands9), "fit_plus_score_s": t_set16_all,
                 "ms_per_iteration_of_the_set": t_set16 / rk16.N_iters * 1e3, "rows_per_s": len(cands9) * rk16.N_iters * rk16.N_rand / t_set16,
                 "best_score": min(x[0] for x in sc16), "speedup_vs_fp32_set": t_set / t_set16,
                 "ranking_order_fp32": [int(i) for i in np.argsort([x[0] for x in sc9], kind="stable")],
                 "ranking_order_bf16": [int(i) for i in np.argsort([x[0] for x in sc16], kind="stable")],
                 "how": "4 launches per iteration of the set: fused bf16 forward / data-gradient chains, the main loop's grouped split-K "
                        "weight-gradient kernel over a light job table (candidate = image of a stacked launch), Adam + bf16 re-pack"}
        ranking = {"fit_ms_per_iter": t_fit / rk.N_iters * 1e3, "candidate_fit_s": t_fit, "candidate_fit_plus_score_s": t_all,
                   "rows_per_s": rk.N_iters * rk.N_rand / t_fit, "score": sc[0],
                   "candidate_set": {"n_candidates": len(cands9), "fit_s": t_set, "fit_s_per_candidate": t_set / len(cands9),
                                     "fit_plus_score_s": t_set_all, "ms_per_iteration_of_the_set": t_set / rk.N_iters * 1e3,
                                     "rows_per_s": len(cands9) * rk.N_iters * rk.N_rand / t_set, "best_score": min(x[0] for x in sc9),
                                     "how": "NPPNetLightBatch: fused forward / data-gradient chains (csrc/npp_light.hip) and one grouped weight-gradient launch, the candidate is a grid dimension"},
                   "candidate_set_bf16": set16,
                   "note": "NPP_Net_light D=4 W=256, exact fp32, fused chains; candidate_fit_s is ONE candidate alone (a candidate set of one: "
                           "7 launches per iteration), candidate_set the 9 candidates of an image together (what search.py's loop amounts to)"}

    # ---- extra: throughput mode -- two independent image fits interleaved on this GPU, one stream each (more images than
    #      GPUs, BASELINE config c3 style): their dependent-launch gaps and under-filled kernels overlap ----
    two_fits = None
    if rank == 0 and not args.no_extras:
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        fits2, pools2 = [fit], [pool]
        with torch.cuda.stream(streams[1]):
            im2, mk2 = syn.synthetic_image(H, seed=1000 + rank)
            fb = CompletionFit(im2, mk2, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=1000 + rank), device=dev,
                               N_rand=8192, ksplit=args.ksplit, seed=1000 + rank, shifts=shifts)
            pb = []
            while len(pb) < len(pool):
                b_ = fb.sample_batch()
                if b_ is not None:
                    pb.append(b_)
            for b_ in pb:
                fb.step_from(b_)
        fits2.append(fb)
        pools2.append(pb)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for i in range(100):
            for r_ in range(2):
                with torch.cuda.stream(streams[r_]):
                    fits2[r_].step_from(pools2[r_][i % len(pool)])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t4
        two_fits = {"rows_per_s": 2 * 100 * n_rows / dt2, "ms_per_iteration_each": dt2 / 200 * 1e3,
                    "note": "2 images per GPU, complete iterations interleaved on 2 streams by one host thread (the launch queue stays full: device-bound, tools/host_probe.py)"}

    # ---- extra: BASELINE config c4's TASK -- the remapping variant (NPP_remapping/train.py: whole image trained, clear-region
    #      sampler mask, 0.3-weighted blurry pixels, contextual + Gram style loss) on a 1024^2 image, complete iterations ----
    remap = None
    if rank == 0 and not args.no_extras:
        Hr = 1024
        im_r, _ = syn.synthetic_image(Hr, seed=7)
        a_r, p_r, sh_r = syn.synthetic_periodicity(Hr, K)
        clear = np.ones((Hr, Hr, 1), np.float32)
        clear[Hr // 3:Hr // 2] = 0.0                              # a blurry band (the reference finds it with blur_detection.py)
        remap = {"workload": f"remapping task, {Hr}x{Hr}, K={K}: complete iterations incl. host sampling (contextual + style loss, "
                             f"whole image = {Hr * Hr} known pixels)"}
        for mode, pf in (("reference", 8), ("fast", 8)):
            fr = CompletionFit(im_r, np.ones((Hr, Hr, 1), np.float32), a_r, p_r, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev,
                               N_rand=8192, seed=0, shifts=sh_r, task="remapping", clear_mask=clear, prefetch=pf, rng_mode=mode,
                               contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False)
            for _ in range(20):
                fr.step_full()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(60):
                fr.step_full()
            torch.cuda.synchronize()
            dt5 = (time.perf_counter() - t5) / 60
            rows_r = fr.N_rand + fr.patch_num * fr.patch_size ** 2
            remap[f"rng_{mode}"] = {"ms_per_iter": dt5 * 1e3, "rows_per_iter": rows_r, "rows_per_s": rows_r / dt5}
            fr.close()
            del fr
        remap["note"] = ("rng_reference reproduces np.random.choice(1 048 576, 8192, replace=False) draw for draw: one full permutation of the pixel pool per draw on the host (native MT19937 stream, group-wise rejection walk, AVX2 where available: ~1.2 ms per permutation, two per iteration, on a producer thread); rng_fast draws on the device")

    # ---- extra: M images per GPU in ONE launch sequence (npp_amd.stack.StackedFit: the image is a grid dimension of every launch) --
    #      BASELINE config c3 at N < 8 GPUs (8 / 4 / 2 images per GPU).  device-only: fixed pre-drawn batch sets (each with its own
    #      random mix of patch sources over the images), like `value`; e2e: step_full() incl. every image's host draw ----
    stacked = None
    if rank == 0 and not args.no_extras:
        from npp_amd.stack import StackedFit
        stacked = {"note": "rows/s of ONE GPU fitting M images at once; x_single = against the single-image loop re-timed right before "
                           "these legs (same thermal state), x_headline_value = against `config.device_only_rows_per_s`; e2e legs against fast-mode end-to-end"}
        e2e1 = (e2e or {}).get("fast_mode", {}).get("ms_per_iter")
        # the single-image loop re-timed HERE (the chip is warm by now and holds a lower clock than during the headline loop,
        # which ran first: a ratio against `value` alone would mix the two conditions)
        for i in range(len(pool)):
            step(i)
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        for i in range(2 * len(pool)):
            step(i)
        torch.cuda.synchronize()
        single_now = n_rows * 2 * len(pool) / (time.perf_counter() - t6)
        stacked["single_image_rows_per_s_retimed_here"] = single_now
        for M_ in (2, 4, 8):
          try:
              fs = []
              for i_ in range(M_):
                  im_, mk_ = syn.synthetic_image(H, seed=2000 + i_)
                  fs.append(CompletionFit(im_, mk_, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=2000 + i_), device=dev,
                                          N_rand=8192, seed=2000 + i_, shifts=shifts, rng_mode="fast"))
              st = StackedFit(fs)
              for _ in range(5):
                  st.step_full()
              ts_ = []
              for _ in range(10):
                  bs_ = st.sample()
                  ts_.append(timed(lambda: st.step_from(bs_), reps=20))
              t_dev = float(np.mean(ts_))
              # the three MLP launches of the stacked iteration between HIP events (same wrapping as in_iteration_times): rows of ALL
              # M images per launch against the launch's duration
              rec_ = {"mlp_fwd_stack": [], "mlp_bwd_patch_stack": [], "mlp_wgrad_stack": []}
              saved_ = {}
              for fn_ in rec_:
                  saved_[fn_] = getattr(ops, fn_)

                  def timed_call_(*a_, __o=saved_[fn_], __k=fn_, **k_):
                      e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                      e0_.record()
                      r_ = __o(*a_, **k_)
                      e1_.record()
                      rec_[__k].append((e0_, e1_))
                      return r_
                  setattr(ops, fn_, timed_call_)
              try:
                  for _ in range(20):
                      st.step_from(bs_)
                  torch.cuda.synchronize()
              finally:
                  for fn_, o_ in saved_.items():
                      setattr(ops, fn_, o_)
              kus_ = {k_: float(np.median([a_.elapsed_time(b_) for a_, b_ in v_])) * 1e3 for k_, v_ in rec_.items() if v_}
              n_act_ = sum(1 for b_ in bs_ if b_ is not None)
              kfl_ = {"mlp_fwd_stack": 2 * fwd_macs * n_rows * n_act_, "mlp_bwd_patch_stack": 2 * (train_macs - 2 * fwd_macs) * n_rows * n_act_,
                      "mlp_wgrad_stack": 2 * fwd_macs * n_rows * n_act_}
              kfr_ = {k_: kfl_[k_] / (v_ * 1e-6) / 1e12 / PEAK_BF16_TFLOPS for k_, v_ in kus_.items()}
              torch.cuda.synchronize()
              t6 = time.perf_counter()
              for _ in range(100):
                  st.step_full()
              torch.cuda.synchronize()
              t_e2e = (time.perf_counter() - t6) / 100
              stacked[f"stacked_M{M_}"] = {"ms_per_stacked_iteration": t_dev * 1e3, "rows_per_s": M_ * n_rows / t_dev,
                                           "x_single": M_ * n_rows / t_dev / single_now, "x_headline_value": M_ * n_rows / t_dev / dev_only_value, "wgrad_ksplit_per_image": st.ksplit,
                                           "e2e_ms_per_stacked_iteration": t_e2e * 1e3, "e2e_rows_per_s": M_ * n_rows / t_e2e,
                                           "e2e_x_single": (M_ * e2e1 * 1e-3 / t_e2e) if e2e1 else None,
                                           "mlp_launch_us": {k_: round(v_, 1) for k_, v_ in kus_.items()},
                                           "mlp_launch_mfma_frac": {k_: round(v_, 4) for k_, v_ in kfr_.items()}}
              st.close()
              del st, fs
              torch.cuda.empty_cache()
              if M_ == 8:
                  # what `python -m npp_amd.run --stack 8` runs by default: every image on the reference's own random stream (host
                  # draws of the 8 images in parallel threads, device half one iteration ahead on the sampler stream)
                  fs = []
                  for i_ in range(M_):
                      im_, mk_ = syn.synthetic_image(H, seed=2000 + i_)
                      fs.append(CompletionFit(im_, mk_, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=2000 + i_), device=dev,
                                              N_rand=8192, seed=2000 + i_, shifts=shifts, rng_mode="reference"))
                  st = StackedFit(fs)
                  for _ in range(5):
                      st.step_full()
                  torch.cuda.synchronize()
                  t6 = time.perf_counter()
                  for _ in range(100):
                      st.step_full()
                  torch.cuda.synchronize()
                  t_ref = (time.perf_counter() - t6) / 100
                  stacked["stacked_M8"]["e2e_reference_rng_ms_per_stacked_iteration"] = t_ref * 1e3
                  stacked["stacked_M8"]["e2e_reference_rng_rows_per_s"] = M_ * n_rows / t_ref
                  st.close()
                  del st, fs
                  torch.cuda.empty_cache()
          except Exception as ex_:                        # (e.g. M = 8 at 1024^2: 96 patches of 160^2 exceed one trunk launch)
            stacked[f"stacked_M{M_}"] = {"error": str(ex_)[:200]}

    if stacked and isinstance(stacked.get("stacked_M8"), dict) and "mlp_launch_mfma_frac" in stacked["stacked_M8"]:
        fr8 = stacked["stacked_M8"]["mlp_launch_mfma_frac"]
        roofline["stacked_M8_fwd_mfma_frac"] = fr8.get("mlp_fwd_stack")
        roofline["stacked_M8_bwd_mfma_frac"] = fr8.get("mlp_bwd_patch_stack")
        roofline["stacked_M8_wgrad_mfma_frac"] = fr8.get("mlp_wgrad_stack")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(K, H, patch, n_pix, fit.patch_num, fit.topk)

    if rank == 0:
        line = {
            "metric": "fitted pixels/sec/GPU (512^2 grid, 256-wide MLP) + iters-to-target-PSNR",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "dtype_note": ("bf16 MFMA operands / fp32 accumulate, master weights, Adam and losses in the forward and the data-gradient chain; "
                           + ("the weight-gradient products read the 8-bit training stash (bf8 x fp8, fp32 accumulate; npp_tune stash8=0 restores bf16 x bf16) -- "
                              "parity: the reference's full 2000-iteration c2 fit reproduced to 55 dB between the two fitted images, every golden trajectory within 0.1 dB"
                              if stash8 else "16-bit training stash")),
            "config": {"workload": f"c2: {H}x{H} completion image, top-{K} proposals, 256-wide x 8-layer NPP_Net; "
                                   f"step = 1 complete optimisation iteration (train.py:133-264) over {n_pix} pixel rows + "
                                   f"2x{patch}^2 patch rows: fused embed+MLP fwd, adaptive robust pixel loss, patch plumbing, "
                                   f"VGG19 trunk + contextual loss + trunk dgrad (+ VGG16/LPIPS on 'same'), bwd chain, wgrad, Adam",
                       "rows_per_step": n_rows, "image": [H, H], "K": K, "width": 256, "ksplit": net.ksplit,
                       "images_per_gpu": 1, "patch_size": patch, "patch_source_mix_in_pool": mix,
                       "trunk_dtype": "fp16 forward / bf16 gradient MFMA, fp32 accumulate",
                       # flat copies of the report's other headline numbers (the driver's parser keeps scalars of `config` only)
                       "value_is": (f"median of {args.windows} consecutive windows of --steps iterations of the loop that samples (reference random stream, producer thread)"
                                    if args.windows > 0 else "--windows 0: the pre-drawn-pool loop (device only) -- a profiling run, not the headline definition"),
                       "window_rows_per_s_min": min(win_rates), "window_rows_per_s_max": max(win_rates),
                       "window_rows_per_s_all": ", ".join(f"{r_:.4g}" for r_ in win_rates),
                       "timed_source_mix": f"val {win_mix[med]['val']} / train {win_mix[med]['train']} / same {win_mix[med]['same']} (the median window; the stream's own draw)",
                       "device_only_rows_per_s": dev_only_value, "device_only_ms_per_step": dev_only_ms_per_step,
                       "device_only_timed_source_mix": f"val {mix_timed['val']} / train {mix_timed['train']} / same {mix_timed['same']}",
                       "device_only_mix_is_50_30_20": bool(mix_ok),
                       "training_stash": "8-bit" if stash8 else "16-bit",
                       "rows_per_s_incl_sampling_reference_rng": _e2e_rate(e2e, "same_stream_native_rng_producer_thread", n_rows),
                       "rows_per_s_incl_sampling_fast_rng": _e2e_rate(e2e, "fast_mode", n_rows),
                       "render_pixels_per_s": render_px_s, "iters_to_28dB": iters_to_target, "psnr_known_after_300_iters_dB": final_psnr,
                       "stacked_M8_rows_per_s_per_gpu": _dig(stacked, "stacked_M8", "rows_per_s"),
                       "stacked_M8_x_single": _dig(stacked, "stacked_M8", "x_single"),
                       "stacked_M8_rows_per_s_incl_sampling_reference_rng": _dig(stacked, "stacked_M8", "e2e_reference_rng_rows_per_s"),
                       "stacked_M4_rows_per_s_per_gpu": _dig(stacked, "stacked_M4", "rows_per_s"),
                       "c4_embedder_1024sq_fp32_frac_of_hbm_peak": _dig(c4, "fp32", "frac_of_hbm_peak"),
                       "c4_render_1024sq_fp32_pixels_per_s": _dig(c4, "render_1024sq_fp32_fused", "pixels_per_s"),
                       "ms_per_iter_val": _dig(per_source, "val"), "ms_per_iter_train": _dig(per_source, "train"),
                       "ms_per_iter_same": _dig(per_source, "same"), "host_enqueue_ms_per_iter": host_enqueue_ms},
            "mlp_only_step": {"ms_per_step": mlp_ms, "rows_per_s": n_rows / (mlp_ms * 1e-3),
                              "mlp_mfma_frac": 2 * train_macs * n_rows / (mlp_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                              },
            "value_per_gpu": value / world,
            "value_incl_sampling": (n_rows / (e2e["same_stream_native_rng_producer_thread"]["ms_per_iter"] * 1e-3)
                                    if e2e and "same_stream_native_rng_producer_thread" in e2e else None),
            "render_pixels_per_s_per_gpu": render_px_s,
            "iters_to_28dB": iters_to_target, "psnr_known_after_300_iters": final_psnr,
            "final_gather_ms": gather_ms, "per_rank_rows_per_s": per_rank if per_rank is not None else [value],
            "device_only_per_rank_rows_per_s": dev_only_per_rank,
            "collective": {"backend": None, "ranks": 1} if dist is None else {"backend": backend + (" (RCCL)" if backend == "nccl" else ""),
                                                                              "ranks": dist.get_world_size()},
            "end_to_end_incl_host_sampling": e2e or None,
            "c4_embedder_1024sq": c4, "proposal_ranking_candidate": ranking, "throughput_mode_2_images_per_gpu": two_fits, "ms_per_iter_by_patch_source": per_source,
            "netwidth_512_fused": w512, "remapping_task_1024sq": remap, "stacked_images_per_gpu": stacked,
            "host_enqueue_ms_per_iter": host_enqueue_ms, "all_ranks_incl_sampling": e2e_ranks,
            "patch_loss_kernels_us": {k_: round(v_ * 1e6, 1) for k_, v_ in patch_kt.items()},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
