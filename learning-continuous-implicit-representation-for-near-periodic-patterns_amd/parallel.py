"""Image-/proposal-parallel sharding: one process per GPU, independent fits, and the single
collective of the job (gather of the fitted outputs).  The reference has no equivalent (its
only multi-GPU hook, nn.DataParallel at models/helpers.py:135-137, splits the pixel batch and
is never active in the scripts); SURVEY.md 8e.  Backend 'nccl' (= RCCL over xGMI) on GPUs,
'gloo' in the CPU tests."""
import torch
import torch.distributed as dist


def shard_units(n_units, rank, world):
    """Units (images or periodicity proposals) owned by `rank`: contiguous, sizes differ by <= 1."""
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def gather_fitted(outputs, stats, group=None):
    """All ranks receive every rank's fitted images and per-image scalars.

    outputs: (n_local, H, W, 3) float32 (n_local may differ by one between ranks);
    stats:   (n_local, S) float32 (e.g. final PSNR, iterations to target).
    Returns (list of per-rank output tensors, list of per-rank stats tensors)."""
    world = dist.get_world_size(group)
    n_local = torch.tensor([outputs.shape[0]], dtype=torch.int64, device=outputs.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    n_max = int(max(int(c.item()) for c in counts))
    pad_o = outputs.new_zeros((n_max,) + tuple(outputs.shape[1:]))
    pad_s = stats.new_zeros((n_max,) + tuple(stats.shape[1:]))
    pad_o[:outputs.shape[0]] = outputs
    pad_s[:stats.shape[0]] = stats
    go = [torch.empty_like(pad_o) for _ in range(world)]
    gs = [torch.empty_like(pad_s) for _ in range(world)]
    dist.all_gather(go, pad_o, group=group)
    dist.all_gather(gs, pad_s, group=group)
    return ([g[:int(c.item())] for g, c in zip(go, counts)], [g[:int(c.item())] for g, c in zip(gs, counts)])


def gather_unit_scalars(stats, n_units, group=None):
    """Per-unit scalars of a job sharded with shard_units(): every rank passes the (n_local, S) rows of ITS units (in
    unit order) and receives the (n_units, S) table of all units in unit order.  Used for the candidate scores of the
    proposal ranking (NPP_proposal/search.py:199-215: distances -> topk), one candidate fit per GPU."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    S = stats.shape[1]
    n_max = (n_units + world - 1) // world
    pad = stats.new_zeros((n_max, S))
    pad[:stats.shape[0]] = stats
    got = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(got, pad, group=group)
    rows = [got[r][:len(shard_units(n_units, r, world))] for r in range(world)]
    return torch.cat(rows, 0)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script, n_ranks, argv, env=None, timeout=None):
    """Start `script` as n_ranks processes of ONE node under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1) and return its exit code.  The CALLER must not have touched the GPU: the ranks are fresh child processes
    (a process that has initialised HIP is never re-executed).  Replaces nn.DataParallel at models/helpers.py:135-137."""
    import os
    import subprocess
    import sys
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this driver (RCCL needs it)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    return subprocess.run(cmd, env=e, timeout=timeout).returncode
