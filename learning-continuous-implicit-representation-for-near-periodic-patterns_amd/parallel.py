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
