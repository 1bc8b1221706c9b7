"""N>1 path on CPU: world-size-2 gloo run of the sharding + final gather."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd.parallel import shard_units, gather_fitted
    mine = shard_units(n_units, rank, world)
    outs = torch.stack([torch.full((4, 5, 3), float(u)) for u in mine]) if mine else torch.zeros((0, 4, 5, 3))
    stats = torch.tensor([[float(u), 10.0 * u] for u in mine]).reshape(-1, 2)
    go, gs = gather_fitted(outs, stats)
    flat = [int(t[0, 0, 0].item()) for g in go for t in g]
    q.put((rank, mine, flat, [s.tolist() for s in gs]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [5, 2])
def test_shard_and_gather_world2(n_units):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + n_units
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_units, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(u for _, mine, _, _ in res for u in mine)
    assert owned == list(range(n_units))                    # disjoint cover
    for _, _, flat, _ in res:
        assert flat == list(range(n_units))                 # every rank sees every unit, in order


def _worker_scalars(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd.parallel import shard_units, gather_unit_scalars
    mine = shard_units(n_units, rank, world)
    stats = torch.tensor([[float(u), 0.5 * u, 100.0 - u] for u in mine], dtype=torch.float32).reshape(-1, 3)
    table = gather_unit_scalars(stats, n_units)
    q.put((rank, table.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [9, 1])
def test_candidate_scores_gather_world2(n_units):
    """The proposal-ranking collective: every rank ends with the (n_candidates, 3) score table in candidate order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + n_units
    procs = [ctx.Process(target=_worker_scalars, args=(r, 2, port, n_units, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[float(u), 0.5 * u, 100.0 - u] for u in range(n_units)]
    for _, table in res:
        assert table == want


def test_shard_units_balanced():
    sys.path.insert(0, ROOT)
    from npp_amd.parallel import shard_units
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            parts = [shard_units(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
