"""Multi-GPU layer on CPU: partition arithmetic and the optional gather over a world_size-2 gloo group."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_bounds_cover_and_balance():
    from ddsp_svc_amd.sharding import shard_bounds, shard_counts
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            c = shard_counts(n, world)
            assert sum(c) == n and max(c) - min(c) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, n_total, dst, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddsp_svc_amd.sharding import gather_utterances, take_shard
    T = 6
    full = torch.arange(n_total * T, dtype=torch.float32).reshape(n_total, T)
    local = take_shard(full, rank, world) * 1.0          # each rank "synthesises" its own slice
    got = gather_utterances(local, n_total, dst=dst)
    ok = True
    if dst is None or rank == dst:
        ok = got is not None and torch.equal(got, full)
    else:
        ok = got is None
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,dst", [(8, 0), (5, 0), (5, None), (1, 0)])
def test_gather_world2_gloo(n_total, dst):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_total * 7 + (0 if dst is None else 3)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, dst, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
