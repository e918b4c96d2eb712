"""world_size-2 gloo test (CPU) of the geometry sharding + the single all_gather exchange."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from auto_oo_amd.parallel import gather_results, shard_geometries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_all_geometries():
    for n, w in [(64, 1), (64, 2), (64, 8), (10, 4), (3, 8)]:
        seen = sorted(g for r in range(w) for g in shard_geometries(n, r, w))
        assert seen == list(range(n))
    with pytest.raises(ValueError):
        shard_geometries(4, 2, 2)


def _fake_eval(g, n_out):
    return torch.arange(n_out, dtype=torch.float64) + 1000.0 * g


def _worker(rank, world, port, n_geom, n_out, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_geometries(n_geom, rank, world)
    local = torch.stack([_fake_eval(g, n_out) for g in mine]) if mine else torch.zeros((0, n_out),
                                                                                       dtype=torch.float64)
    full = gather_results(local, mine, n_geom, dist)
    expect = torch.stack([_fake_eval(g, n_out) for g in range(n_geom)])
    ret[rank] = bool(torch.equal(full, expect))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_geom", [8, 7])
def test_gather_world2_gloo(n_geom):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + n_geom
    mp.spawn(_worker, args=(world, port, n_geom, 5, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)) and len(ret) == world


def test_gather_single_process():
    mine = shard_geometries(5, 0, 1)
    local = torch.stack([_fake_eval(g, 3) for g in mine])
    assert torch.equal(gather_results(local, mine, 5, None), local)
