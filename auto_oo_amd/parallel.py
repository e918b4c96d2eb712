"""Geometry sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all.  The only naturally parallel
axis of the hot path is the batch of molecular geometries of a Berry-phase loop
(examples/Tutorial_Berry_phase.ipynb): every geometry owns its AO integrals (27 MB at cc-pVDZ
shape), nothing is split, and the per-geometry results (energy, gradient, Newton step ...: a few
KB) are exchanged ONCE per batch with a single all_gather -- latency-bound, so the collective is
kept out of the per-evaluation path.
"""
import torch


def shard_geometries(n_geom, rank, world):
    """Cyclic partition: geometry g lives on rank g % world."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_geom, world))


_INDEX_CACHE = {}


def _index_tensor(my_geoms, device):
    """Device copy of the shard's geometry indices, made once (a host-to-device copy inside a timed loop is a
    blocking call)."""
    key = (tuple(my_geoms), str(device))
    idx = _INDEX_CACHE.get(key)
    if idx is None:
        if len(_INDEX_CACHE) > 64:
            _INDEX_CACHE.clear()
        idx = _INDEX_CACHE[key] = torch.as_tensor(list(my_geoms), device=device)
    return idx


def gather_results(local, my_geoms, n_geom, dist=None):
    """local [len(my_geoms), n_out] -> [n_geom, n_out] on every rank (row g = geometry g).

    One all_gather of equally sized, zero-padded blocks (ranks may own one geometry more or less)."""
    n_out = local.shape[1]
    if dist is None or not dist.is_initialized():
        full = torch.zeros((n_geom, n_out), dtype=local.dtype, device=local.device)
        full[_index_tensor(my_geoms, local.device)] = local
        return full
    world = dist.get_world_size()
    per = (n_geom + world - 1) // world
    # gloo (CPU rehearsal) exchanges host tensors; nccl (= RCCL) exchanges device tensors over xGMI
    xdev = torch.device("cpu") if dist.get_backend() == "gloo" else local.device
    block = torch.zeros((per, n_out), dtype=local.dtype, device=xdev)
    block[:local.shape[0]] = local.to(xdev)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    # row of block r, position i  ->  geometry r + i * world (cyclic partition): ONE gather with a
    # cached index instead of a scatter per rank (this runs inside the benchmark's timed region)
    stacked = torch.stack(blocks).to(local.device)                     # [world, per, n_out]
    return stacked.permute(1, 0, 2).reshape(per * world, n_out)[:n_geom].contiguous()
