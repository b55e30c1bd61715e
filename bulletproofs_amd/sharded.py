"""Point-range sharded MSM across the GPUs of one node.

Each rank (one process per GPU) runs the single-GPU pipeline on its contiguous range; the
only exchange is an all_gather of the 72-byte normalised partial results (RCCL over xGMI
with the "nccl" backend, gloo in the CPU tests), summed with bpgpu_points_sum.
"""
import numpy as np

from . import ffi


def shard_range(n: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank's points: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def combine_partials(curve: int, part_xy_inf: np.ndarray, device=None):
    """all_gather the per-rank partial (9 x u64) and return the sum (9 x u64) on every rank."""
    import torch
    import torch.distributed as dist

    part = np.ascontiguousarray(part_xy_inf, dtype=np.uint64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return part
    mine = torch.from_numpy(part.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    bufs = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, mine)
    allp = torch.stack(bufs).cpu().numpy().view(np.uint64)
    return ffi.points_sum(curve, allp)


def msm_sharded(curve: int, d_points, d_scalars, n_local: int, device=None):
    """This rank's MSM over its device-resident shard, combined across ranks."""
    return combine_partials(curve, ffi.msm_device(curve, d_points, d_scalars, n_local), device)
