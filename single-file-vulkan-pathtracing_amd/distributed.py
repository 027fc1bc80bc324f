"""Multi-GPU sharding of the radiance loop: one process per GPU, pixel tiles interleaved over the
ranks, ONE collective per presented image (a sum-reduce of the zero-padded float films to rank 0
over RCCL/xGMI; `gloo` on CPU for the tests).

The path shards with no data-path exchange: a pixel's radiance depends only on (pixel, sample
index, scene) (raygen.rgen:47-48, 88-90 touch nothing but the own texel).  Every rank holds the
whole scene and builds the same LBVH; rank r renders the 8x8 pixel tiles with
(tile_x + tile_y) % world == r -- interleaved, because contiguous bands are badly unbalanced
(44 % of the image, the border, terminates after one ray).  Outside its tiles a rank's film is
exactly 0, so the sum over ranks reproduces the single-GPU film bit for bit (x + 0 == x).
"""
import numpy as np

TILE = 8


def tile_owner(width, height, world):
    """-> int array [H, W]: the rank that renders each pixel (mirror of ensure_work() in
    csrc/wavefront.hip)."""
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    return (tx + ty) % world


def owned_mask(width, height, rank, world):
    return tile_owner(width, height, world) == rank


def reduce_film(film_tensor, dst=0, group=None):
    """Sum the per-rank films into rank `dst` (in place). film_tensor: torch tensor [H, W, 3] f32
    living where the process group's backend expects it (cuda for nccl/RCCL, cpu for gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(film_tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return film_tensor


def sum_counters(values, device, group=None):
    """All-reduce a few exact integer counters (ray counts) as int64."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(x) for x in t.tolist()]
