"""Sample-sharded multi-GPU rendering: one process per GPU (torch.distributed,
backend "nccl" == RCCL on ROCm), no data-path collective while rendering, one
all-reduce(sum, float32) of the range histogram at the end.

The reference has no multi-process path at all (TBB over image blocks only,
src/librender/integrator.cpp:125-159); paths are i.i.d., so rank g renders the
global path indices [g*N/G, (g+1)*N/G) through bf_launch.path_offset and the
union is bit-identical in sample set to a 1-GPU run.
"""
import torch
import torch.distributed as dist


def shard_range(n_paths, rank, world):
    """Contiguous, exhaustive, non-overlapping split of [0, n_paths)."""
    lo = n_paths * rank // world
    hi = n_paths * (rank + 1) // world
    return lo, hi - lo


def render_sharded(render_fn, n_paths, n_floats, device="cpu", group=None):
    """render_fn(path_offset, count, out_tensor) accumulates `count` paths
    starting at global index `path_offset` into out_tensor (float32[n_floats]);
    returns the all-reduced histogram and this rank's (offset, count)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    off, cnt = shard_range(n_paths, rank, world)
    hist = torch.zeros(n_floats, dtype=torch.float32, device=device)
    render_fn(off, cnt, hist)
    if world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist, (off, cnt)


def render_cube_sharded(render_fn, n_paths, shape, device="cpu", group=None):
    """Sweep variant (BASELINE configs[4]): render_fn(path_offset, count, cube) accumulates this rank's share of the
    paths of EVERY frame / pulse into cube (float32[shape], e.g. [n_pulses, cells, 3]); the whole cube is reduced
    with ONE all-reduce after the last local kernel ("shard paths within each pulse")."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    off, cnt = shard_range(n_paths, rank, world)
    cube = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    render_fn(off, cnt, cube)
    if world > 1:
        dist.all_reduce(cube, op=dist.ReduceOp.SUM, group=group)
    return cube, (off, cnt)
