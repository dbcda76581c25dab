"""Samples-per-pixel sharding across GPUs.

The seed of a path depends only on (pixel, global sample index) -- TEA(pixel, frame * spp + i),
the deterministic stand-in for ray_gen.rgen:40 -- so frames are independent units of work.  Rank r
of N renders frames r, r + N, r + 2N, ... into its own float32 sum buffer (rgb += frame result,
a += 1) and one sum all-reduce (RCCL on GPUs, gloo in the CPU tests) yields the buffer a single
process would have produced, up to float summation order.  There is no other exchange step.
"""


def frames_for_rank(rank, world, n_frames, first_frame=0):
    """Frame indices rank `rank` renders out of [first_frame, first_frame + n_frames)."""
    return list(range(first_frame + rank, first_frame + n_frames, world))


def allreduce_sum(accum):
    """In-place sum of the accumulation buffer over all ranks (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    return accum
