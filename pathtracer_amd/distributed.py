"""Sample-partitioned rendering over the GPUs of one node.

The reference is a single-device program; this is the multi-GPU extension.  Frames (one sample per
pixel each, PathTracer.cpp:729) are dealt round-robin: rank g of G renders frame indices
f with (f - first) % G == g.  The scene is replicated on every GPU, each rank accumulates into its
own RGBA32F buffer, and one all-reduce (sum) over that buffer - .w carries the sample count and sums
with the radiance - gives every rank the image of all samples.  Partitioned rendering needs the
per-frame reseed option (PT_OPT_RESEED): with the reference's carried LCG state frame f+1 of a pixel
cannot start before frame f.
"""
import numpy as np


def partition_samples(first_frame, num_frames, rank, world):
    """Frame indices rank `rank` renders: first_frame + rank, + world, ... below first_frame + num_frames."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(first_frame + rank, first_frame + num_frames, world))


def render_partitioned(render_frame, first_frame, num_frames, rank, world):
    """Calls render_frame(frame_index) for this rank's share; returns the number of frames rendered."""
    frames = partition_samples(first_frame, num_frames, rank, world)
    for f in frames:
        render_frame(f)
    return len(frames)


def partition_block(first_frame, num_frames, rank, world):
    """(first index, count) of the contiguous block of frame indices rank `rank` renders when the frames are dealt in blocks:
    the first num_frames % world ranks get one frame more.  With the per-frame reseed the image of a set of frames does not
    depend on who renders which, and one pt_render_samples call per rank lets the frame train run at its steady rate
    (a call of n frames takes n + numRays steps)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    per, extra = divmod(num_frames, world)
    return first_frame + rank * per + min(rank, extra), per + (1 if rank < extra else 0)


def render_block(ctx, first_frame, num_frames, rank, world):
    """Renders this rank's block with one pt_render_samples call; returns the number of frames rendered."""
    first, count = partition_block(first_frame, num_frames, rank, world)
    if count:
        ctx.render_samples(first, count)
    return count


def allreduce_accumulation_torch(ctx, group=None):
    """Sums the accumulation buffers of all ranks in place with torch.distributed (RCCL on GPUs).
    The context's buffer is aliased as a torch tensor, no copy."""
    import torch
    import torch.distributed as dist

    class _Alias:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}

    ctx.synchronize()
    t = torch.as_tensor(_Alias(ctx.accumulation_device_ptr(), ctx.W * ctx.H * 4), device=torch.device("cuda", ctx.device_id))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.synchronize()
    return t


def allreduce_accumulation_host(acc, group=None):
    """CPU twin for tests with the gloo backend: sums a host accumulation array [H,W,4] over ranks in place."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(acc))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    acc[...] = t.numpy()
    return acc
