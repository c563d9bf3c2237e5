"""Multi-GPU rendering: samples per pixel are split across ranks (one process per GPU), each rank
renders its sample range of the WHOLE frame into its own radiance-sum buffer, and one sum-reduce
(RCCL over xGMI when the backend is "nccl") brings the buffers together on rank 0, which divides by
the sample count and tone-maps (main.cpp:1226-1231 must run after the sum).  The reference has no
counterpart: it is one process with an OpenMP loop over pixels (main.cpp:1237).

The render itself is a callback so that the partition / reduce logic can be exercised on CPU with
the gloo backend.
"""


def sample_range(total_spp, world, rank):
    """Contiguous split of [0, total_spp) over `world` ranks; the first total_spp % world ranks get
    one extra sample.  Returns (begin, count); count may be 0 when there are more ranks than samples."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("rank %r outside world %r" % (rank, world))
    base, extra = divmod(int(total_spp), int(world))
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def render_frame(render_range, buf, total_spp, group=None):
    """render_range(begin, count, buf): write the radiance SUM of samples [begin, begin+count) of
    every pixel into `buf` (a torch tensor on this rank's device; zeros when count == 0).
    Afterwards rank 0's buf holds the sum over all samples.  Returns (begin, count) of this rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    begin, count = sample_range(total_spp, world, rank)
    if count > 0:
        render_range(begin, count, buf)
    else:
        buf.zero_()
    if world > 1:
        dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=group)
    return begin, count
