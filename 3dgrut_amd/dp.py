"""Per-view data parallelism for the 3DGUT train step (new functionality — the reference is single-GPU).

One process per GPU, full replica of the Gaussians, rank r renders view `step*world + r`; the per-Gaussian
gradients are additive over views, so one exchange per optimiser step: SUM all-reduce (RCCL over xGMI on the
GPU box, gloo in the CPU tests) followed by 1/world, which makes the step's loss the mean over its views.
No other collective is on the path; the visibility mask is MAX-reduced only when SelectiveAdam is on.
"""
import torch
import torch.distributed as dist


def view_index(step: int, rank: int, world: int, n_views: int) -> int:
    """Rank r's view in step s: a contiguous block of `world` views per step, wrapped over the dataset."""
    return (step * world + rank) % n_views


def allreduce_mean_(tensors, world: int, group=None):
    """In-place mean over ranks of every tensor in `tensors` (largest first so the big SH message starts early)."""
    if world <= 1:
        return
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
             for t in sorted(tensors, key=lambda t: -t.numel())]
    for w in works:
        w.wait()
    inv = 1.0 / world
    for t in tensors:
        t.mul_(inv)


def allreduce_max_(tensor, world: int, group=None):
    if world > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group)


def allgather_rows_(out, local, world: int, group=None):
    """out[world, ...] <- every rank's `local` tensor (rank order).  out[rank] may alias local."""
    if world <= 1:
        if out[0].data_ptr() != local.data_ptr():
            out[0].copy_(local)
        return
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant (gloo)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        for r, t in enumerate(parts):
            out[r].copy_(t)
