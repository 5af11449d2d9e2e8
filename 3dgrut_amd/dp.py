"""Per-view data parallelism for the 3DGUT train step (new functionality — the reference is single-GPU).

One process per GPU, full replica of the Gaussians, rank r renders view `step*world + r`; the per-Gaussian
gradients are additive over views, so one exchange per optimiser step: SUM all-reduce (RCCL over xGMI on the
GPU box, gloo in the CPU tests) followed by 1/world, which makes the step's loss the mean over its views.
No other collective is on the path; the visibility mask is MAX-reduced only when SelectiveAdam is on.
"""
import os

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return None


def _skip(world: int) -> bool:
    """World 1 needs no exchange.  GUT_DP_FORCE_COLLECTIVES=1 (with an initialised process group) still issues every
    collective so the RCCL call sequence can be exercised on a one-GPU box (bench.py --force-exchange)."""
    if world > 1:
        return False
    return not (os.environ.get("GUT_DP_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized())


def _stage_on_cpu(t, group=None) -> bool:
    """gloo cannot reduce device tensors on every build: stage through host memory there (tests only; the GPU box
    runs RCCL, which takes device tensors directly)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def view_index(step: int, rank: int, world: int, n_views: int) -> int:
    """Rank r's view in step s: a contiguous block of `world` views per step, wrapped over the dataset."""
    return (step * world + rank) % n_views


def allreduce_mean_(tensors, world: int, group=None):
    """In-place mean over ranks of every tensor in `tensors` (largest first so the big SH message starts early)."""
    if _skip(world):
        return
    inv = 1.0 / world
    if any(_stage_on_cpu(t, group) for t in tensors):
        for t in tensors:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h.mul_(inv))
        return
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
             for t in sorted(tensors, key=lambda t: -t.numel())]
    for w in works:
        w.wait()
    for t in tensors:
        t.mul_(inv)


def allreduce_max_(tensor, world: int, group=None):
    if not _skip(world):
        if _stage_on_cpu(tensor, group):
            h = tensor.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            tensor.copy_(h)
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group)


def allreduce_max_async(tensor, world: int, group=None):
    """MAX all-reduce in place; returns an object with .wait() (orders the CURRENT stream behind the collective)."""
    if _skip(world):
        return _Done()
    if _stage_on_cpu(tensor, group):
        h = tensor.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        tensor.copy_(h)
        return _Done()
    return dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group, async_op=True)


def allreduce_sum_async(tensor, world: int, group=None):
    """SUM all-reduce; returns an object with .wait() (no-op object for world 1 / host-staged backends)."""
    class _Done:
        def wait(self):
            return None
    if _skip(world):
        return _Done()
    if _stage_on_cpu(tensor, group):
        h = tensor.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        tensor.copy_(h)
        return _Done()
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allgather_rows_(out, local, world: int, group=None):
    """out[world, ...] <- every rank's `local` tensor (rank order).  out[rank] may alias local."""
    if _skip(world):
        if out[0].data_ptr() != local.data_ptr():
            out[0].copy_(local)
        return
    if _stage_on_cpu(local, group):
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, local.cpu(), group=group)
        for r, t in enumerate(parts):
            out[r].copy_(t)
        return
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant (gloo)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        for r, t in enumerate(parts):
            out[r].copy_(t)


class _Done:
    def wait(self):
        return None


def allgather_rows_async(out, local, world: int, group=None):
    """out[world, rows, ...] <- every rank's `local` [rows, ...]; returns an object with .wait()."""
    if _skip(world):
        out[0].copy_(local)
        return _Done()
    if _stage_on_cpu(local, group):
        allgather_rows_(out, local, world, group)
        return _Done()
    return dist.all_gather_into_tensor(out, local, group=group, async_op=True)


def exchange_gradient_records(records, count, world: int, scratch: dict, group=None):
    """Sparse gradient exchange: every rank contributes the first `count` rows of its `records` [capacity, 16] tensor (one
    64-byte record per Gaussian the rank's view gave a gradient to, gut_compact_gradient_rows).  Returns (gathered, counts):
    gathered[r, :counts[r]] are rank r's records, identical on every rank.  Two collectives: the W counts (then ONE host
    read-back, which sizes the payload), and an all-gather of max(counts) records per rank — what crosses xGMI scales with
    the Gaussians the views actually touched, not with the size of the scene.
    `scratch` keeps the receive buffer between steps (grown on demand).  `count`: one-element integer tensor on records' device."""
    dev = records.device
    if _skip(world):
        c = int(count.item())
        return records[None, :c], [c]
    cnt = count.to(torch.int64).reshape(1)
    if _stage_on_cpu(cnt, group) or not cnt.is_cuda:
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, cnt.cpu(), group=group)
        counts = [int(p.item()) for p in parts]
    else:
        allc = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt, group=group)
        counts = [int(x) for x in allc.cpu().tolist()]
    maxc = max(counts)
    width = records.shape[1]
    need = world * max(maxc, 1) * width
    flat = scratch.get("flat")
    if flat is None or flat.numel() < need or flat.device != dev:
        flat = torch.empty(int(need * 1.25) + width, dtype=records.dtype, device=dev)
        scratch["flat"] = flat
    gathered = flat[:world * maxc * width].view(world, maxc, width)
    if maxc:
        allgather_rows_(gathered, records[:maxc], world, group)
    return gathered, counts
