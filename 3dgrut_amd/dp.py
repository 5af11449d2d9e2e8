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


class RecordExchange:
    """Sparse gradient exchange WITHOUT a host round trip in the middle of the step (the forward's trick for the intersection
    count, applied to the record count): the payload all-gather is sized from the counts of PREVIOUS steps,

        start()        counts all-gather (device) + all-gather of `capacity` records per rank — both queued, nothing read back;
                       the caller scatters view v's first min(count_v, capacity) records with the count taken ON THE DEVICE
                       (gut_scatter_gradient_records_dev);
        host_counts()  the W counts on the host: waits for a 4 W-byte copy made on a side stream right behind the counts
                       collective — by then the payload all-gather and the scatters are already queued, the GPU does not idle;
        tail()         only when some rank had more records than the capacity assumed: all-gather of records[capacity:max],
                       to be scattered (host counts) BEFORE the optimiser kernel; every rank takes the same path, so the sums are
                       still formed in one order everywhere and the replicas stay bit-identical.

    The first exchange (and any exchange after reset()) has no history: it reads the counts first, like the reference's forward.
    capacity = 1.25 x the largest count of the recent steps (slowly forgetting old peaks) + min_capacity, a multiple of `granule`."""

    def __init__(self, world, group=None, min_capacity=1024, granule=1024):
        self.world, self.group, self.granule = int(world), group, int(granule)
        self.capacity = 0
        self._peak = 0.0
        self.min_capacity = int(min_capacity)
        self._flat = None
        self._copy_stream = self._copied = self._host = None
        self.counts = None
        self.overflows = 0
        self.payload_bytes_per_rank = 0   # bytes this rank received in the last exchange (diagnostics / bench)

    def reset(self):
        self.capacity, self._peak = 0, 0.0

    def _buffer(self, rows, width, like):
        need = self.world * max(rows, 1) * width
        if self._flat is None or self._flat.numel() < need or self._flat.device != like.device or self._flat.dtype != like.dtype:
            self._flat = torch.empty(int(need * 1.25) + width, dtype=like.dtype, device=like.device)
        return self._flat[:self.world * rows * width].view(self.world, rows, width)

    def _gather_counts(self, count):
        cnt = count.reshape(1).to(torch.int32)
        if _skip(self.world):
            return cnt.clone()
        if _stage_on_cpu(cnt, self.group) or not cnt.is_cuda:
            parts = [torch.zeros(1, dtype=torch.int32) for _ in range(self.world)]
            dist.all_gather(parts, cnt.cpu(), group=self.group)
            return torch.cat(parts).to(cnt.device)
        allc = torch.empty(self.world, dtype=torch.int32, device=cnt.device)
        dist.all_gather_into_tensor(allc, cnt, group=self.group)
        return allc

    def _counts_to_host(self, allc):
        if not allc.is_cuda:
            self._pending = None
            self.counts = [int(x) for x in allc.tolist()]
            return
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=allc.device)
            self._copied = torch.cuda.Event()
            self._host = torch.empty(self.world, dtype=torch.int32).pin_memory()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(allc.device))
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            self._host.copy_(allc, non_blocking=True)
            self._copied.record(self._copy_stream)
        allc.record_stream(self._copy_stream)
        self._pending = True
        self.counts = None

    def start(self, records, count):
        """records [rows_max, width] (this rank's first `count` rows are valid), count: one-element integer tensor on the same
        device.  Returns (gathered [W, capacity, width], counts_dev [W] int32, capacity)."""
        allc = self._gather_counts(count)
        if self.capacity == 0:            # no history: size the payload from the real counts (blocks once)
            self.counts = [int(x) for x in allc.cpu().tolist()]
            self._pending = None
            self._update_capacity(max(self.counts))
        else:
            self._counts_to_host(allc)
        cap = min(self.capacity, records.shape[0])
        gathered = self._buffer(cap, records.shape[1], records)
        if cap:
            if _skip(self.world):
                gathered[0].copy_(records[:cap])
            else:
                allgather_rows_(gathered, records[:cap], self.world, self.group)
        self._cap_used = cap
        self.payload_bytes_per_rank = self.world * cap * records.shape[1] * records.element_size()
        return gathered, allc, cap

    def host_counts(self):
        if self.counts is None:
            self._copied.synchronize()
            self.counts = [int(x) for x in self._host.tolist()]
        return self.counts

    def _update_capacity(self, maxc):
        self._peak = max(float(maxc), self._peak * 0.999)
        want = int(self._peak * 1.25) + self.min_capacity
        self.capacity = (want + self.granule - 1) // self.granule * self.granule

    def tail(self, records):
        """After host_counts(): None, or (gathered_tail [W, extra, width], [extra_v per view]) when a rank had more records than
        the payload carried; also moves the capacity for the next step."""
        counts = self.host_counts()
        cap, maxc = self._cap_used, max(counts)
        out = None
        if maxc > cap:
            self.overflows += 1
            extra = maxc - cap
            pad = records
            if records.shape[0] < maxc:   # (cannot happen: the record buffer holds one record per Gaussian)
                raise RuntimeError("record buffer smaller than a rank's record count")
            g = torch.empty((self.world, extra, records.shape[1]), dtype=records.dtype, device=records.device)
            if _skip(self.world):
                g[0].copy_(pad[cap:maxc])
            else:
                allgather_rows_(g, pad[cap:maxc].contiguous(), self.world, self.group)
            self.payload_bytes_per_rank += g.numel() * g.element_size()
            out = (g, [max(0, c - cap) for c in counts])
        self._update_capacity(maxc)
        return out


def replica_checksums(tensors):
    """One int64 per tensor: the wrapping sum of its bits taken as int32 words (exact, order-independent)."""
    return torch.stack([t.detach().contiguous().view(torch.int32).sum(dtype=torch.int64) for t in tensors])


def assert_replicas_identical(tensors, world: int, group=None, what="trainer state"):
    """Raises RuntimeError on EVERY rank unless all ranks hold bit-identical copies of `tensors` (checksums MIN- and MAX-reduced:
    two small collectives).  The data-parallel step relies on the replicas staying identical without ever comparing them; this is
    the comparison, run on step 0 and every few hundred steps (NativeTrainStep.replica_check_every)."""
    if _skip(world):
        return
    cs = replica_checksums(tensors)
    lo, hi = cs.clone(), cs.clone()
    if _stage_on_cpu(cs, group):
        lo, hi = lo.cpu(), hi.cpu()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo.cpu(), hi.cpu()):
        bad = [i for i, (a, b) in enumerate(zip(lo.cpu().tolist(), hi.cpu().tolist())) if a != b]
        raise RuntimeError(f"[3dgut] data-parallel replicas diverged: {what}, tensors {bad} differ between ranks")


def collective_info(world: int, group=None):
    """What the communicator looks like from this rank (for the bench line of an N > 1 run)."""
    info = dict(backend=None, world_size_seen=1, rccl_version=None)
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend(group)
        info["world_size_seen"] = dist.get_world_size(group)
        if info["backend"] == "nccl":   # "nccl" IS RCCL on ROCm
            try:
                info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:
                info["rccl_version"] = f"unavailable ({type(e).__name__})"
    return info
