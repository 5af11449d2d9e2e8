"""World-size-2 gloo test of the data-parallel host logic: disjoint view assignment and the
SUM-then-divide gradient exchange, run as real processes over torch.distributed on CPU."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

dp = importlib.import_module("3dgrut_amd.dp")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    g48 = torch.randn((500, 48), generator=g); g12 = torch.randn((500, 12), generator=g)
    ref48, ref12 = g48.clone(), g12.clone()
    dp.allreduce_mean_([g12, g48], world)
    vis = (torch.arange(500) % (rank + 2) == 0).float()
    dp.allreduce_max_(vis, world)
    views = [dp.view_index(s, rank, world, 7) for s in range(6)]
    torch.save(dict(g48=g48, g12=g12, ref48=ref48, ref12=ref12, vis=vis, views=views), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gradient_exchange(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{k}.pt")) for k in range(world)]
    mean48 = (r[0]["ref48"] + r[1]["ref48"]) / 2; mean12 = (r[0]["ref12"] + r[1]["ref12"]) / 2
    for k in range(world):
        assert torch.allclose(r[k]["g48"], mean48, atol=1e-6) and torch.allclose(r[k]["g12"], mean12, atol=1e-6)
    assert torch.equal(r[0]["g48"], r[1]["g48"])  # replicas see bit-identical reduced gradients
    exp_vis = ((torch.arange(500) % 2 == 0) | (torch.arange(500) % 3 == 0)).float()
    assert torch.equal(r[0]["vis"], exp_vis) and torch.equal(r[1]["vis"], exp_vis)
    # per step the ranks take consecutive, distinct views; over time all views are visited
    for s in range(6):
        assert r[0]["views"][s] != r[1]["views"][s] and r[1]["views"][s] == (r[0]["views"][s] + 1) % 7
    assert set(r[0]["views"]) | set(r[1]["views"]) == set(range(7))


def test_single_rank_is_a_no_op():
    t = torch.ones(4)
    dp.allreduce_mean_([t], 1)
    assert torch.equal(t, torch.ones(4)) and dp.view_index(5, 0, 1, 4) == 1


COUNTS = [(5, 40, 0), (0, 0, 0), (300, 7, 12), (20, 280, 3), (1, 2, 3)]   # ragged, all empty, beyond the carried capacity, within it


# the node's world size: one rank never has a record (its view looks away from the scene), counts ragged over two orders of magnitude,
# one step in which a single rank overflows the carried capacity, one in which every rank is empty
COUNTS8 = [(5, 40, 0, 17, 3, 0, 29, 1), (0, 0, 0, 0, 0, 0, 0, 0), (12, 7, 0, 300, 9, 2, 44, 8), (250, 280, 0, 3, 120, 64, 65, 1), (1, 2, 0, 4, 5, 6, 7, 8)]


def _records_worker(rank, world, port, out_dir, table=None, torn_rank=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = dp.RecordExchange(world, min_capacity=8, granule=8)
    outs = []
    for step, counts in enumerate(COUNTS if table is None else table):
        g = torch.Generator().manual_seed(1000 * step + rank)
        rec = torch.randn((300, 16), generator=g)          # capacity = "number of Gaussians"; rows beyond the count are stale
        cap_before = ex.capacity
        gathered, counts_dev, cap = ex.start(rec, torch.tensor([counts[rank]], dtype=torch.int32))
        # what gut_scatter_gradient_records_dev would take from the payload: min(count, capacity) records per view
        rows = [gathered[r, :min(int(counts_dev[r]), cap)].clone() for r in range(world)]
        got = ex.host_counts()
        tail = ex.tail(rec)
        if tail is not None:
            rows = [torch.cat([rows[r], tail[0][r, :tail[1][r]]]) for r in range(world)]
        outs.append(dict(counts=got, rows=rows, mine=rec[:counts[rank]].clone(), cap_before=cap_before, cap=cap, overflowed=tail is not None,
                         bytes=ex.payload_bytes_per_rank))
    torch.save(outs, os.path.join(out_dir, f"rec{rank}.pt"))
    # replica comparison: identical tensors pass, one differing bit raises on every rank
    same = [torch.arange(12.0).reshape(3, 4), torch.ones(5)]
    dp.assert_replicas_identical(same, world)
    diff = [t.clone() for t in same]
    if rank == torn_rank:
        diff[1][2] = 1.0000001
    try:
        dp.assert_replicas_identical(diff, world, what="test")
        raised = False
    except RuntimeError as e:
        raised = "tensors [1] differ" in str(e)
    torch.save(dict(raised=raised, info=dp.collective_info(world)), os.path.join(out_dir, f"chk{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_three_rank_sparse_record_exchange(tmp_path):
    """RecordExchange: every rank ends with every rank's records (ragged counts, zero counts, a step whose counts exceed the
    capacity carried from earlier steps -> tail exchange), bit-identical; the payload is sized from PREVIOUS steps' counts."""
    world = 3
    mp.spawn(_records_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rec{k}.pt")) for k in range(world)]
    for step, counts in enumerate(COUNTS):
        for k in range(world):
            assert r[k][step]["counts"] == list(counts)
            for src in range(world):
                assert r[k][step]["rows"][src].shape == (counts[src], 16)
                assert torch.equal(r[k][step]["rows"][src], r[src][step]["mine"])
    s = r[0]
    assert s[0]["cap_before"] == 0 and s[0]["cap"] == 64 and not s[0]["overflowed"]          # first step: sized from its own counts (40 * 1.25 + 8)
    assert s[1]["cap"] == 64 and s[2]["cap"] == 64 and s[2]["overflowed"]                   # 300 > 64: tail exchange
    assert s[3]["cap"] == 300 and not s[3]["overflowed"]                                     # capacity followed the peak (clamped to the buffer)
    assert s[4]["cap"] == 300 and s[4]["bytes"] == world * 300 * 64
    chk = [torch.load(os.path.join(tmp_path, f"chk{k}.pt")) for k in range(world)]
    assert all(c["raised"] for c in chk)
    assert chk[0]["info"]["backend"] == "gloo" and chk[0]["info"]["world_size_seen"] == world


def test_eight_rank_sparse_record_exchange(tmp_path):
    """The exchange at the node's world size (no 8-GPU node is available to the builder, so this is what the
    N = 8 bench run's host logic has been through): eight gloo ranks, one of them never has a record, ragged counts, a step in which
    ONE rank overflows the capacity carried from earlier steps (every rank must take the tail path), an all-empty step; every rank
    ends every step with every rank's records, bit-identical and in rank order; the replica self-check trips on every rank when the
    LAST rank's copy differs in one bit."""
    world = 8
    mp.spawn(_records_worker, args=(world, _free_port(), str(tmp_path), COUNTS8, world - 1), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rec{k}.pt")) for k in range(world)]
    for step, counts in enumerate(COUNTS8):
        for k in range(world):
            assert r[k][step]["counts"] == list(counts)
            for src in range(world):
                assert r[k][step]["rows"][src].shape == (counts[src], 16)
                assert torch.equal(r[k][step]["rows"][src], r[src][step]["mine"])
            assert r[k][step]["cap"] == r[0][step]["cap"] and r[k][step]["overflowed"] == r[0][step]["overflowed"]   # same path on every rank
    s = r[0]
    assert [x["overflowed"] for x in s] == [False, False, True, False, False]
    assert s[0]["cap"] == 64 and s[2]["cap"] == 64 and s[3]["cap"] == 300 and s[4]["bytes"] == world * 300 * 64
    chk = [torch.load(os.path.join(tmp_path, f"chk{k}.pt")) for k in range(world)]
    assert all(c["raised"] for c in chk) and chk[0]["info"]["world_size_seen"] == world


def test_single_rank_record_exchange():
    rec = torch.arange(64.0).reshape(4, 16)
    ex = dp.RecordExchange(1, min_capacity=2, granule=2)
    gathered, counts_dev, cap = ex.start(rec, torch.tensor([3], dtype=torch.int32))
    assert ex.host_counts() == [3] and int(counts_dev[0]) == 3 and cap == 4 and torch.equal(gathered[0, :3], rec[:3]) and ex.tail(rec) is None
    dp.assert_replicas_identical([rec], 1)
    assert dp.collective_info(1)["world_size_seen"] == 1


# ---------------------------------------------------------------------------------------------------
# densification under data parallelism (SURVEY §8e + N3): each rank accumulates the statistics of ITS view
# (strategy/gs.py:106-115); GSStrategy.densify(world) adds them up over the ranks, and the clone / split that follows is
# the same on every rank (children drawn from a generator seeded with (seed, step)): the replicas stay identical.
# ---------------------------------------------------------------------------------------------------
class _CpuStepper:
    def __init__(self, model):
        n = model.num_gaussians
        self.model = model
        g = torch.Generator().manual_seed(4)
        self.m12, self.v12 = torch.randn((n, 12), generator=g) * 1e-3, torch.rand((n, 12), generator=g) * 1e-6
        self.m48, self.v48 = torch.randn((n, 48), generator=g) * 1e-3, torch.rand((n, 48), generator=g) * 1e-6
        self.post_backward_hook, self.row_listeners, self.step_id = None, [], 600

    def resize_workspace(self):
        pass


def _densify_scene():
    scenes = importlib.import_module("3dgrut_amd.scenes")
    sc = scenes.scene_c1(300, 8)
    sc["scale"][:150] = 0.001   # clone candidates
    sc["scale"][150:] = 0.5     # split candidates
    return sc


def _view_grads(view, n=300):
    g = torch.Generator().manual_seed(50 + view)
    grad = torch.randn((n, 3), generator=g) * 10.0 ** torch.empty((n, 1)).uniform_(-5, -2, generator=g)
    grad[torch.rand(n, generator=g) < 0.5] = 0.0
    return grad, torch.tensor([0.5 * view, -1.0, 3.0 + view])


def _densify_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    native = importlib.import_module("3dgrut_amd.native"); strategy = importlib.import_module("3dgrut_amd.strategy")
    st = _CpuStepper(native.NativeGaussianModel(_densify_scene(), device="cpu"))
    gs = strategy.GSStrategy(st, seed=11).attach()
    for step in range(3):                                  # three steps, one view per rank and step
        st.post_backward_hook(*_view_grads(dp.view_index(step, rank, world, 6)))
    gs.densify(scene_extent=1.0, step=600, world=world)
    torch.save(dict(raw=st.model.raw, features=st.model.features, m12=st.m12, v48=st.v48, n=st.model.num_gaussians),
               os.path.join(out_dir, f"d{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_densification_statistics_are_additive_and_replicas_stay_identical(tmp_path):
    world = 2
    mp.spawn(_densify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"d{k}.pt")) for k in range(world)]
    assert r[0]["n"] == r[1]["n"] > 300
    for k in ("raw", "features", "m12", "v48"):
        assert torch.equal(r[0][k], r[1][k]), k           # bit-identical replicas after clone + split
    # one process that saw all six views' statistics densifies the same way
    native = importlib.import_module("3dgrut_amd.native"); strategy = importlib.import_module("3dgrut_amd.strategy")
    st = _CpuStepper(native.NativeGaussianModel(_densify_scene(), device="cpu"))
    gs = strategy.GSStrategy(st, seed=11).attach()
    for step in range(3):
        for rank in range(world):
            st.post_backward_hook(*_view_grads(dp.view_index(step, rank, world, 6)))
    gs.densify(scene_extent=1.0, step=600, world=1)
    assert st.model.num_gaussians == r[0]["n"]
    assert torch.allclose(st.model.raw, r[0]["raw"], rtol=1e-6, atol=1e-7) and torch.equal(st.m12, r[0]["m12"])
