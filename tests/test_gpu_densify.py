"""Densification ("next" row N3) on the LIVE native trainer, on the GPU: strategy.GSStrategy drives a NativeTrainStep through
the reference's callback sequence (trainer.py:741-760: post_backward -> optimizer step -> post_optimizer_step), so the number
of Gaussians goes up (clone, split) and down (prune) between steps while everything that is sized by N follows: the handle's
grow-only scratch, the trainer's workspace (resize_workspace), the Morton re-sort (restore_spatial_order), the wave
ownership of the side-stream optimiser pass with a ragged last wave, the binning capacity (overflow redo).

After every change of N:
  * the optimiser state is attached to the right rows (rows carry an id in an SH column that never receives a gradient),
  * the next view's forward and backward are checked against the CPU oracle (integer buffers bit-exact, image 2e-4 with
    oracle-attributed outliers, gradients per row),
  * the two-pass optimiser (side stream) still leaves the rows it owns bit-identical to the one-pass kernel.
"""
import importlib

import numpy as np
import pytest
import torch

from tests.common import COLOUR_TOL, ROW_FLIP_BOUND, cams, check_colour_outliers, check_gradients_per_row, check_tile_traversal, make_view, scenes

pytestmark = pytest.mark.gpu
gut = importlib.import_module("3dgrut_amd")
native = importlib.import_module("3dgrut_amd.native")
strategy = importlib.import_module("3dgrut_amd.strategy")
oracle = importlib.import_module("oracle.oracle")
DEV = "cuda:0"
W = H = 512
FX = 700.0
EXTENT = 1.3
TAG = 47    # SH coefficient 15, blue: with SH degree 0 active it never receives a gradient, so it keeps whatever it is given


def _views():
    return [make_view("pinhole", W, H, cams.orbit_c2w(4.0, 50.0 * k + 7.0, 20.0), fx=FX, fy=FX) for k in range(4)]


def _batch(view, gt=None):
    b = gut.Batch(rays_ori=torch.as_tensor(view["ro"], device=DEV), rays_dir=torch.as_tensor(view["rd"], device=DEV),
                  T_to_world=torch.as_tensor(view["c2w"])[None], **view["intrinsics_kw"])
    b.rgb_gt = gt
    return b


def _state(st):
    return dict(raw=st.model.raw, features=st.model.features, m12=st.m12, v12=st.v12, m48=st.m48, v48=st.v48)


def _check_against_oracle(st, view, label):
    raster = st.raster
    n = st.model.num_gaussians
    rgba, dist, hits, vis = st.forward(_batch(view))
    act, sph = st.activate().cpu().numpy(), st.model.features.cpu().numpy()
    ref = oracle.forward(view["oracle_cam"], W, H, act, sph, view["ro"], view["rd"], sh_degree=st.model.n_active_features)
    stats = raster.stats()
    assert stats["num_particles"] == n and ref["M"] == stats["num_intersections"] > 10_000
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), (label, key)
    for key in ("unsorted_keys", "sorted_keys"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint64), ref[key]), (label, key)
    assert np.array_equal(raster.debug_buffer("tile_ranges").cpu().numpy().view(np.uint32).reshape(-1, 2), ref["tile_ranges"])
    # image and traversal depths by the same model as every other frame (round 4: no scene-specific bound any more): the colour
    # tolerance scales with the image's range (after a few optimiser steps against a disturbed target the colours leave [0, 1]);
    # a pixel near a threshold is allowed what ITS near-threshold entries can move it by; a tile's depth may differ only where one of
    # its rays' decisions is within FLIP_MARGIN_BOUND noise widths of a threshold
    margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, budget_bound=ROW_FLIP_BOUND)
    check_tile_traversal(raster.debug_buffer("tile_traversed_fwd").cpu().numpy().view(np.uint32), ref["tile_traversed_fwd"], margins, W, H, label)
    check_colour_outliers(rgba.cpu().numpy(), hits.cpu().numpy(), ref, margins, label=label,
                          tol=COLOUR_TOL * max(1.0, float(np.abs(ref["rgba"]).max())), budget=pixel_budget)
    rgba_grad = np.random.default_rng(5).normal(size=(H, W, 4)).astype(np.float32)
    dens_g, sph_g, _, budget = oracle.backward(view["oracle_cam"], ref, rgba_grad, np.zeros((H, W, 1), np.float32), flip_bound=ROW_FLIP_BOUND)
    b, sensor, poses, rgba_, dist_ = st._ctx
    g12 = torch.empty((n, 12), dtype=torch.float32, device=DEV)
    g48 = torch.empty((n, 48), dtype=torch.float32, device=DEV)
    raster.trace_bwd(st.step_id, st.model.n_active_features, st.act, st.model.features, b.rays_ori.contiguous(), b.rays_dir.contiguous(),
                     None, sensor, poses.timestamps_us[0], poses.timestamps_us[1], poses.T_world_sensors[0], poses.T_world_sensors[1],
                     rgba_, torch.as_tensor(rgba_grad, device=DEV), dist_, None, out=(g12, g48))
    check_gradients_per_row(g12.cpu().numpy(), g48.cpu().numpy(), dens_g, sph_g, label, budget, sh_degree=st.model.n_active_features)
    check_tile_traversal(raster.debug_buffer("tile_traversed_bwd").cpu().numpy().view(np.uint32), ref["tile_traversed_bwd"], margins, W, H, label + " bwd")


def _check_state_follows_rows(before, st, label):
    """`before`: the trainer's state before the surgery, when row i carried id i.  Every row that existed before, survived and had
    optimiser state carries its own parameters and moments, whatever position it was moved to; every other row (clone, split
    child, or a Gaussian no view has touched yet) has all-zero moments and the SH colours of the row whose id it carries."""
    now = {k: v.cpu() for k, v in _state(st).items()}
    tags = now["features"][:, TAG].long()
    assert int(tags.min()) >= 0 and int(tags.max()) < before["raw"].shape[0]
    fresh = (now["m12"] == 0).all(1) & (now["v12"] == 0).all(1) & (now["m48"] == 0).all(1) & (now["v48"] == 0).all(1)
    old = ~fresh
    src = tags[old]
    assert src.unique().numel() == src.numel(), f"{label}: two rows with optimiser state share an id"
    for name in ("raw", "m12", "v12", "m48", "v48", "features"):
        assert torch.equal(now[name][old], before[name][src]), f"{label}: {name} is not attached to the rows it belonged to"
    assert torch.equal(now["features"][fresh], before["features"][tags[fresh]]), f"{label}: colours of the added rows"
    assert torch.equal(now["raw"][fresh][:, 3:8], before["raw"][tags[fresh]][:, 3:8]), f"{label}: density / rotation of the added rows"
    trained_before = int(((before["m12"] != 0).any(1) | (before["v12"] != 0).any(1)).sum())
    assert int(old.sum()) > 0.5 * trained_before > 0, (label, int(old.sum()), trained_before)
    return int(old.sum()), int(fresh.sum())


def test_densification_on_the_live_trainer():
    n0 = 200_000
    sc = scenes.scene_lego_like(n=n0, seed=6)
    views = _views()
    model = native.NativeGaussianModel(sc, device=DEV, sh_degree=0, spatial_order=True)
    with torch.no_grad():
        model.features[:, TAG] = torch.arange(n0, device=DEV, dtype=torch.float32)      # row ids (exact in fp32)
    st = native.NativeTrainStep(model, gut.Tracer({"render": {}}), scene_extent=EXTENT, overlap_optimizer=True)
    # targets: the same scene with its colours and positions disturbed -> real photometric gradients on what the views see
    with torch.no_grad():
        keep = {k: v.clone() for k, v in dict(raw=model.raw, features=model.features).items()}
        model.raw[:, 0:3] += 0.004 * torch.randn_like(model.raw[:, 0:3])
        model.features[:, 0:3] += 0.3 * torch.randn_like(model.features[:, 0:3])
        gts = [st.forward(_batch(v))[0][..., :3].clone()[None].contiguous() for v in views]
        model.raw.copy_(keep["raw"]); model.features.copy_(keep["features"])   # (in-place: the cached activation rows are dropped)
    gs = strategy.GSStrategy(st, seed=11, schedule=dict(densify=(2, 100, 4), prune=(2, 100, 6), reset_density=(-1, -1, 1),
                                                        density_decay=(-1, -1, 1))).attach()
    assert st.post_backward_hook is not None
    sizes, overflows0 = [n0], st.raster.stats()["binning_overflows"]
    changed = 0
    for step in range(0, 13):
        v = step % len(views)
        loss, _ = st.step(_batch(views[v], gts[v]))
        assert np.isfinite(float(loss))
        assert st.step_id == step + 1
        if step in (4, 8, 12):    # densify fires at 4, 8, 12; prune at 6, 12 (check_step_condition on the schedule above)
            # thresholds from the statistics actually gathered: about a third of the Gaussians the views touched densify
            seen = gs.grad_norm_denom.squeeze(1) > 0
            assert int(seen.sum()) > 20_000
            avg = (gs.grad_norm_accum.squeeze(1) / gs.grad_norm_denom.squeeze(1).clamp(min=1))[seen]
            gs.clone_thr = gs.split_thr = float(torch.quantile(avg[:1_000_000], 0.67))
            gs.rel_size = float(torch.exp(model.raw[:, 8:11]).max(1).values.median()) / EXTENT    # half clone, half split
        if step == 6:             # make sure the prune has something to remove, and that N ends ragged
            with torch.no_grad():
                st.model.raw[torch.arange(0, st.model.num_gaussians, 9, device=DEV), 3] = -9.0
        st.sync_moments()   # the stored moments are decayed lazily: bring them up to date before looking at them
        before = {k: v.clone().cpu() for k, v in _state(st).items()}
        n_before = st.model.num_gaussians
        if gs.post_optimizer_step(step, EXTENT):
            changed += 1
            n_now = st.model.num_gaussians
            sizes.append(n_now)
            kept, added = _check_state_follows_rows(before, st, f"step {step}")
            print(f"[densify] step {step}: N {n_before} -> {n_now}; {kept} rows kept their optimiser state, {added} rows without state")
            with torch.no_grad():   # fresh ids for the next round (the column never receives a gradient)
                st.model.features[:, TAG] = torch.arange(n_now, device=DEV, dtype=torch.float32)
            assert st.act.shape[0] == n_now and st.m48.shape[0] == n_now and gs.grad_norm_accum.shape[0] == n_now
            assert model.spatial_order and model.permutation.shape[0] == n_now
            assert (n_now < n_before) if step == 6 else (n_now > n_before)
            # Morton order restored: neighbouring rows are neighbours in space again
            p = st.model.raw[:, 0:3]
            assert float((p[1:] - p[:-1]).norm(dim=1).median()) < 0.1 * float((p[torch.randperm(n_now, device=DEV)] - p).norm(dim=1).median())
            _check_against_oracle(st, views[(step + 1) % len(views)], f"after step {step}, N = {n_now}")
    assert changed == 4 and len(set(sizes)) == len(sizes), sizes      # steps 4, 6, 8, 12
    assert max(sizes) > 1.15 * n0
    print(f"[densify] N over the run: {sizes}; binning overflows {st.raster.stats()['binning_overflows'] - overflows0}")
    # ---- the two-pass optimiser on the re-sized, re-sorted state: rows of the side stream's waves bit-identical to one pass ----
    from tests.test_gpu_native import _rows_in_unwalked_waves, exact_wave_mask
    gs.detach()
    n = st.model.num_gaussians
    if n % 64 == 0:
        gs.ops.keep(torch.arange(n, device=DEV) < n - 5)
        n = st.model.num_gaussians
    ref_model = native.NativeGaussianModel.from_tensors(st.model.raw.clone(), st.model.features.clone(), sh_degree=0, spatial_order=True)
    ref = native.NativeTrainStep(ref_model, gut.Tracer({"render": {}}), scene_extent=EXTENT, overlap_optimizer=False)
    ref.step_id = st.step_id
    for k in range(3):
        for name, t in _state(ref).items():
            t.copy_(_state(st)[name])
        ref.step_id = st.step_id
        ref.wave_step.copy_(st.wave_step)   # which step each wave's stored moments belong to travels with them
        b = _batch(views[k], gts[k])
        ref.step(b); st.step(b)
        torch.cuda.synchronize()
        cnt = st.raster.debug_buffer("tiles_count")
        assert cnt.numel() == n and torch.equal(cnt, ref.raster.debug_buffer("tiles_count"))
        exact = exact_wave_mask(cnt, _rows_in_unwalked_waves(st.raster, n))
        rows = st.raster.stats()["side_stream_rows"]
        assert rows == int(exact.sum()) and rows > 0 and ref.raster.stats()["side_stream_rows"] == 0
        tail = exact[n - (n % 64):]
        assert n % 64 != 0 and (bool(tail.all()) or bool((~tail).all()))     # the ragged last wave is owned as a whole
        print(f"[densify] two-pass step {k}: N = {n} ({n % 64} rows in the last wave), side stream took {rows} rows")
        for name, t in list(_state(st).items()) + [("act", st.act)]:
            r = ref.act if name == "act" else _state(ref)[name]
            assert torch.equal(r[exact], t[exact]), f"two-pass step {k}: {name} (rows of the side stream's waves)"
            differs = (r[~exact] - t[~exact]).abs() > 1e-6 + 1e-5 * r[~exact].abs()   # float-atomic noise of the backward, see test_gpu_fullsize
            assert float(differs.float().mean()) < 2e-3, f"two-pass step {k}: {name} (walked waves): {float(differs.float().mean())}"


def test_position_gradient_statistics_kernel_against_torch():
    """gut_position_gradient_statistics (GSStrategy.update_gradient_buffer on the GPU) against the reference's torch lines
    (gs.py:106-115) on strided [N,12] rows: same rows counted, accumulated norms to 1e-6."""
    g = torch.Generator().manual_seed(11)
    n = 10_003
    model = native.NativeGaussianModel(scenes.scene_c1(n, 3), device=DEV)
    st = native.NativeTrainStep(model, gut.Tracer({"render": {}}), scene_extent=1.0)
    gs = strategy.GSStrategy(st)
    g12 = (torch.randn((n, 12), generator=g) * 10.0 ** torch.empty((n, 1)).uniform_(-6, -2, generator=g)).to(DEV)
    g12[torch.rand(n, generator=g) < 0.4, 0:3] = 0.0
    g12[5, 0:3] = torch.tensor([0.0, 0.0, 1e-7])                       # one non-zero component is enough
    sensor = torch.tensor([0.3, -1.2, 2.0])
    gs.grad_norm_accum.uniform_(0.0, 1e-3)
    gs.grad_norm_denom.random_(0, 5)
    accum0, denom0 = gs.grad_norm_accum.clone(), gs.grad_norm_denom.clone()
    gs.update_gradient_buffer(g12[:, 0:3], sensor)                      # the kernel (GPU tensors)
    pg = g12[:, 0:3]
    mask = (pg != 0).max(dim=1)[0]
    dist = (model.raw[:, 0:3][mask] - sensor.to(DEV)).norm(dim=1, keepdim=True)
    accum0[mask] += torch.norm(pg[mask] * dist, dim=-1, keepdim=True) / 2
    denom0[mask] += 1
    assert 0.5 * n < int(mask.sum()) < 0.7 * n and bool(mask[5])
    assert torch.equal(gs.grad_norm_denom, denom0)
    assert torch.allclose(gs.grad_norm_accum, accum0, rtol=1e-6, atol=1e-12)


def test_statistics_fused_into_the_optimiser_equal_the_hook():
    """GSStrategy.attach() on the native trainer: with one view and the fused optimiser the statistics of gs.py:106-115 are
    accumulated by the optimiser kernel itself (gut_set_position_gradient_statistics) and the step keeps its side-stream pass;
    against the same trainer with the statistics taken by the hook between backward and optimiser: the same rows counted in every
    step, the same accumulated norms and the same parameters up to the float-atomic order of the backward."""
    sc = scenes.scene_c1(60_000, 9)
    views = _views()
    gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(2)).to(DEV)
    runs = {}
    for fused in (True, False):
        model = native.NativeGaussianModel(sc, device=DEV, spatial_order=True)
        st = native.NativeTrainStep(model, gut.Tracer({"render": {}}), scene_extent=EXTENT, overlap_optimizer=True)
        gs = strategy.GSStrategy(st).attach()
        assert st.post_backward_hook is not None and st.fused_statistics is not None
        if not fused:
            st.fused_statistics = None                    # the hook alone: backward epilogue -> hook -> optimiser
        calls = []
        inner = gs.update_gradient_buffer
        gs.update_gradient_buffer = lambda pg, sp: (calls.append(1), inner(pg, sp))[1]
        for k in range(5):
            st.step(_batch(views[k % 4], gt))
            if k == 1:   # the first step with statistics (step 0 has none, gs.py:64): the two trainers still hold the same parameters
                first = (gs.grad_norm_accum.clone(), gs.grad_norm_denom.clone())
        st.sync_moments()
        assert len(calls) == (0 if fused else 4)           # fused: the hook is never called (unfused: every step but step 0)
        assert st.overlap_optimizer is True
        runs[fused] = (first, gs.grad_norm_accum.clone(), gs.grad_norm_denom.clone(), model.raw.clone(), st.raster.stats())
    (f1, a1, d1, r1, s1), (f0, a0, d0, r0, s0) = runs[True], runs[False]
    assert s1["side_stream_rows"] > 0 and s0["side_stream_rows"] == 0   # only the fused form keeps the side-stream pass
    # one view's statistics from (up to the atomics of step 0's update) the same parameters: the same rows counted — but for a
    # handful whose gradient is a sum of float atomics that cancels to exactly zero in one order only — and the same norms
    assert int(f0[1].sum()) > 5_000 and int((f1[1] != f0[1]).sum()) <= 3
    same = (f1[1] == f0[1]).squeeze(1)
    rel = ((f1[0] - f0[0]).abs() / f0[0].abs().clamp_min(1e-3 * float(f0[0].max())))[same]
    assert float(rel.max()) <= 2e-3 and float(torch.quantile(rel[:1_000_000], 0.999)) <= 2e-4, (float(rel.max()),)
    # four views later the two runs have drifted apart by their atomics' orders; the statistics still agree as a whole
    assert int(d0.max()) >= 2 and int((d1 != d0).sum()) <= 5e-3 * d0.numel()
    assert float((a1 - a0).norm() / a0.norm()) <= 2e-2
    # (Adam's first steps move a parameter by about its learning rate whatever the size of its gradient: a row whose gradient is
    #  float-atomic noise around zero can go either way, so single values differ by a learning rate; the trainers agree as a whole)
    close = torch.isclose(r1, r0, rtol=1e-2, atol=1e-4)
    assert float(close.double().mean()) > 0.999 and float((r1 - r0).abs().max()) < 0.5, (float(close.double().mean()), float((r1 - r0).abs().max()))
    # detach at densify.end_iteration: both entries go
    gs.detach()
    assert st.post_backward_hook is None and st.fused_statistics is None
