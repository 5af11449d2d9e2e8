"""Parity at BASELINE.json's full size (bicycle-like stand-in: 6 M Gaussians, 1237x822), through the C ABI.

Two layers:
  * size-independent properties of the intermediate buffers and of the backward (scan = running sum of the counts, sorted
    keys ordered on the sorted bits and STABLE, sort is a permutation (order-independent checksums), tile ranges partition
    the list, image bounded, backward linear in the upstream gradient, culled Gaussians get exactly zero gradient,
    integer buffers reproducible run to run);
  * the CPU oracle on the very same inputs (a few seconds of host time on the GPU box's cores): every integer buffer
    bit-exact, projection floats bit-exact, colours within the tolerances of tests/test_gpu_parity.py.
"""
import importlib
import time

import numpy as np
import pytest
import torch

from tests.common import (ROW_FLIP_BOUND, cams, check_colour_outliers, check_gradients_per_row, check_side_stream_rows_are_gradient_free, rel_l2,
                          scenes)

pytestmark = pytest.mark.gpu
gut = importlib.import_module("3dgrut_amd")
native = importlib.import_module("3dgrut_amd.native")
pose = importlib.import_module("3dgrut_amd.pose")
DEV = "cuda:0"
N, W, H, FX = 6_000_000, 1237, 822, 1040.0


@pytest.fixture(scope="module")
def frame():
    sc = scenes.scene_outdoor_like(n=N, seed=2)
    model = native.NativeGaussianModel(sc, device=DEV, spatial_order=True)   # the bench's storage order (Morton)
    tracer = gut.Tracer({"render": {}})
    stepper = native.NativeTrainStep(model, tracer, scene_extent=5.0)
    ro, rd = cams.pinhole_rays(W, H, FX, FX)
    K = cams.pinhole_intrinsics_dict(W, H, FX, FX)
    c2w = cams.orbit_c2w(4.5, 7.0, 12.0)
    batch = gut.Batch(rays_ori=torch.as_tensor(ro, device=DEV), rays_dir=torch.as_tensor(rd, device=DEV),
                      T_to_world=torch.as_tensor(c2w)[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
    rgba, dist, hits, vis = stepper.forward(batch)
    raster = tracer.tracer_wrapper
    buf = {k: raster.debug_buffer(k) for k in ("tiles_count", "tiles_offset", "unsorted_keys", "unsorted_ids", "sorted_keys",
                                                "sorted_ids", "tile_ranges")}
    return dict(sc=sc, model=model, stepper=stepper, raster=raster, batch=batch, rgba=rgba, dist=dist, hits=hits, vis=vis, buf=buf,
                ro=ro, rd=rd, K=K, c2w=c2w, stats=raster.stats())


def _bwd(fr, rgba_grad):
    st = fr["stepper"]
    b, sensor, poses, rgba, dist_ = st._ctx
    g12 = torch.empty((N, 12), dtype=torch.float32, device=DEV)
    g48 = torch.empty((N, 48), dtype=torch.float32, device=DEV)
    fr["raster"].trace_bwd(st.step_id, fr["model"].n_active_features, st.act, fr["model"].features, b.rays_ori.contiguous(),
                           b.rays_dir.contiguous(), None, sensor, poses.timestamps_us[0], poses.timestamps_us[1],
                           poses.T_world_sensors[0], poses.T_world_sensors[1], rgba, rgba_grad, dist_, None, out=(g12, g48))
    return g12, g48


def test_structure_of_the_intermediate_buffers(frame):
    b, st = frame["buf"], frame["stats"]
    cnt = b["tiles_count"].long()
    M = int(cnt.sum())
    T = st["num_tiles"]
    assert M == st["num_intersections"] and M > 5_000_000
    assert torch.equal(torch.cumsum(cnt, 0), b["tiles_offset"].long() & 0xFFFFFFFF)           # K2: inclusive scan
    uk, sk = b["unsorted_keys"], b["sorted_keys"]
    ui, si = b["unsorted_ids"].long() & 0xFFFFFFFF, b["sorted_ids"].long() & 0xFFFFFFFF
    assert uk.numel() == M and sk.numel() == M
    # K3: every Gaussian emits exactly tiles_count entries, contiguous, in Gaussian order
    assert torch.equal(torch.bincount(ui, minlength=N), cnt)
    assert bool((ui[1:] >= ui[:-1]).all())
    # K4: a permutation of the pairs (order-independent checksums over keys and key-id pairs) ...
    assert int(uk.sum()) == int(sk.sum()) and int((uk ^ (ui * 0x9E3779B1)).sum()) == int((sk ^ (si * 0x9E3779B1)).sum())
    tile, depth = sk >> 32, sk & 0xFFFFFFFF
    assert int(tile.max()) < T
    assert bool((sk[1:] >= sk[:-1]).all())                                                     # ... ordered on (tile, depth bits)
    same = sk[1:] == sk[:-1]
    assert bool((si[1:][same] > si[:-1][same]).all())                                          # ... and stable
    # K5: ranges partition the list by tile
    rng = b["tile_ranges"].long().reshape(-1, 2) & 0xFFFFFFFF
    per_tile = torch.bincount(tile, minlength=T)
    assert torch.equal(rng[:, 1] - rng[:, 0], per_tile)
    nz = per_tile > 0
    assert torch.equal(rng[nz, 0], (torch.cumsum(per_tile, 0) - per_tile)[nz])
    # depth bits are the float bits of a positive distance
    d = depth.int().view(torch.float32)
    assert bool(torch.isfinite(d).all()) and float(d.min()) > 0.0


def test_image_bounds_and_visibility(frame):
    rgba, dist, hits, vis = frame["rgba"], frame["dist"], frame["hits"], frame["vis"]
    assert bool(torch.isfinite(rgba).all()) and bool(torch.isfinite(dist).all())
    assert float(rgba[..., 3].max()) <= 1.0 and float(rgba.min()) >= 0.0
    assert float(rgba[..., 3].mean()) > 0.9                      # the stand-in is an opaque outdoor scene
    assert float(hits.min()) >= 0.0 and float(hits.mean()) > 20.0
    cnt = frame["buf"]["tiles_count"]
    assert bool((vis.reshape(-1)[cnt > 0] > 0).all())            # a Gaussian with tiles is visible


def test_backward_is_linear_and_culled_gaussians_get_zero(frame):
    g = torch.Generator(device="cpu").manual_seed(9)
    g1 = torch.randn((H, W, 4), generator=g).to(DEV)
    g2 = torch.randn((H, W, 4), generator=g).to(DEV)
    a12, a48 = _bwd(frame, g1)
    b12, b48 = _bwd(frame, g2)
    c12, c48 = _bwd(frame, 0.7 * g1 - 1.3 * g2)
    assert bool(torch.isfinite(c12).all()) and bool(torch.isfinite(c48).all())
    for x, y, z in ((a12, b12, c12), (a48, b48, c48)):
        lin = 0.7 * x - 1.3 * y
        assert float((lin - z).norm() / z.norm()) <= 1e-4
    culled = frame["buf"]["tiles_count"] == 0
    assert int(culled.sum()) > 1_000_000
    assert float(c12[culled].abs().max()) == 0.0 and float(c48[culled].abs().max()) == 0.0


def test_integer_buffers_are_reproducible(frame):
    st = frame["stepper"]
    st.forward(frame["batch"])
    again = frame["raster"].debug_buffer("sorted_ids")
    assert torch.equal(again, frame["buf"]["sorted_ids"])
    assert torch.equal(frame["raster"].debug_buffer("sorted_keys"), frame["buf"]["sorted_keys"])


def test_full_size_frame_against_the_oracle(frame):
    """The CPU oracle on the identical 6 M-Gaussian frame: integer buffers and projection floats bit-exact, image within the
    tolerances of test_gpu_parity (rgba 2e-4 — see the note at the assertion —, hit counts on <= 0.1 % of the pixels)."""
    oracle = importlib.import_module("oracle.oracle")
    st = frame["stepper"]
    act = st.activate().cpu().numpy()
    sph = frame["model"].features.cpu().numpy()
    tq = pose.sensor_pose_from_c2w(frame["c2w"]).T_world_sensors[0]
    K = frame["K"]
    ocam = dict(model="pinhole", principal_point=K["principal_point"], focal_length=K["focal_length"], radial=K["radial_coeffs"],
                tangential=K["tangential_coeffs"], thin_prism=K["thin_prism_coeffs"], pose_start=tq)
    t0 = time.time()
    ref = oracle.forward(ocam, W, H, act, sph, frame["ro"], frame["rd"], sh_degree=3)
    print(f"oracle forward at full size: {time.time() - t0:.1f} s")
    b = frame["buf"]
    assert ref["M"] == frame["stats"]["num_intersections"]
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        assert np.array_equal(b[key].cpu().numpy().view(np.uint32), ref[key]), key
    for key in ("unsorted_keys", "sorted_keys"):
        assert np.array_equal(b[key].cpu().numpy().view(np.uint64), ref[key]), key
    assert np.array_equal(b["tile_ranges"].cpu().numpy().view(np.uint32).reshape(-1, 2), ref["tile_ranges"])
    # the lazily ordered lists the compositors walked are a prefix of the reference's sorted lists in every tile
    ordered = frame["raster"].debug_buffer("ordered_ids").cpu().numpy().view(np.uint32)
    written = ordered != 0xFFFFFFFF
    assert written.sum() >= frame["stats"]["traversed_fwd"] and np.array_equal(ordered[written], ref["sorted_ids"][written])
    for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
        got = frame["raster"].debug_buffer(key).cpu().numpy().view(np.uint32)
        exp = np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)
        assert np.array_equal(got, exp), f"{key}: {(got != exp).sum()} of {got.size} words differ"
    # colours: 2e-4 as in test_gpu_parity.  Over a million pixels x ~100 blended hits some responses sit within fp32 noise of
    # the min_response / min_alpha thresholds and flip between "hit" and "no hit" (hardware exp/rcp + FMA vs libm): such a
    # pixel is excused ONLY if the oracle's own per-pixel decision margins say so (tests/common.check_colour_outliers)
    margins, pixel_budget = oracle.render_margins(ocam, ref, budget_bound=ROW_FLIP_BOUND)
    check_colour_outliers(frame["rgba"].cpu().numpy(), frame["hits"].cpu().numpy(), ref, margins, label="bicycle_like_6M", budget=pixel_budget)
    assert frame["stats"]["traversed_fwd"] == ref["traversed_fwd"]
    # backward of the same frame: gradients w.r.t. the activated tracer inputs, relative L2 <= 2e-3 per parameter block
    rgba_grad = np.random.default_rng(4).normal(size=(H, W, 4)).astype(np.float32)
    t0 = time.time()
    dens_g, sph_g, _, budget = oracle.backward(ocam, ref, rgba_grad, np.zeros((H, W, 1), np.float32), flip_bound=ROW_FLIP_BOUND)
    print(f"oracle backward at full size: {time.time() - t0:.1f} s")
    g12, g48 = _bwd(frame, torch.as_tensor(rgba_grad, device=DEV))
    g12, g48 = g12.cpu().numpy(), g48.cpu().numpy()
    for name, sl in (("positions", slice(0, 3)), ("density", slice(3, 4)), ("rotation", slice(4, 8)), ("scale", slice(8, 11))):
        assert rel_l2(g12[:, sl], dens_g[:, sl]) <= 2e-3, name
    assert rel_l2(g48, sph_g) <= 2e-3
    assert frame["stats"]["num_intersections"] == ref["M"] and frame["raster"].stats()["traversed_bwd"] == ref["traversed_bwd"]
    # ... and per ROW (a block-wide L2 over 6 M rows would hide a few thousand wrong small rows)
    check_gradients_per_row(g12, g48, dens_g, sph_g, "bicycle_like_6M", budget)
    # the waves the side-stream optimiser pass takes from this frame: the oracle — which ends its rays by its own rule — and the
    # GPU both leave every one of their rows without a gradient, exactly
    from tests.test_gpu_native import _rows_in_unwalked_waves, exact_wave_mask
    owned = exact_wave_mask(frame["raster"].debug_buffer("tiles_count"), _rows_in_unwalked_waves(frame["raster"], N)).cpu().numpy()
    check_side_stream_rows_are_gradient_free(owned, g12, g48, dens_g, sph_g, "bicycle_like_6M", min_rows=4_500_000)


def test_two_pass_optimiser_at_full_size_with_finely_interleaved_rows():
    """The early (side-stream) and the late optimiser pass run CONCURRENTLY at this size and, with the scene's random row order,
    write rows that share cache lines (48-byte rows, 64/128-byte lines) from different XCDs.  Their union must still be exactly
    the one-pass result: rows without tiles bit-identical, rows with tiles to the float-atomic noise of the backward, for the
    parameters, both moments and the activations, over two steps on two views."""
    sc = scenes.scene_outdoor_like(n=N, seed=2)
    ro, rd = cams.pinhole_rays(W, H, FX, FX)
    K = cams.pinhole_intrinsics_dict(W, H, FX, FX)
    gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(4)).to(DEV)
    steppers = []
    for overlap in (False, True):
        model = native.NativeGaussianModel(sc, device=DEV)            # scene order: rows with / without tiles interleave finely
        steppers.append(native.NativeTrainStep(model, gut.Tracer({"render": {"enable_kernel_timings": True}}), scene_extent=5.0,
                                               overlap_optimizer=overlap))
    ref, ovl = steppers
    state = lambda st: dict(raw=st.model.raw, features=st.model.features, m12=st.m12, v12=st.v12, m48=st.m48, v48=st.v48, act=st.act)
    for k in range(2):
        for name, t in state(ovl).items():
            t.copy_(state(ref)[name])
        c2w = cams.orbit_c2w(4.5, 7.0 + 45.0 * k, 12.0)
        for st in (ref, ovl):
            st.step(gut.Batch(rays_ori=torch.as_tensor(ro, device=DEV), rays_dir=torch.as_tensor(rd, device=DEV),
                              T_to_world=torch.as_tensor(c2w)[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K))
        torch.cuda.synchronize()
        cnt = ovl.raster.debug_buffer("tiles_count")
        assert torch.equal(cnt, ref.raster.debug_buffer("tiles_count"))
        early = cnt == 0
        assert 1_000_000 < int(early.sum()) < N - 1_000_000
        for name, t in state(ovl).items():
            r = state(ref)[name]
            assert torch.equal(r[early], t[early]), f"step {k}: {name} (rows without tiles)"
            a, b = r[~early], t[~early]
            if name in ("m12", "m48", "v12", "v48"):
                # moments are (sums of) the gradients: equal up to the float-atomic noise of the compositing backward
                tol = 1e-5 * float(a.abs().max()) + 1e-12
                assert float((a - b).abs().max()) <= 2 * tol, f"step {k}: {name} (rows with tiles)"
            else:
                # Adam's update is ~ lr * sign(g) while the second moment is young: where a gradient component is pure atomic
                # noise its sign, hence the parameter, may differ by 2 lr; a clobbered row would differ everywhere
                differs = (a - b).abs() > 1e-6 + 1e-5 * a.abs()
                assert float(differs.float().mean()) < 2e-3, f"step {k}: {name}: {float(differs.float().mean())} of the elements differ"
                assert float((a - b).abs().max()) <= 0.12, f"step {k}: {name} (rows with tiles)"
    kt = ovl.raster.kernel_times()
    assert kt["optimizer_early"] > 0 and kt["optimizer"] > 0


def test_side_stream_optimiser_at_full_size_in_spatial_order():
    """The bench's configuration: Morton-ordered rows, so whole 64-row waves qualify for the side stream — the waves without
    tiles from the end of binning, the waves the forward walked nothing of from the start of the backward compositor: more than
    5 M of the 6 M rows are updated CONCURRENTLY with the compositing kernels and the pass over the walked waves.  Every row of
    those waves must be bit-identical to the one-pass step (parameters, both moments, activations), the rest equal up to the
    float-atomic noise of the backward, over three steps on three views."""
    from tests.test_gpu_native import _rows_in_unwalked_waves, exact_wave_mask
    sc = scenes.scene_outdoor_like(n=N, seed=2)
    ro, rd = cams.pinhole_rays(W, H, FX, FX)
    K = cams.pinhole_intrinsics_dict(W, H, FX, FX)
    gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(4)).to(DEV)
    steppers = []
    for overlap in (False, True):
        model = native.NativeGaussianModel(sc, device=DEV, spatial_order=True)
        steppers.append(native.NativeTrainStep(model, gut.Tracer({"render": {"enable_kernel_timings": True}}), scene_extent=5.0,
                                               overlap_optimizer=overlap))
    ref, ovl = steppers
    for st in steppers:   # non-zero moments everywhere, as in the bench: every row the side stream touches really changes
        g = torch.Generator(device=DEV).manual_seed(7)
        for m_, v_ in ((st.m12, st.v12), (st.m48, st.v48)):
            m_.normal_(0.0, 1e-6, generator=g)
            v_.fill_(1e-8)
    state = lambda st: dict(raw=st.model.raw, features=st.model.features, m12=st.m12, v12=st.v12, m48=st.m48, v48=st.v48, act=st.act)
    for k in range(3):
        for name, t in state(ovl).items():
            t.copy_(state(ref)[name])
        before = {name: t.clone() for name, t in state(ref).items() if name in ("raw", "m48")}
        c2w = cams.orbit_c2w(4.5, 7.0 + 45.0 * k, 12.0)
        for st in (ref, ovl):
            st.step(gut.Batch(rays_ori=torch.as_tensor(ro, device=DEV), rays_dir=torch.as_tensor(rd, device=DEV),
                              T_to_world=torch.as_tensor(c2w)[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K))
        torch.cuda.synchronize()
        cnt = ovl.raster.debug_buffer("tiles_count")
        assert torch.equal(cnt, ref.raster.debug_buffer("tiles_count"))
        exact = exact_wave_mask(cnt, _rows_in_unwalked_waves(ovl.raster, N))
        rows = int(exact.sum())
        assert rows > 4_500_000 and ovl.raster.stats()["side_stream_rows"] == rows
        for name, t in state(ovl).items():
            r = state(ref)[name]
            assert torch.equal(r[exact], t[exact]), f"step {k}: {name} (rows of the side stream's waves)"
            a, b = r[~exact], t[~exact]
            if name in ("m12", "m48", "v12", "v48"):
                tol = 1e-5 * float(a.abs().max()) + 1e-12
                assert float((a - b).abs().max()) <= 2 * tol, f"step {k}: {name} (rows of the walked waves)"
            else:
                differs = (a - b).abs() > 1e-6 + 1e-5 * a.abs()
                assert float(differs.float().mean()) < 2e-3, f"step {k}: {name}: {float(differs.float().mean())} of the elements differ"
        # and they did move (zero-gradient Adam step on non-zero moments), i.e. the comparison above is not vacuous
        assert float((state(ovl)["raw"][exact] - before["raw"][exact]).abs().max()) > 0
        # lazy moment decay: the stored moments of those waves are NOT rewritten (they are brought up to date in registers) ...
        assert torch.equal(state(ovl)["m48"][exact], before["m48"][exact]) and ovl.lazy_moments
        last_exact, last_before = exact, before
    # ... until something asks for them: both forms then hold the same, decayed, moments
    ref.sync_moments(); ovl.sync_moments()
    assert torch.equal(ref.m48[last_exact], ovl.m48[last_exact]) and torch.equal(ref.v12[last_exact], ovl.v12[last_exact])
    assert float((ovl.m48[last_exact] - last_before["m48"][last_exact]).abs().max()) > 0
    assert int(ovl.wave_step.min()) == ovl.step_id == 3
    kt = ovl.raster.kernel_times()
    assert kt["optimizer_early"] > 0 and kt["optimizer_early_2"] > 0 and kt["optimizer"] > 0


def test_data_parallel_side_stream_at_scale():
    """The data-parallel form of the same idea on one rank (no collectives: the choreography of torch streams, events and the
    handle-free kernels is what is under test): gut_mark_walked_waves -> gut_adam_unwalked_waves on a side stream under the
    backward, records + scatter + gut_sh_adam_step_ex(flags) on the main stream.  Rows of the waves the forward walked nothing
    of must be bit-identical to the fused one-pass step, the others equal up to float-atomic noise, over three views."""
    from tests.test_gpu_native import _rows_in_unwalked_waves
    n = 1_500_000
    sc = scenes.scene_outdoor_like(n=n, seed=2)
    ro, rd = cams.pinhole_rays(W, H, FX, FX)
    K = cams.pinhole_intrinsics_dict(W, H, FX, FX)
    gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(4)).to(DEV)
    steppers = []
    for kw in (dict(overlap_optimizer=False), dict(fuse_epilogue=False, dp_exchange="sparse", dp_side_stream=True)):
        model = native.NativeGaussianModel(sc, device=DEV, spatial_order=True)
        steppers.append(native.NativeTrainStep(model, gut.Tracer({"render": {}}), scene_extent=5.0, **kw))
    ref, dps = steppers
    for st in steppers:
        g = torch.Generator(device=DEV).manual_seed(7)
        for m_, v_ in ((st.m12, st.v12), (st.m48, st.v48)):
            m_.normal_(0.0, 1e-6, generator=g)
            v_.fill_(1e-8)
    state = lambda st: dict(raw=st.model.raw, features=st.model.features, m12=st.m12, v12=st.v12, m48=st.m48, v48=st.v48, act=st.act)
    for k in range(3):
        for name, t in state(dps).items():
            t.copy_(state(ref)[name])
        c2w = cams.orbit_c2w(4.5, 7.0 + 45.0 * k, 12.0)
        for st in (ref, dps):
            st.step(gut.Batch(rays_ori=torch.as_tensor(ro, device=DEV), rays_dir=torch.as_tensor(rd, device=DEV),
                              T_to_world=torch.as_tensor(c2w)[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K))
        torch.cuda.synchronize()
        exact = _rows_in_unwalked_waves(dps.raster, n)
        assert torch.equal(exact, _rows_in_unwalked_waves(ref.raster, n)) and int(exact.sum()) > n // 2
        assert torch.equal(dps.wave_flags.bool().repeat_interleave(64)[:n], ~exact)
        assert not bool(dps.g12.any()) and not bool(dps.mrgb[0].any())
        for name, t in state(dps).items():
            r = state(ref)[name]
            assert torch.equal(r[exact], t[exact]), f"step {k}: {name} (rows of waves no view walked)"
            a, b = r[~exact], t[~exact]
            if name in ("m12", "m48", "v12", "v48"):
                tol = 1e-5 * float(a.abs().max()) + 1e-12
                assert float((a - b).abs().max()) <= 2 * tol, f"step {k}: {name} (rows of the walked waves)"
            else:
                differs = (a - b).abs() > 1e-6 + 1e-5 * a.abs()
                assert float(differs.float().mean()) < 2e-3, f"step {k}: {name}: {float(differs.float().mean())} of the elements differ"
