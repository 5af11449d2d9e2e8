"""GPU parity: libgut_hip.so (through the C ABI, via the Tracer/SplatRaster mirror) vs the CPU oracle.

Bar (BASELINE.json north_star): integer tile/key buffers bit-exact; colour buffers within a stated fp32
tolerance.  Tolerances used here:
  * projection floats (position, conic, extent, depth, precomputed RGB): bit-exact (shared numerics contract);
  * rgba: max |diff| <= 2e-4 (the compositor uses v_exp/v_rcp/v_rsq hardware approximations, ~1 ulp each,
    and FMA contraction; the oracle uses glibc expf without contraction);
  * hit count: may differ where a response sits within 1e-6 of the 0.0113 / 1/255 thresholds: <= 0.1 % pixels;
  * gradients: relative L2 error <= 2e-3 per parameter block against the oracle's double-accumulated sums
    (the GPU sums fp32 partials in a hardware-dependent tree/atomic order).
"""
import importlib

import numpy as np
import pytest
import torch

from tests.common import FISHEYE_DIST, ROW_FLIP_BOUND, cams, check_colour_outliers, fisheye_max_angle_edge_case, make_view, rel_l2, scenes, to_batch

pytestmark = pytest.mark.gpu

gut = importlib.import_module("3dgrut_amd")
oracle = importlib.import_module("oracle.oracle")
DEV = "cuda:0"

DIST = dict(radial=[0.05, -0.02, 0.003, 0.01, 0.002, -0.001], tangential=[0.002, -0.001], thin_prism=[0.001, 0.0, -0.0005, 0.0])

CASES = {
    # name: (scene fn, view kind, W, H, camera, kwargs)
    "c1_pinhole_128": (lambda: scenes.scene_c1(1000, 0), "pinhole", 128, 128, ((0, 0, -4), (0, 0, 0)), dict(fx=128)),
    "c1_list_intrinsics": (lambda: scenes.scene_c1(1000, 0), "pinhole_list", 128, 128, ((0, 0, -4), (0, 0, 0)), dict(fx=128)),
    "ragged_100x70": (lambda: scenes.scene_c1(700, 3), "pinhole", 100, 70, ((0.5, -0.3, -3.0), (0, 0.1, 0)), dict(fx=90, fy=95)),
    "distorted_pinhole": (lambda: scenes.scene_c1(800, 4), "pinhole", 160, 96, ((0.2, 0.1, -3.2), (0, 0, 0)), dict(fx=120, distortion=DIST)),
    "fisheye_144x96": (lambda: scenes.scene_c1(900, 5), "fisheye", 144, 96, ((0.1, 0.0, -1.5), (0, 0, 0.5)), dict()),
    # non-zero OpenCV-fisheye polynomial (cameraProjections.cuh:105-128; the COLMAP loader passes the camera's k1..k4,
    # dataset_colmap.py:165-177): the Horner chain of project_fisheye multiplies real coefficients
    "fisheye_distorted": (lambda: scenes.scene_c1(900, 12), "fisheye", 144, 96, ((0.1, 0.0, -1.5), (0, 0, 0.5)), dict(distortion=FISHEYE_DIST)),
    # ... and with a field-of-view clamp INSIDE the image (max_angle 0.55 rad against the dataset rule's 1.34): sigma points beyond the
    # cone are projected with the clamped angle and marked invalid, Gaussians straddle the cone
    "fisheye_clamped_cone": (lambda: scenes.scene_c1(900, 13), "fisheye", 144, 96, ((0.1, 0.0, -1.5), (0, 0, 0.5)),
                             dict(distortion=dict(FISHEYE_DIST, max_angle=0.55))),
    "inside_cloud": (lambda: scenes.scene_c1(1500, 6), "pinhole", 96, 96, ((0.05, 0.02, -0.1), (0, 0, 1)), dict(fx=60)),
    "dense_big_splats": (lambda: _big(), "pinhole", 64, 64, ((0, 0, -3), (0, 0, 0)), dict(fx=64)),
}


def _big():
    sc = scenes.scene_c1(400, 7)
    sc["scale"] = (sc["scale"] * 4.0).astype(np.float32)
    sc["density"] = np.clip(sc["density"] * 1.5, 0, 0.999).astype(np.float32)
    return sc


def _oracle_inputs(sc, sh_degree=3):
    """Activated parameters exactly as the GPU model produces them (oracle and GPU must see identical bits)."""
    model = gut_model(sc, sh_degree)
    with torch.no_grad():
        d12 = torch.cat([model.positions, model.get_density(), model.get_rotation(), model.get_scale(),
                         torch.zeros_like(model.get_density())], 1).cpu().numpy()
        return model, d12, model.get_features().cpu().numpy()


def _run_gpu(sc, view, sh_degree, rgba_grad=None, dist_grad=None, timings=False, model=None, lazy=None, render_conf=None):
    model = model if model is not None else gut_model(sc, sh_degree)
    tr = gut.Tracer({"render": dict(render_conf or {}, enable_kernel_timings=timings)})
    if lazy is not None:
        tr.tracer_wrapper.set_lazy_tile_order(lazy)
    batch = to_batch(view, DEV)
    out = tr.render(model, batch, train=True, frame_id=0)
    res = dict(out=out, tracer=tr, model=model)
    with torch.no_grad():  # the exact activated tensors the tracer consumed (exp(log(s)) != s in fp32)
        res["density12"] = torch.cat([model.positions, model.get_density(), model.get_rotation(), model.get_scale(),
                                      torch.zeros_like(model.get_density())], 1).cpu().numpy()
        res["sph48"] = model.get_features().cpu().numpy()
    if rgba_grad is not None:
        rg = torch.as_tensor(rgba_grad, device=DEV)
        loss = (out["pred_rgb"][0] * rg[..., :3]).sum() + (out["pred_opacity"][0] * rg[..., 3:]).sum()
        if dist_grad is not None:
            loss = loss + (out["pred_dist"][0] * torch.as_tensor(dist_grad, device=DEV)).sum()
        loss.backward()
    return res


def gut_model(sc, sh_degree):
    m = importlib.import_module("3dgrut_amd.model")
    return m.GaussianModel(sc, device=DEV, sh_degree=sh_degree)


def _activated_grads(model, dens_g, sph_g):
    """Oracle grads are w.r.t. the activated tracer inputs; chain them through the same activations in float64
    so they are comparable with the nn.Parameter grads autograd produced on the GPU."""
    with torch.enable_grad():
        raw = {k: getattr(model, k).detach().double().cpu().requires_grad_(True)
               for k in ("positions", "rotation", "scale", "density", "features_albedo", "features_specular")}
        rot = torch.nn.functional.normalize(raw["rotation"], dim=1)
        scl = torch.exp(raw["scale"])
        dns = torch.sigmoid(raw["density"])
        feats = torch.cat([raw["features_albedo"], raw["features_specular"]], 1)
        dg = torch.as_tensor(dens_g)
        tot = (raw["positions"] * dg[:, 0:3]).sum() + (dns * dg[:, 3:4]).sum() + (rot * dg[:, 4:8]).sum() + \
              (scl * dg[:, 8:11]).sum() + (feats * torch.as_tensor(sph_g)).sum()
        tot.backward()
    return {k: v.grad.numpy() for k, v in raw.items()}


def _check_ordered_ids(raster, ref):
    """What the compositors actually walked (the product path orders every tile lazily, chunk by chunk) must be the prefix
    of the reference's fully sorted list: identical ids on every position the forward staged, padding ids elsewhere, and
    every tile staged exactly the 256-entry chunks its traversal depth needs."""
    ordered = raster.debug_buffer("ordered_ids").cpu().numpy().view(np.uint32)
    written = ordered != 0xFFFFFFFF
    assert np.array_equal(ordered[written], ref["sorted_ids"][written])
    trav = raster.debug_buffer("tile_traversed_fwd").cpu().numpy().view(np.uint32)
    for t, (b, e) in enumerate(ref["tile_ranges"]):
        if e > b:
            staged = int(written[b:e].sum())
            assert written[b:b + staged].all()                              # a prefix
            assert staged >= min(int(trav[t]), e - b) and (staged % 256 == 0 or staged == e - b)


@pytest.mark.parametrize("name", list(CASES))
def test_forward_buffers_and_image(name):
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    sh = 3
    model, d12, sph = _oracle_inputs(sc, sh)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=sh)
    res = _run_gpu(sc, view, sh, model=model)
    raster = res["tracer"].tracer_wrapper
    st = raster.stats()
    assert st["num_intersections"] == ref["M"]
    assert st["num_visible"] == int((ref["tiles_count"] > 0).sum())
    assert st["sort_end_bit"] == ref["end_bit"]
    # integer structure: exact
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, ref[key]), key
    for key in ("unsorted_keys", "sorted_keys"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ref[key]), key
    got = raster.debug_buffer("tile_ranges").cpu().numpy().view(np.uint32).reshape(-1, 2)
    assert np.array_equal(got, ref["tile_ranges"])
    _check_ordered_ids(raster, ref)
    # projection floats: bit-exact under the shared numerics contract
    for key, refkey in (("proj_pos", "proj_pos"), ("conic_opacity", "conic_opacity"), ("extent", "extent"), ("depth", "depth"),
                        ("feat", "feat")):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        exp = np.ascontiguousarray(ref[refkey]).reshape(-1).view(np.uint32)
        assert np.array_equal(got, exp), f"{key}: {(got != exp).sum()} of {got.size} words differ"
    vis = res["out"]["mog_visibility"].detach().cpu().numpy()[:, 0]
    assert np.array_equal(vis > 0, ref["visibility"] > 0)
    # image: colour, opacity and hit distance within COLOUR_TOL and the hit count equal on every pixel except where the oracle's own
    # decision margins say a hit / no-hit or termination decision of that ray may flip between two fp32 evaluations
    rgb = res["out"]["pred_rgb"][0].detach().cpu().numpy()
    op = res["out"]["pred_opacity"][0].detach().cpu().numpy()
    d = res["out"]["pred_dist"][0].detach().cpu().numpy()
    hits = res["out"]["hits_count"][0].detach().cpu().numpy()
    margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, budget_bound=ROW_FLIP_BOUND)
    rep = check_colour_outliers(np.concatenate([rgb, op], -1), hits, ref, margins, label=name, dist_gpu=d, budget=pixel_budget)
    assert rep["outliers"] <= 1e-3 * rep["pixels"]
    assert st["traversed_fwd"] == ref["traversed_fwd"]


@pytest.mark.parametrize("name", ["c1_pinhole_128", "ragged_100x70", "fisheye_144x96", "fisheye_distorted", "fisheye_clamped_cone",
                                  "dense_big_splats", "inside_cloud"])
@pytest.mark.parametrize("with_dist_grad", [False, True])
def test_backward_gradients(name, with_dist_grad):
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rng = np.random.default_rng(11)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    dist_grad = (0.1 * rng.normal(size=(H, W, 1))).astype(np.float32) if with_dist_grad else None
    model0, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    dens_g, sph_g, feat_g = oracle.backward(view["oracle_cam"], ref, rgba_grad,
                                            dist_grad if with_dist_grad else np.zeros((H, W, 1), np.float32))
    res = _run_gpu(sc, view, 3, rgba_grad=rgba_grad, dist_grad=dist_grad, model=model0)
    model = res["model"]
    exp = _activated_grads(model, dens_g, sph_g)
    for k, e in exp.items():
        g = getattr(model, k).grad.cpu().numpy()
        err = rel_l2(g, e)
        assert err <= 2e-3, f"{name}/{k}: rel L2 {err}"
    st = res["tracer"].tracer_wrapper.stats()
    assert st["traversed_bwd"] == ref["traversed_bwd"]


@pytest.mark.parametrize("sh", [0, 1, 2])
def test_lower_sh_degrees(sh):
    sc = scenes.scene_c1(600, 9)
    view = make_view("pinhole", 80, 64, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=80)
    W, H = 80, 64
    rgba_grad = np.random.default_rng(1).normal(size=(H, W, 4)).astype(np.float32)
    model0, d12, sph = _oracle_inputs(sc, sh)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=sh)
    dens_g, sph_g, _ = oracle.backward(view["oracle_cam"], ref, rgba_grad, np.zeros((H, W, 1), np.float32))
    res = _run_gpu(sc, view, sh, rgba_grad=rgba_grad, model=model0)
    rgb = res["out"]["pred_rgb"][0].detach().cpu().numpy()
    assert np.abs(rgb - ref["rgba"][..., :3]).max() <= 2e-4
    exp = _activated_grads(res["model"], dens_g, sph_g)
    g = res["model"].features_specular.grad.cpu().numpy()
    nc = (sh + 1) ** 2
    assert np.all(g[:, 3 * (nc - 1):] == 0), "coefficients above the active degree must get zero gradient"
    assert rel_l2(res["model"].features_albedo.grad.cpu().numpy(), exp["features_albedo"]) <= 2e-3
    if sh:
        assert rel_l2(g, exp["features_specular"]) <= 2e-3


def test_empty_and_culled_scenes():
    view = make_view("pinhole", 64, 48, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=64)
    raster = gut.SplatRaster({"render": {}})
    sensor, poses = gut.Tracer.create_camera_parameters(to_batch(view, DEV))
    ro = torch.as_tensor(view["ro"], device=DEV); rd = torch.as_tensor(view["rd"], device=DEV)
    # N = 0
    rgba, dist, hits, vis = raster.trace(0, 3, torch.zeros((0, 12), device=DEV), torch.zeros((0, 48), device=DEV), ro, rd, None,
                                         sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1])
    assert float(rgba.abs().max()) == 0 and float((dist - 1e6).abs().max()) == 0 and float(hits.abs().max()) == 0 and vis.shape == (0, 1)
    # everything behind the camera: M = 0 -> the reference returns before rendering, outputs keep their initial values
    sc = scenes.scene_c1(200, 1)
    sc["positions"][:, 2] -= 20.0
    d12 = torch.as_tensor(scenes.pack_density(sc), device=DEV); sph = torch.as_tensor(sc["features"], device=DEV)
    rgba, dist, hits, vis = raster.trace(0, 3, d12, sph, ro, rd, None, sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1])
    assert raster.stats()["num_intersections"] == 0
    assert float(rgba.abs().max()) == 0 and float(hits.abs().max()) == 0 and float(vis.abs().max()) == 0
    assert float((dist - 1e6).abs().max()) == 0
    dg, sg = raster.trace_bwd(0, 3, d12, sph, ro, rd, None, sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1],
                              rgba, torch.ones_like(rgba), dist, torch.zeros_like(dist))
    assert float(dg.abs().max()) == 0 and float(sg.abs().max()) == 0


def test_error_behaviour():
    raster = gut.SplatRaster({"render": {}})
    view = make_view("pinhole", 32, 32, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=32)
    sensor, poses = gut.Tracer.create_camera_parameters(to_batch(view, DEV))
    ro = torch.as_tensor(view["ro"], device=DEV); rd = torch.as_tensor(view["rd"], device=DEV)
    d12 = torch.zeros((4, 12), device=DEV); sph = torch.zeros((4, 48), device=DEV)
    with pytest.raises(RuntimeError):  # backward before any forward
        raster.trace_bwd(0, 3, d12, sph, ro, rd, None, sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1],
                         torch.zeros((32, 32, 4), device=DEV), torch.zeros((32, 32, 4), device=DEV),
                         torch.zeros((32, 32, 1), device=DEV), torch.zeros((32, 32, 1), device=DEV))
    with pytest.raises(RuntimeError):  # wrong dtype (voidDataPtr throws in the reference, splatRaster.cpp:86)
        raster.trace(0, 3, d12.double(), sph, ro, rd, None, sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1])
    with pytest.raises(RuntimeError):  # unsupported variant is rejected, not silently rendered
        gut.SplatRaster({"render": {"splat": {"k_buffer_size": 17}}})
    # the data-parallel helpers need their context too
    import ctypes as C
    lib = raster._lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flags = torch.zeros(4, dtype=torch.uint8, device=DEV)
    assert lib.gut_mark_walked_waves(raster._handle, st, flags.data_ptr()) != 0       # no forward yet
    rec = torch.zeros((4, 16), device=DEV); cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="no backward context"):
        raster.compact_gradient_rows(d12, rec, cnt)
    with pytest.raises(ValueError):
        raster.compact_gradient_rows(d12, torch.zeros((3, 16), device=DEV), cnt)       # fewer record slots than particles
    with pytest.raises(RuntimeError):
        raster.set_early_extra_percent(101)
    sensor.cam.shutter = 7  # not a ShutterType
    with pytest.raises(RuntimeError):
        raster.trace(0, 3, d12, sph, ro, rd, None, sensor, 0, 1, poses.T_world_sensors[0], poses.T_world_sensors[1])


# every render.* / render.splat.* key gut_create ACCEPTS away from its render/3dgut.yaml default (the reference turns them into -D
# defines, setup_3dgut.py:47-70): conf -> (oracle parameter overrides).  What gut_create rejects is tested in test_error_behaviour
# and tests/test_cpu_host.py.
TOGGLES = {
    "global_z_order_off": ({"splat": {"global_z_order": False}}, dict(global_z_order=0)),                  # gutProjector.cuh:315-321
    "rect_bounding_off": ({"splat": {"rect_bounding": False}}, dict(rect_bounding=0)),                     # :110-114
    "tight_opacity_bounding_off": ({"splat": {"tight_opacity_bounding": False}}, dict(tight_opacity_bounding=0)),   # :100-108
    "tile_based_culling_off": ({"splat": {"tile_based_culling": False}}, dict(tile_culling=0)),            # :279-293, :361-375
    "all_bounding_and_culling_off": ({"splat": {"rect_bounding": False, "tight_opacity_bounding": False, "tile_based_culling": False,
                                                "global_z_order": False}},
                                     dict(rect_bounding=0, tight_opacity_bounding=0, tile_culling=0, global_z_order=0)),
    "ut_half_alpha": ({"splat": {"ut_alpha": 0.5, "ut_beta": 1.0, "ut_kappa": 1.0}}, dict(ut_alpha=0.5, ut_beta=1.0, ut_kappa=1.0)),   # :150-201
    "ut_wide_spread_and_margin": ({"splat": {"ut_alpha": 1.25, "ut_beta": 2.5, "ut_kappa": 0.5, "ut_in_image_margin_factor": 0.3}},
                                  dict(ut_alpha=1.25, ut_beta=2.5, ut_kappa=0.5, ut_margin=0.3)),
    "thresholds": ({"particle_kernel_min_response": 0.05, "particle_kernel_min_alpha": 0.01, "particle_kernel_max_alpha": 0.9,
                    "min_transmittance": 0.01},
                   dict(min_kernel_density=0.05, alpha_threshold=0.01, max_alpha=0.9, min_transmittance=0.01)),
    "thresholds_loose": ({"particle_kernel_min_response": 0.002, "particle_kernel_min_alpha": 0.001, "particle_kernel_max_alpha": 0.999,
                          "min_transmittance": 1e-6},
                         dict(min_kernel_density=0.002, alpha_threshold=0.001, max_alpha=0.999, min_transmittance=1e-6)),
    # the reference's generalised Gaussian kernels (GAUSSIAN_PARTICLE_KERNEL_DEGREE, threedgut.cuh:35; particleResponse<> /
    # particleResponseGrd<>, gaussianParticles.cuh:211-306): the `kGeneral` instantiations of the compositors (gut_render_general.hip)
    "kernel_degree_0_linear": ({"particle_kernel_degree": 0}, dict(kernel_degree=0)),
    "kernel_degree_1_laplacian": ({"particle_kernel_degree": 1}, dict(kernel_degree=1)),
    "kernel_degree_3": ({"particle_kernel_degree": 3}, dict(kernel_degree=3)),
    "kernel_degree_4": ({"particle_kernel_degree": 4}, dict(kernel_degree=4)),
    "kernel_degree_5": ({"particle_kernel_degree": 5}, dict(kernel_degree=5)),
    "kernel_degree_8": ({"particle_kernel_degree": 8}, dict(kernel_degree=8)),
    "kernel_degree_4_thresholds": ({"particle_kernel_degree": 4, "particle_kernel_min_response": 0.05, "particle_kernel_min_alpha": 0.01},
                                   dict(kernel_degree=4, min_kernel_density=0.05, alpha_threshold=0.01)),
    "hitcounts_off": ({"enable_hitcounts": False}, dict(enable_hitcounts=0)),                              # rayPayload.cuh:44-46,126-128
}


@pytest.mark.parametrize("toggle", list(TOGGLES))
@pytest.mark.parametrize("name", ["c1_pinhole_128", "fisheye_distorted", "dense_big_splats"])
def test_config_toggles_against_the_oracle(name, toggle):
    """The configuration keys the library accepts away from their defaults reach the kernels as run-time constants
    (gut_api.cpp: build_consts; the reference compiles a variant per configuration).  Each is checked like the default variant:
    integer buffers and projection floats bit-exact, image 2e-4, gradients 2e-3 per block, traversal counts equal — and the key must
    really change the result (the oracle's result with it differs from the default's)."""
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    conf, overrides = TOGGLES[toggle]
    prm = oracle.default_params()
    for k, v in overrides.items():
        assert hasattr(prm, k), k
        setattr(prm, k, v)
    model, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3, params=prm)
    ref0 = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    assert ref["M"] > 0
    changed = (ref["M"] != ref0["M"] or not np.array_equal(ref["sorted_keys"], ref0["sorted_keys"]) or not np.array_equal(ref["rgba"], ref0["rgba"])
               or not np.array_equal(ref["proj_pos"], ref0["proj_pos"]) or not np.array_equal(ref["extent"], ref0["extent"])
               or not np.array_equal(ref["hits"], ref0["hits"]))
    assert changed, f"{toggle} changes nothing on {name}: the case does not test it"
    rng = np.random.default_rng(17)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    dist_grad = (0.1 * rng.normal(size=(H, W, 1))).astype(np.float32)
    dens_g, sph_g, _ = oracle.backward(view["oracle_cam"], ref, rgba_grad, dist_grad, params=prm)
    tr = gut.Tracer({"render": conf})
    out = tr.render(model, to_batch(view, DEV), train=True, frame_id=0)
    raster = tr.tracer_wrapper
    st = raster.stats()
    assert st["num_intersections"] == ref["M"] and st["num_visible"] == int((ref["tiles_count"] > 0).sum())
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    for key in ("unsorted_keys", "sorted_keys"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint64), ref[key]), key
    assert np.array_equal(raster.debug_buffer("tile_ranges").cpu().numpy().view(np.uint32).reshape(-1, 2), ref["tile_ranges"])
    _check_ordered_ids(raster, ref)
    for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        exp = np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)
        assert np.array_equal(got, exp), f"{key}: {(got != exp).sum()} of {got.size} words differ"
    rgba = np.concatenate([out["pred_rgb"][0].detach().cpu().numpy(), out["pred_opacity"][0].detach().cpu().numpy()], -1)
    margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, params=prm, budget_bound=ROW_FLIP_BOUND)
    check_colour_outliers(rgba, out["hits_count"][0].detach().cpu().numpy(), ref, margins, label=f"{name}/{toggle}", budget=pixel_budget)
    assert st["traversed_fwd"] == ref["traversed_fwd"]
    rg = torch.as_tensor(rgba_grad, device=DEV)
    loss = (out["pred_rgb"][0] * rg[..., :3]).sum() + (out["pred_opacity"][0] * rg[..., 3:]).sum() + \
           (out["pred_dist"][0] * torch.as_tensor(dist_grad, device=DEV)).sum()
    loss.backward()
    for k, e in _activated_grads(model, dens_g, sph_g).items():
        err = rel_l2(getattr(model, k).grad.cpu().numpy(), e)
        assert err <= 2e-3, f"{name}/{toggle}/{k}: rel L2 {err}"
    assert raster.stats()["traversed_bwd"] == ref["traversed_bwd"]


@pytest.mark.parametrize("degree", [0, 1, 2])
@pytest.mark.parametrize("name", ["c1_pinhole_128", "fisheye_distorted"])
def test_radiance_sph_degree_below_three(name, degree):
    """render.particle_radiance_sph_degree = d < 3 (PARTICLE_RADIANCE_NUM_COEFFS = (d + 1)^2, setup_3dgut.py:48): the radiance rows and
    their gradient are [N, 3 (d+1)^2], the model keeps features_specular [N, 3 (d+1)^2 - 3] (model.py:139-154).  The reference reads
    exactly that many coefficients per particle (gaussianParticles.cuh:208-216) and evaluates the active degree <= d: the oracle is
    fed the same rows zero-extended to its 16 coefficients.  Through the whole surface (Tracer.render -> _Autograd -> backward), with
    the usual checks; the gradient of the coefficients that do not exist must be zero in the oracle."""
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    nc = (degree + 1) ** 2
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    model = importlib.import_module("3dgrut_amd.model").GaussianModel(sc, device=DEV, sh_degree=degree, max_n_features=degree)
    assert model.features_specular.shape[1] == 3 * nc - 3 and model.get_features().shape[1] == 3 * nc
    with torch.no_grad():
        d12 = torch.cat([model.positions, model.get_density(), model.get_rotation(), model.get_scale(),
                         torch.zeros_like(model.get_density())], 1).cpu().numpy()
        sph = np.zeros((d12.shape[0], 48), np.float32)
        sph[:, :3 * nc] = model.get_features().cpu().numpy()
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=degree)
    rng = np.random.default_rng(23)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    dens_g, sph_g, _ = oracle.backward(view["oracle_cam"], ref, rgba_grad, np.zeros((H, W, 1), np.float32))
    assert np.abs(sph_g[:, 3 * nc:]).max() == 0 and np.abs(sph_g[:, :3 * nc]).max() > 0
    tr = gut.Tracer({"render": {"particle_radiance_sph_degree": degree}})
    assert tr.tracer_wrapper.sph_degree == degree
    out = tr.render(model, to_batch(view, DEV), train=True, frame_id=0)
    raster = tr.tracer_wrapper
    assert raster.stats()["num_intersections"] == ref["M"] and ref["M"] > 0
    for key in ("tiles_count", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
    rgba = np.concatenate([out["pred_rgb"][0].detach().cpu().numpy(), out["pred_opacity"][0].detach().cpu().numpy()], -1)
    margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, budget_bound=ROW_FLIP_BOUND)
    check_colour_outliers(rgba, out["hits_count"][0].detach().cpu().numpy(), ref, margins, label=f"{name}/sph{degree}", budget=pixel_budget)
    rg = torch.as_tensor(rgba_grad, device=DEV)
    ((out["pred_rgb"][0] * rg[..., :3]).sum() + (out["pred_opacity"][0] * rg[..., 3:]).sum()).backward()
    assert model.features_specular.grad is None or model.features_specular.grad.shape == model.features_specular.shape
    for k, e in _activated_grads(model, dens_g, sph_g[:, :3 * nc]).items():
        if e.size == 0:
            continue
        err = rel_l2(getattr(model, k).grad.cpu().numpy(), e)
        assert err <= 2e-3, f"{name}/sph{degree}/{k}: rel L2 {err}"
    # one more active degree than the handle has coefficients for is refused (the reference would read past its coefficient array)
    model.n_active_features = degree + 1
    with pytest.raises(RuntimeError, match="active SH degrees"):
        tr.render(model, to_batch(view, DEV), train=True, frame_id=1)
    # ... and so are the entry points that are laid out for 16 coefficients
    batch = to_batch(view, DEV)
    sensor, poses = gut.Tracer.create_camera_parameters(batch)
    with pytest.raises(RuntimeError, match="particle_radiance_sph_degree"):
        raster.trace_model_fields(0, degree, model.positions.detach(), model.get_density().detach(), model.get_rotation().detach(),
                                  model.get_scale().detach(), model.features_albedo.detach(), torch.zeros((d12.shape[0], 45), device=DEV),
                                  batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), sensor, 0, 1, poses.T_world_sensors[0],
                                  poses.T_world_sensors[1])


@pytest.mark.parametrize("ulps_above", [0, 1])
def test_fisheye_theta_equal_to_max_angle(ulps_above):
    """SURVEY §8c KAT on the device: OpenCV fisheye with non-zero polynomial coefficients and max_angle == the fp32 atan2f of a
    degenerate opaque Gaussian's direction (tests/common.fisheye_max_angle_edge_case; cameraProjections.cuh:119,127).  At equality
    the Gaussian gets no tile, one ulp above it gets one — on the GPU exactly as in the oracle — and every integer buffer and
    projection float of the frame (the cone cuts through the other Gaussians) is bit-identical."""
    sc, view, theta = fisheye_max_angle_edge_case(ulps_above)
    W, H = view["W"], view["H"]
    model, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    assert ref["tiles_count"][0] == ulps_above and ref["visibility"][0] == ulps_above
    res = _run_gpu(sc, view, 3, model=model)
    raster = res["tracer"].tracer_wrapper
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    for key in ("unsorted_keys", "sorted_keys"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint64), ref[key]), key
    for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
    vis = res["out"]["mog_visibility"].detach().cpu().numpy()[:, 0]
    assert np.array_equal(vis > 0, ref["visibility"] > 0) and bool(vis[0] > 0) == bool(ulps_above)
    rgb = res["out"]["pred_rgb"][0].detach().cpu().numpy()
    assert np.abs(rgb - ref["rgba"][..., :3]).max() <= 2e-4


@pytest.mark.parametrize("shutter", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["pinhole", "fisheye", "fisheye_distorted"])
def test_rolling_shutter_projection(shutter, kind):
    """projectPointWithShutter<5> with distinct start/end poses (cameraProjections.cuh:146-185): tile/key buffers and
    projection floats stay bit-exact (slerp uses the shared deterministic acos/sin), image within tolerance."""
    pose_mod = importlib.import_module("3dgrut_amd.pose")
    sc = scenes.scene_c1(700, 40 + shutter)
    W, H = 96, 80
    distortion = FISHEYE_DIST if kind == "fisheye_distorted" else None
    kind = "fisheye" if distortion else kind
    view = make_view(kind, W, H, cams.look_at_c2w((0.1, 0.0, -3.0 if kind == "pinhole" else -1.6), (0, 0, 0)), fx=90 if kind == "pinhole" else None,
                     distortion=distortion)
    end_c2w = cams.look_at_c2w((0.25, -0.1, -2.9 if kind == "pinhole" else -1.55), (0.05, 0.0, 0.0))
    tq_end = pose_mod.sensor_pose_from_c2w(end_c2w).T_world_sensors[0]
    ocam = dict(view["oracle_cam"], shutter=shutter, pose_end=tq_end)
    model, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(ocam, W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    raster = gut.SplatRaster({"render": {}})
    sensor, poses = gut.Tracer.create_camera_parameters(to_batch(view, DEV))
    sensor.cam.shutter = shutter
    ro = torch.as_tensor(view["ro"], device=DEV); rd = torch.as_tensor(view["rd"], device=DEV)
    rgba, dist, hits, vis = raster.trace(0, 3, torch.as_tensor(d12, device=DEV), torch.as_tensor(sph, device=DEV), ro, rd, None, sensor,
                                         0, 1, poses.T_world_sensors[0], tq_end)
    assert raster.stats()["num_intersections"] == ref["M"] and ref["M"] > 0
    for key in ("tiles_count", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    assert np.array_equal(raster.debug_buffer("sorted_keys").cpu().numpy().view(np.uint64), ref["sorted_keys"])
    for key in ("proj_pos", "conic_opacity", "extent", "depth"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
    assert np.abs(rgba.cpu().numpy() - ref["rgba"]).max() <= 2e-4
    # and the rolling-shutter result must actually differ from the global-shutter one
    glob = oracle.forward(dict(view["oracle_cam"]), W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    assert not np.array_equal(glob["proj_pos"], ref["proj_pos"])


@pytest.mark.parametrize("iterations", [0, 2, 9])
@pytest.mark.parametrize("kind", ["pinhole", "fisheye_distorted"])
def test_rolling_shutter_iteration_count(iterations, kind):
    """render.splat.n_rolling_shutter_iterations away from 5 (GAUSSIAN_N_ROLLING_SHUTTER_ITERATIONS, the trip count of
    projectPointWithShutter's refinement loop, cameraProjections.cuh:174): a run-time value here.  Same checks as the default count,
    and a low count must really move the projection."""
    pose_mod = importlib.import_module("3dgrut_amd.pose")
    sc = scenes.scene_c1(700, 47)
    W, H = 96, 80
    distortion = FISHEYE_DIST if kind == "fisheye_distorted" else None
    kind = "fisheye" if distortion else kind
    view = make_view(kind, W, H, cams.look_at_c2w((0.1, 0.0, -3.0 if kind == "pinhole" else -1.6), (0, 0, 0)), fx=90 if kind == "pinhole" else None,
                     distortion=distortion)
    end_c2w = cams.look_at_c2w((0.25, -0.1, -2.9 if kind == "pinhole" else -1.55), (0.05, 0.0, 0.0))
    tq_end = pose_mod.sensor_pose_from_c2w(end_c2w).T_world_sensors[0]
    ocam = dict(view["oracle_cam"], shutter=0, pose_end=tq_end)
    model, d12, sph = _oracle_inputs(sc, 3)
    prm = oracle.default_params()
    prm.rolling_shutter_iterations = iterations
    ref = oracle.forward(ocam, W, H, d12, sph, view["ro"], view["rd"], sh_degree=3, params=prm)
    ref5 = oracle.forward(ocam, W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    if iterations < 5:
        assert not np.array_equal(ref["proj_pos"], ref5["proj_pos"])
    raster = gut.SplatRaster({"render": {"splat": {"n_rolling_shutter_iterations": iterations}}})
    sensor, poses = gut.Tracer.create_camera_parameters(to_batch(view, DEV))
    sensor.cam.shutter = 0
    ro = torch.as_tensor(view["ro"], device=DEV); rd = torch.as_tensor(view["rd"], device=DEV)
    rgba, dist, hits, vis = raster.trace(0, 3, torch.as_tensor(d12, device=DEV), torch.as_tensor(sph, device=DEV), ro, rd, None, sensor,
                                         0, 1, poses.T_world_sensors[0], tq_end)
    assert raster.stats()["num_intersections"] == ref["M"] and ref["M"] > 0
    for key in ("tiles_count", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    assert np.array_equal(raster.debug_buffer("sorted_keys").cpu().numpy().view(np.uint64), ref["sorted_keys"])
    for key in ("proj_pos", "conic_opacity", "extent", "depth"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
    assert np.abs(rgba.cpu().numpy() - ref["rgba"]).max() <= 2e-4


def test_host_count_follows_the_device_count_every_frame():
    """The intersection count reaches the host through system-scope stores of the scan kernel into coherent pinned memory and an event
    WITHOUT a system-scope fence (gut_api.cpp: kOrderingEvent, k_scan_wave_sums).  300 frames over four views with different counts (275 k - 447 k):
    the host's count of every frame (stats: what sizes the binning buffers and detects an overflow) must be that frame's device
    count, never the previous frame's."""
    sc = scenes.scene_c1(60000, 12)
    W, H = 160, 120
    views = [make_view("pinhole", W, H, cams.look_at_c2w(eye, (0, 0, 0)), fx=f) for eye, f in
             (((0.0, 0.0, -4.0), 150.0), ((0.3, 0.2, -1.2), 60.0), ((2.5, 0.1, -0.4), 220.0), ((0.0, 0.1, -9.0), 400.0))]
    model = gut_model(sc, 3)
    tr = gut.Tracer({"render": {}})
    raster = tr.tracer_wrapper
    batches = [to_batch(v, DEV) for v in views]
    seen = set()
    with torch.no_grad():
        for k in range(300):
            tr.render(model, batches[(k * 7 + k // 5) % len(batches)], train=False, frame_id=k)
            m_host = raster.stats()["num_intersections"]
            m_dev = int(raster.debug_buffer("tiles_count").view(torch.int32).sum().item())
            assert m_host == m_dev, f"frame {k}: host {m_host}, device {m_dev}"
            seen.add(m_host)
    assert len(seen) == len(views) and max(seen) > 1.5 * min(seen)   # every view has its own count: a stale one would have shown


def test_timings_surface():
    sc = scenes.scene_c1(500, 2)
    view = make_view("pinhole", 64, 64, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=64)
    res = _run_gpu(sc, view, 3, rgba_grad=np.ones((64, 64, 4), np.float32), timings=True)
    assert res["out"]["frame_time_ms"] > 0
    t = res["tracer"].timings
    assert t["forward_render"] > 0 and t["backward_render"] > 0
    kt = res["tracer"].tracer_wrapper.kernel_times()
    assert all(kt[k] >= 0 for k in ("project", "sort", "render", "render_bwd", "project_bwd"))


# ---------------------------------------------------------------------------------------------------
# sorted variant (render.splat.k_buffer_size > 0), SURVEY §8a row a14
# ---------------------------------------------------------------------------------------------------
def _run_sorted(view, model, K, rgba_grad=None, dist_grad=None, reference_backward=None, kernel_degree=2):
    splat = {"k_buffer_size": K}
    if reference_backward is not None:          # None: the library's default (the reference's form since ABI 5)
        splat["sorted_reference_backward"] = reference_backward
    tr = gut.Tracer({"render": {"splat": splat, "particle_kernel_degree": kernel_degree}})
    out = tr.render(model, to_batch(view, DEV), train=True, frame_id=0)
    if rgba_grad is not None:
        rg = torch.as_tensor(rgba_grad, device=DEV)
        loss = (out["pred_rgb"][0] * rg[..., :3]).sum() + (out["pred_opacity"][0] * rg[..., 3:]).sum()
        if dist_grad is not None:
            loss = loss + (out["pred_dist"][0] * torch.as_tensor(dist_grad, device=DEV)).sum()
        loss.backward()
    return out


@pytest.mark.parametrize("name", ["c1_pinhole_128", "ragged_100x70", "fisheye_144x96", "dense_big_splats", "inside_cloud"])
@pytest.mark.parametrize("K", [1, 4, 16])
def test_sorted_variant_forward(name, K):
    """k-buffer compositing vs oracle_render_kbuffer on identical tile lists.  Tolerances: as the unsorted image test;
    the hit order can flip where two hit distances agree to ~1 ulp (hardware rcp/rsq vs libm), so allow 0.1 % of
    pixels to exceed the colour tolerance."""
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    model, d12, sph = _oracle_inputs(sc, 3)
    fwd = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    ref = oracle.render_kbuffer(view["oracle_cam"], fwd, K=K)
    out = _run_sorted(view, model, K)
    rgb = out["pred_rgb"][0].detach().cpu().numpy()
    op = out["pred_opacity"][0].detach().cpu().numpy()
    d = out["pred_dist"][0].detach().cpu().numpy()
    hits = out["hits_count"][0].detach().cpu().numpy()
    bad = (np.abs(rgb - ref["rgba"][..., :3]).max(-1) > 2e-4) | (np.abs(op - ref["rgba"][..., 3:]).max(-1) > 2e-4) | \
          (np.abs(d - ref["dist"]).max(-1) > 2e-4 * max(1.0, float(np.abs(ref["dist"]).max())))
    assert bad.mean() <= 1e-3, f"{bad.sum()} of {bad.size} pixels differ"
    assert (hits != ref["hits"]).mean() <= 2e-3
    if K == 1 and name == "c1_pinhole_128":
        # K=1 composites in list order: must agree with the unsorted compositor
        ref0 = _run_gpu(sc, view, 3, model=model)["out"]
        assert (out["pred_rgb"] - ref0["pred_rgb"]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("degree", [0, 1, 4])
def test_sorted_variant_with_generalised_kernels(degree):
    """k_buffer_size > 0 together with particle_kernel_degree != 2: forward against oracle_render_kbuffer, backward against float64
    autograd of the ordered composite — the TRUE derivative of the response, which is what slang autodiff gives the reference's sorted
    variant (gaussianParticles.slang:119-164), also for degree 1 where the unsorted CUDA backward has its own form."""
    prt = importlib.import_module("oracle.per_ray_torch")
    K = 8
    mk, kind, W, H, (eye, tgt), kw = CASES["c1_pinhole_128"]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rng = np.random.default_rng(15)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    model, d12, sph = _oracle_inputs(sc, 3)
    prm = oracle.default_params()
    prm.kernel_degree = degree
    fwd = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3, params=prm)
    max_order = int(fwd["hits"].max()) + 64
    ref = oracle.render_kbuffer(view["oracle_cam"], fwd, K=K, max_order=max_order, params=prm)
    L = max(int(ref["order_count"].max()), 1)
    params = dict(positions=d12[:, 0:3], density=d12[:, 3:4], rotation=d12[:, 4:8], scale=d12[:, 8:11], features=sph)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    rgba, dist = prt.composite_ordered(params, view["tq"], W, H, view["ro"], view["rd"], ref["order_ids"][:, :L], ref["order_count"],
                                       kernel_degree=degree)
    assert np.abs(rgba.detach().numpy().reshape(H, W, 4) - ref["rgba"]).max() <= 5e-5
    (rgba * torch.tensor(rgba_grad.reshape(-1, 4), dtype=torch.float64)).sum().backward()
    dens_g = np.zeros((d12.shape[0], 12))
    dens_g[:, 0:3] = params["positions"].grad.numpy(); dens_g[:, 3:4] = params["density"].grad.numpy()
    dens_g[:, 4:8] = params["rotation"].grad.numpy(); dens_g[:, 8:11] = params["scale"].grad.numpy()
    exp = _activated_grads(model, dens_g, params["features"].grad.numpy())
    out = _run_sorted(view, model, K, rgba_grad=rgba_grad, reference_backward=False, kernel_degree=degree)
    rgb = out["pred_rgb"][0].detach().cpu().numpy()
    bad = np.abs(rgb - ref["rgba"][..., :3]).max(-1) > 2e-4
    assert bad.mean() <= 1e-3, f"{bad.sum()} of {bad.size} pixels differ"
    assert (out["hits_count"][0].detach().cpu().numpy() != ref["hits"]).mean() <= 2e-3
    for k, e in exp.items():
        err = rel_l2(getattr(model, k).grad.cpu().numpy(), e)
        assert err <= (2e-3 if degree else 5e-3), f"degree {degree}/{k}: rel L2 {err}"


@pytest.mark.parametrize("name", ["c1_pinhole_128", "fisheye_144x96", "dense_big_splats"])
@pytest.mark.parametrize("with_dist_grad", [False, True])
def test_sorted_variant_backward(name, with_dist_grad):
    """Gradients of the sorted variant vs float64 autograd through the exact ordered compositing
    (oracle/per_ray_torch.composite_ordered on the per-pixel order recorded by the C oracle).  Tolerance 2e-3 relative
    L2 per parameter block, as for the unsorted backward."""
    prt = importlib.import_module("oracle.per_ray_torch")
    K = 8
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rng = np.random.default_rng(5)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    dist_grad = (0.1 * rng.normal(size=(H, W, 1))).astype(np.float32) if with_dist_grad else None
    model, d12, sph = _oracle_inputs(sc, 3)
    fwd = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    max_order = int(fwd["hits"].max()) + 64
    ref = oracle.render_kbuffer(view["oracle_cam"], fwd, K=K, max_order=max_order)
    L = int(ref["order_count"].max())
    assert L <= max_order
    params = dict(positions=d12[:, 0:3], density=d12[:, 3:4], rotation=d12[:, 4:8], scale=d12[:, 8:11], features=sph)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    rgba, dist = prt.composite_ordered(params, view["tq"], W, H, view["ro"], view["rd"], ref["order_ids"][:, :max(L, 1)],
                                       ref["order_count"])
    assert np.abs(rgba.detach().numpy().reshape(H, W, 4) - ref["rgba"]).max() <= 5e-5
    loss = (rgba * torch.tensor(rgba_grad.reshape(-1, 4), dtype=torch.float64)).sum()
    if with_dist_grad:
        loss = loss + (dist * torch.tensor(dist_grad.reshape(-1), dtype=torch.float64)).sum()
    loss.backward()
    dens_g = np.zeros((d12.shape[0], 12))
    dens_g[:, 0:3] = params["positions"].grad.numpy()
    dens_g[:, 3:4] = params["density"].grad.numpy()
    dens_g[:, 4:8] = params["rotation"].grad.numpy()
    dens_g[:, 8:11] = params["scale"].grad.numpy()
    exp = _activated_grads(model, dens_g, params["features"].grad.numpy())
    _run_sorted(view, model, K, rgba_grad=rgba_grad, dist_grad=dist_grad, reference_backward=False)   # the exact-derivative option
    for k, e in exp.items():
        g = getattr(model, k).grad.cpu().numpy()
        err = rel_l2(g, e)
        assert err <= 2e-3, f"{name}/{k}: rel L2 {err}"


@pytest.mark.parametrize("name", ["c1_pinhole_128", "dense_big_splats"])
def test_sorted_variant_reference_backward(name):
    """GUT_OPT_SORTED_REFERENCE_BACKWARD: the reference's own form of the sorted backward (un-do recurrence with the UNCLAMPED
    colour, gutKBufferRenderer.cuh:127-131 / shRadiativeParticles.slang:179-207) against its restatement in
    oracle/per_ray_torch (_ReferenceUndoColour) on the per-pixel order recorded by the C oracle.  The scene is given negative
    SH dc terms on a third of the Gaussians so that the two forms really differ; the test also checks that they do, and that
    the option changes nothing on a scene without negative colours."""
    prt = importlib.import_module("oracle.per_ray_torch")
    K = 8
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    sc["features"] = sc["features"].copy()
    sc["features"][::3, 0:3] -= 2.5      # colour = 0.282 * dc + 0.5 + ... < 0 on these
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rng = np.random.default_rng(8)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    model, d12, sph = _oracle_inputs(sc, 3)
    fwd = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    assert (fwd["feat"][fwd["tiles_count"] > 0] < 0).any()
    max_order = int(fwd["hits"].max()) + 64
    ref = oracle.render_kbuffer(view["oracle_cam"], fwd, K=K, max_order=max_order)
    L = max(int(ref["order_count"].max()), 1)
    grads = {}
    for mode in (False, True):
        params = dict(positions=d12[:, 0:3], density=d12[:, 3:4], rotation=d12[:, 4:8], scale=d12[:, 8:11], features=sph)
        params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
        rgba, dist = prt.composite_ordered(params, view["tq"], W, H, view["ro"], view["rd"], ref["order_ids"][:, :L], ref["order_count"],
                                           reference_undo_colour=mode)
        assert np.abs(rgba.detach().numpy().reshape(H, W, 4) - ref["rgba"]).max() <= 5e-5
        (rgba * torch.tensor(rgba_grad.reshape(-1, 4), dtype=torch.float64)).sum().backward()
        dens_g = np.zeros((d12.shape[0], 12))
        dens_g[:, 0:3] = params["positions"].grad.numpy(); dens_g[:, 3:4] = params["density"].grad.numpy()
        dens_g[:, 4:8] = params["rotation"].grad.numpy(); dens_g[:, 8:11] = params["scale"].grad.numpy()
        grads[mode] = _activated_grads(model, dens_g, params["features"].grad.numpy())
    # the two forms differ on the geometry / density gradients (not on the colour gradient) of this scene ...
    assert rel_l2(grads[True]["density"], grads[False]["density"]) > 1e-2
    assert rel_l2(grads[True]["features_albedo"], grads[False]["features_albedo"]) <= 1e-12
    # ... and the kernel follows whichever is selected; with nothing selected it follows the REFERENCE (ABI 5)
    for mode in (False, True, None):
        model.zero_grad(set_to_none=True)
        _run_sorted(view, model, K, rgba_grad=rgba_grad, reference_backward=mode)
        for k, e in grads[True if mode is None else mode].items():
            g = getattr(model, k).grad.cpu().numpy()
            assert rel_l2(g, e) <= 2e-3, f"{name}/{k} reference_backward={mode}: rel L2 {rel_l2(g, e)}"


def test_unusable_rays_do_not_poison_gradients():
    """Rays with NaN / inf directions never hit anything (oracle: every comparison with NaN is false).  (A ray whose direction
    is exactly zero is out of contract: safe_normalize gives the reference response 1 for every listed particle in its
    backward; here such a ray contributes nothing — DESIGN.md deviation 7.)  The backward
    evaluates non-hitting lanes with zero weights instead of masking them, so such lanes must be neutralised explicitly:
    the gradients must stay finite and equal the oracle's for the same rays."""
    mk, kind, W, H, (eye, tgt), kw = CASES["c1_pinhole_128"]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rd = view["rd"].copy()
    r3 = rd.reshape(H, W, 3)  # a view of rd
    r3[40, 50] = np.nan
    r3[41, 51] = (np.inf, 0.0, 1.0)
    r3[10:12, 100:104] = np.nan
    view = dict(view, rd=rd)
    rng = np.random.default_rng(2)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    model0, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    dens_g, sph_g, _ = oracle.backward(view["oracle_cam"], ref, rgba_grad, np.zeros((H, W, 1), np.float32))
    assert np.isfinite(dens_g).all() and np.isfinite(sph_g).all()
    res = _run_gpu(sc, view, 3, rgba_grad=rgba_grad, model=model0)
    exp = _activated_grads(res["model"], dens_g, sph_g)
    for k, e in exp.items():
        g = getattr(res["model"], k).grad.cpu().numpy()
        assert np.isfinite(g).all(), k
        assert rel_l2(g, e) <= 2e-3, f"{k}: rel L2 {rel_l2(g, e)}"


@pytest.mark.parametrize("name", ["c1_pinhole_128", "fisheye_144x96", "dense_big_splats", "inside_cloud"])
def test_lazy_tile_order_is_equivalent(name):
    """GUT_OPT_LAZY_TILE_ORDER (the default) against the full radix sort: same image,
    same traversal depths, same gradients; the walked lists are prefixes of the reference's fully sorted lists."""
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rgba_grad = np.random.default_rng(3).normal(size=(H, W, 4)).astype(np.float32)
    model, d12, sph = _oracle_inputs(sc, 3)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
    a = _run_gpu(sc, view, 3, rgba_grad=rgba_grad, model=model, lazy=False)
    ga = {k: getattr(model, k).grad.clone() for k in ("positions", "rotation", "scale", "density", "features_albedo")}
    model.zero_grad(set_to_none=True)
    b = _run_gpu(sc, view, 3, rgba_grad=rgba_grad, model=model, lazy=True)
    assert torch.equal(a["out"]["pred_rgb"], b["out"]["pred_rgb"]) and torch.equal(a["out"]["pred_dist"], b["out"]["pred_dist"])
    ra, rb = a["tracer"].tracer_wrapper, b["tracer"].tracer_wrapper
    assert torch.equal(ra.debug_buffer("tile_traversed_fwd"), rb.debug_buffer("tile_traversed_fwd"))
    assert np.array_equal(rb.debug_buffer("sorted_ids").cpu().numpy().view(np.uint32), ref["sorted_ids"])
    _check_ordered_ids(rb, ref)
    for k, g in ga.items():
        assert rel_l2(getattr(model, k).grad.cpu().numpy(), g.cpu().numpy()) <= 1e-5, k


@pytest.mark.parametrize("name", ["c1_pinhole_128", "fisheye_distorted", "dense_big_splats"])
def test_forward_tile_launch_order_is_transparent(name):
    """GUT_OPT_FORWARD_TILE_ORDER: the forward compositor launched longest-lists-first (what the library switches to by itself when
    the last frames walked more than a quarter of their lists) against image order: same image, same traversal depths, same walked
    lists, same gradients up to the float-atomic order; and the automatic mode decides from the walked share of the frame before
    the last (read back with the intersection count)."""
    mk, kind, W, H, (eye, tgt), kw = CASES[name]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    rgba_grad = np.random.default_rng(3).normal(size=(H, W, 4)).astype(np.float32)
    model = gut_model(sc, 3)
    outs = []
    for mode in (0, 1):
        tr = gut.Tracer({"render": {}})
        tr.tracer_wrapper.set_forward_tile_order(mode)
        model.zero_grad(set_to_none=True)
        out = tr.render(model, to_batch(view, DEV), train=True, frame_id=0)
        rg = torch.as_tensor(rgba_grad, device=DEV)
        ((out["pred_rgb"][0] * rg[..., :3]).sum() + (out["pred_opacity"][0] * rg[..., 3:]).sum()).backward()
        r = tr.tracer_wrapper
        outs.append(dict(rgb=out["pred_rgb"].detach().clone(), dist=out["pred_dist"].detach().clone(), hits=out["hits_count"].detach().clone(),
                         trav=r.debug_buffer("tile_traversed_fwd"), ordered=r.debug_buffer("ordered_ids"),
                         grads={k: getattr(model, k).grad.clone() for k in ("positions", "rotation", "scale", "density", "features_albedo")}))
    a, b = outs
    assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["dist"], b["dist"]) and torch.equal(a["hits"], b["hits"])
    assert torch.equal(a["trav"], b["trav"]) and torch.equal(a["ordered"], b["ordered"])
    for k in a["grads"]:
        assert rel_l2(b["grads"][k].cpu().numpy(), a["grads"][k].cpu().numpy()) <= 1e-5, k
    with pytest.raises(RuntimeError):
        gut.Tracer({"render": {}}).tracer_wrapper.set_forward_tile_order(2)


def test_forward_queued_before_the_count_is_known_overflow_and_padding():
    """The forward sizes its binning buffers from earlier frames and queues everything before this frame's intersection
    count has been read back (gut_api.cpp).  Frame 1 on a handle has nothing to size from (blocking path); a later frame
    with FEWER intersections runs over a padded list; one with MORE overflows the capacity and is transparently redone.
    All three must give the oracle's integer buffers and image."""
    sc = scenes.scene_c1(3000, 31)
    model, d12, sph = _oracle_inputs(sc, 3)
    tr = gut.Tracer({"render": {}})
    raster = tr.tracer_wrapper
    W, H = 128, 96
    cams_ = [((0, 0, -6.0), 110.0),    # frame 1: medium
             ((0, 0, -14.0), 110.0),   # far away: fewer intersections than the capacity (padded tail)
             ((0, 0, -2.2), 110.0),    # close: many more than the capacity -> overflow, binning redone
             ((0, 0, -14.0), 110.0)]   # and back again
    counts, overflows = [], []
    for eye, fx in cams_:
        view = make_view("pinhole", W, H, cams.look_at_c2w(eye, (0, 0, 0)), fx=fx)
        ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
        out = tr.render(model, to_batch(view, DEV), train=False)
        st = raster.stats()
        counts.append(st["num_intersections"]); overflows.append(st["binning_overflows"])
        assert st["num_intersections"] == ref["M"]
        for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
            assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
        for key in ("unsorted_keys", "sorted_keys"):
            assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint64), ref[key]), key
        assert np.array_equal(raster.debug_buffer("tile_ranges").cpu().numpy().view(np.uint32).reshape(-1, 2), ref["tile_ranges"])
        _check_ordered_ids(raster, ref)
        rgb = out["pred_rgb"][0].detach().cpu().numpy()
        assert np.abs(rgb - ref["rgba"][..., :3]).max() <= 2e-4
        assert st["traversed_fwd"] == ref["traversed_fwd"]
    assert counts[1] < counts[0] < counts[2]
    assert overflows == [0, 0, 1, 1], (counts, overflows)


def test_field_wise_trace_equals_the_packed_trace():
    """gut_trace_fields / gut_trace_bwd_fields (what _Autograd calls on this library's wrapper: four activated tensors in, four
    gradient tensors out, rows packed inside the library) against gut_trace / gut_trace_bwd on the torch.cat-ed [N,12] rows (the
    reference's own form, still taken for any wrapper with only the pybind surface): images and integer buffers bit-identical,
    gradients equal up to the float-atomic order of the backward, the same rows exactly zero."""
    mk, kind, W, H, (eye, tgt), kw = CASES["c1_pinhole_128"]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    model = gut_model(sc, 3)
    with torch.no_grad():
        pos, dns, rot, scl = model.positions.detach(), model.get_density(), model.get_rotation(), model.get_scale()
        sph = model.get_features()
        d12 = torch.cat([pos, dns, rot, scl, torch.zeros_like(dns)], 1).contiguous()
    batch = to_batch(view, DEV)
    sensor, poses = gut.Tracer.create_camera_parameters(batch)
    ts, ps = poses.timestamps_us, poses.T_world_sensors
    a, b = gut.Tracer({"render": {}}).tracer_wrapper, gut.Tracer({"render": {}}).tracer_wrapper
    ro, rd = batch.rays_ori.contiguous(), batch.rays_dir.contiguous()
    out_a = a.trace(0, 3, d12, sph, ro, rd, None, sensor, ts[0], ts[1], ps[0], ps[1])
    out_b = b.trace_fields(0, 3, pos, dns, rot, scl, sph, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1])
    for x, y in zip(out_a, out_b):
        assert torch.equal(x, y)
    for key in ("tiles_count", "sorted_ids", "sorted_keys", "tile_ranges"):
        assert torch.equal(a.debug_buffer(key), b.debug_buffer(key)), key
    g = torch.randn((H, W, 4), generator=torch.Generator().manual_seed(3)).to(DEV)
    dg = 0.1 * torch.randn((H, W, 1), generator=torch.Generator().manual_seed(4)).to(DEV)
    g12, g48 = a.trace_bwd(0, 3, d12, sph, ro, rd, None, sensor, ts[0], ts[1], ps[0], ps[1], out_a[0], g, out_a[1], dg)
    pg, ng, rg, sg, fg = b.trace_bwd_fields(0, 3, pos.shape[0], sph, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1], out_b[0], g, out_b[1], dg)
    assert pg.shape == (1000, 3) and ng.shape == (1000, 1) and rg.shape == (1000, 4) and sg.shape == (1000, 3) and fg.shape == (1000, 48)
    for got, ref in ((pg, g12[:, 0:3]), (ng, g12[:, 3:4]), (rg, g12[:, 4:8]), (sg, g12[:, 8:11]), (fg, g48)):
        assert torch.equal(got == 0, ref == 0)
        assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) <= 1e-5
    # a plain trace() on the handle invalidates the packed rows: the field-wise backward must not run on them
    b.trace(0, 3, d12, sph, ro, rd, None, sensor, ts[0], ts[1], ps[0], ps[1])
    with pytest.raises(RuntimeError, match="no gut_trace_fields forward"):
        b.trace_bwd_fields(0, 3, pos.shape[0], sph, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1], out_b[0], g, out_b[1], dg)


@pytest.mark.parametrize("n,degree", [(1000, 3), (1001, 3), (1002, 2), (1003, 1), (1000, 0), (63, 3), (1, 3)])
def test_model_field_trace_equals_the_field_wise_trace(n, degree):
    """gut_trace_model_fields / gut_trace_bwd_model_fields (the SH coefficients as the model's features_albedo [N,3] and
    features_specular [N,45] instead of their torch.cat) against gut_trace_fields on the concatenation: forward outputs and integer
    buffers bit-identical, gradients equal up to the float-atomic order with the same entries exactly zero.  Row counts whose
    [N,45] tensor ends inside a 16-byte word (N mod 4 != 0), a partial last wave, one wave, one row; the words behind both
    [N,45] tensors are sentinels the kernels must neither use nor overwrite."""
    mk, kind, W, H, (eye, tgt), kw = CASES["c1_pinhole_128"]
    sc = scenes.scene_c1(n, 0) if n >= 1000 else {k: v[:n] for k, v in mk().items()}
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    model = gut_model(sc, 3)
    with torch.no_grad():
        pos, dns, rot, scl = model.positions.detach(), model.get_density(), model.get_rotation(), model.get_scale()
        sph = model.get_features().contiguous()
        alb = model.features_albedo.detach().contiguous()
        spec_buf = torch.full((n * 45 + 16,), float("nan"), device=DEV)
        spec = spec_buf[: n * 45].view(n, 45)
        spec.copy_(model.features_specular.detach())
    batch = to_batch(view, DEV)
    sensor, poses = gut.Tracer.create_camera_parameters(batch)
    ts, ps = poses.timestamps_us, poses.T_world_sensors
    a, b = gut.Tracer({"render": {}}).tracer_wrapper, gut.Tracer({"render": {}}).tracer_wrapper
    ro, rd = batch.rays_ori.contiguous(), batch.rays_dir.contiguous()
    out_a = a.trace_fields(0, degree, pos, dns, rot, scl, sph, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1])
    out_b = b.trace_model_fields(0, degree, pos, dns, rot, scl, alb, spec, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1])
    for x, y in zip(out_a, out_b):
        assert torch.equal(x, y)
    for key in ("tiles_count", "sorted_ids", "sorted_keys", "tile_ranges"):
        assert torch.equal(a.debug_buffer(key), b.debug_buffer(key)), key
    assert torch.equal(a.debug_buffer("feat"), b.debug_buffer("feat"))       # the view-dependent colours K1 made of the coefficients
    g = torch.randn((H, W, 4), generator=torch.Generator().manual_seed(3)).to(DEV)
    dg = 0.1 * torch.randn((H, W, 1), generator=torch.Generator().manual_seed(4)).to(DEV)
    pg, ng, rg, sg, fg = a.trace_bwd_fields(0, degree, n, sph, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1], out_a[0], g, out_a[1], dg)
    spec_g_buf = torch.full((n * 45 + 16,), 7.0, device=DEV)
    outs = [torch.empty((n, c), device=DEV) for c in (3, 1, 4, 3, 3)] + [spec_g_buf[: n * 45].view(n, 45)]
    got = b.trace_bwd_model_fields(0, degree, n, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1], out_b[0], g, out_b[1], dg, out=outs)
    assert bool((spec_g_buf[n * 45:] == 7.0).all())                       # nothing written behind the tensor
    assert got[5].data_ptr() == spec_g_buf.data_ptr()
    if n >= 1000:
        assert int((fg != 0).any(dim=1).sum()) > 100
    for name, x, ref in (("pos", got[0], pg), ("dns", got[1], ng), ("rot", got[2], rg), ("scl", got[3], sg), ("alb", got[4], fg[:, :3]),
                         ("spec", got[5], fg[:, 3:])):
        assert x.shape == ref.shape, name
        assert torch.equal(x == 0, ref == 0), name
        assert bool(torch.isfinite(x).all()), name
        if float(ref.abs().max()) > 0:
            assert rel_l2(x.cpu().numpy(), ref.contiguous().cpu().numpy()) <= 1e-5, name
    if degree == 0:
        assert not bool(got[5].any())
    # a plain trace() on the handle invalidates the packed rows
    d12 = torch.cat([pos, dns, rot, scl, torch.zeros_like(dns)], 1).contiguous()
    b.trace(0, degree, d12, sph, ro, rd, None, sensor, ts[0], ts[1], ps[0], ps[1])
    with pytest.raises(RuntimeError, match="forward on this handle"):
        b.trace_bwd_model_fields(0, degree, n, ro, rd, sensor, ts[0], ts[1], ps[0], ps[1], out_b[0], g, out_b[1], dg)


@pytest.mark.parametrize("case", ["c1_pinhole_128", "fisheye_distorted", "dense_big_splats", "ragged_100x70"])
def test_raw_parameter_render_against_the_oracle_and_the_activated_render(case):
    """Tracer.raw_parameters (the product default for a model with the reference's activation callables: the model's PRE-ACTIVATION
    rotation / scale / density tensors go to gut_trace_raw_model_fields, which applies normalize / exp / sigmoid in-kernel, and the
    backward returns raw-parameter gradients): (1) the rows the library activated equal torch's own activations to a few ulps;
    (2) the oracle fed THOSE rows agrees like everywhere else — integer buffers and projection floats bit for bit, image within
    tolerance; (3) the parameter gradients equal the ones autograd produces through torch's activations (the reference's path) up to
    the float-atomic order and the ulps of (1)."""
    mk, kind, W, H, (eye, tgt), kw = CASES[case]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    batch = to_batch(view, DEV)
    rgba_grad = torch.as_tensor(np.random.default_rng(21).normal(size=(H, W, 4)).astype(np.float32), device=DEV)
    grads, outs, rasters = {}, {}, {}
    for raw in (False, True):
        model = gut_model(sc, 3)
        tracer = gut.Tracer({"render": {}})
        tracer.raw_parameters = raw
        out = tracer.render(model, batch, train=True)
        ((out["pred_rgb"][0] * rgba_grad[..., :3]).sum() + (out["pred_opacity"][0] * rgba_grad[..., 3:]).sum()).backward()
        grads[raw] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        outs[raw], rasters[raw] = out, tracer.tracer_wrapper
        if raw:
            rows = tracer.tracer_wrapper.debug_buffer("packed_rows").reshape(-1, 12).cpu().numpy()
            with torch.no_grad():
                ref_rows = torch.cat([model.positions, model.get_density(), model.get_rotation(), model.get_scale()], 1).cpu().numpy()
            # (1) in-kernel activation vs torch's: a few ulps (expf / division vs reciprocal-multiply), never more
            assert np.allclose(rows[:, :11], ref_rows, rtol=4e-7, atol=1e-30), float(np.abs(rows[:, :11] / ref_rows - 1).max())
            assert np.allclose(rows[:, 11], np.linalg.norm(model.rotation.detach().cpu().numpy(), axis=1), rtol=1e-6)
            # (2) the oracle on the rows the kernels read (pad column zeroed: the oracle's rows are [pos | dns | quat | scale | 0])
            d12 = rows.copy(); d12[:, 11] = 0.0
            sph = model.get_features().detach().cpu().numpy()
            ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=3)
            r = tracer.tracer_wrapper
            for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
                assert np.array_equal(r.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
            assert np.array_equal(r.debug_buffer("sorted_keys").cpu().numpy().view(np.uint64), ref["sorted_keys"])
            for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
                got = r.debug_buffer(key).cpu().numpy().view(np.uint32)
                assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
            margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, budget_bound=ROW_FLIP_BOUND)
            rgba = np.concatenate([out["pred_rgb"][0].detach().cpu().numpy(), out["pred_opacity"][0].detach().cpu().numpy()], -1)
            check_colour_outliers(rgba, out["hits_count"][0].detach().cpu().numpy(), ref, margins, label=f"{case}/raw", budget=pixel_budget)
    # (3) the two paths: same image up to what a few ulps in the inputs do, same gradients w.r.t. the model's parameters
    assert float((outs[True]["pred_rgb"] - outs[False]["pred_rgb"]).abs().max()) <= 5e-5
    assert set(grads[True]) == set(grads[False]) and {"rotation", "scale", "density"} <= set(grads[True])
    for k in grads[False]:
        a, b = grads[True][k].cpu().numpy(), grads[False][k].cpu().numpy()
        assert a.shape == b.shape and np.isfinite(a).all(), k
        assert rel_l2(a, b) <= 2e-4, (k, rel_l2(a, b))


@pytest.mark.parametrize("case", ["c1_pinhole_128", "fisheye_144x96"])
def test_render_with_the_models_two_feature_tensors(case):
    """Tracer.render on a model that exposes get_features_albedo / get_features_specular (the reference's MixtureOfGaussians does,
    model.py:68-72) goes through gut_trace_model_fields; with tracer.split_features = False through get_features() and its torch.cat:
    same image bit for bit, same parameter gradients up to the float-atomic order."""
    mk, kind, W, H, (eye, tgt), kw = CASES[case]
    sc = mk()
    view = make_view(kind, W, H, cams.look_at_c2w(eye, tgt), **kw)
    batch = to_batch(view, DEV)
    gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(6)).to(DEV)
    grads, images = [], []
    for split in (True, False):
        model = gut_model(sc, 3)
        tracer = gut.Tracer({"render": {}})
        tracer.split_features = split
        out = tracer.render(model, batch, train=True)
        (out["pred_rgb"] - gt).abs().mean().backward()
        images.append(out["pred_rgb"].detach())
        grads.append({k: p.grad.detach() for k, p in model.named_parameters()})
    assert torch.equal(images[0], images[1])
    assert set(grads[0]) == set(grads[1]) and "features_specular" in grads[0]
    for k in grads[0]:
        assert grads[0][k].shape == grads[1][k].shape, k
        assert torch.equal(grads[0][k] == 0, grads[1][k] == 0), k
        assert rel_l2(grads[0][k].cpu().numpy(), grads[1][k].cpu().numpy()) <= 1e-5, k
