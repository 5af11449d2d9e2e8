"""Host side of the boundary against fixtures produced by the REFERENCE's own Python (tests/golden/gen_host_golden.py,
run in the build container; the fixtures are data, nothing of the reference travels).

Pinned here: `_Autograd` packing / argument order / 12-slot backward, `create_camera_parameters` (dataset -> plugin shutter
mapping, keyword arguments of fromOpenCV*, pose), `Tracer.render`'s output dict, pinhole and OpenCV-fisheye camera rays,
fisheye max_angle rule, COLMAP readers (binary + text, ordering), scene extent, SH constants, the exponential
position-LR schedule and the SH-degree step rule.  The device kernels stay "parity unpinned" (DESIGN.md §3).
"""
import importlib
import json
import os

import numpy as np
import pytest
import torch

gut = importlib.import_module("3dgrut_amd")
tracer_mod = importlib.import_module("3dgrut_amd.tracer")
cams = importlib.import_module("3dgrut_amd.cameras")
io_colmap = importlib.import_module("3dgrut_amd.io_colmap")
schedule = importlib.import_module("3dgrut_amd.schedule")
capi = importlib.import_module("3dgrut_amd._capi")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLD, "host_golden.npz"))


@pytest.fixture(scope="module")
def R():
    with open(os.path.join(GOLD, "host_golden.json")) as f:
        return json.load(f)


class _Raster:
    """Recording tracer_wrapper returning the fixture's canned outputs (same role as the generator's FakeRaster)."""

    def __init__(self, G):
        self.c = {k: torch.as_tensor(G["ag_canned_" + k]) for k in ("rgba", "dist", "hits", "vis", "dens_grd", "sph_grd")}
        self.trace_args = self.bwd_args = None

    def trace(self, *a):
        self.trace_args = a
        return self.c["rgba"].clone(), self.c["dist"].clone(), self.c["hits"].clone(), self.c["vis"].clone()

    def trace_bwd(self, *a):
        self.bwd_args = a
        return self.c["dens_grd"].clone(), self.c["sph_grd"].clone()

    def collect_times(self):
        return {"forward_render": 1.25}


def _leaves(G):
    t = lambda k: torch.as_tensor(G["ag_in_" + k]).clone()
    leaves = {k: t(k).requires_grad_(True) for k in ("pos", "rot", "scl", "dns", "sph")}
    rays = {k: t(k).requires_grad_(True) for k in ("ray_ori", "ray_dir")}
    return leaves, rays


def test_autograd_packing_argument_order_and_backward_slots(G, R):
    """threedgut_tracer/tracer.py:159-286 driven with the same inputs and the same canned native outputs."""
    rec = R["autograd"]
    leaves, rays = _leaves(G)
    w = _Raster(G)
    poses = gut.tracer.SensorPose3D(T_world_sensors=[torch.as_tensor(G["ag_pose_start"]), torch.as_tensor(G["ag_pose_end"])],
                                    timestamps_us=rec["ts"])
    sensor = ("sensor-params-object",)
    out = gut.Tracer._Autograd.apply(w, rec["frame_id"], rec["n_active_features"], rays["ray_ori"], rays["ray_dir"], leaves["pos"],
                                     leaves["rot"], leaves["scl"], leaves["dns"], leaves["sph"], sensor, poses)
    assert len(out) == 4
    for k, o in zip(("rgba", "dist", "hits", "vis"), out):
        assert np.array_equal(o.detach().numpy(), G["ag_out_" + k])
    (out[0] * torch.as_tensor(G["ag_up_rgba_grd"])).sum().add((out[1] * torch.as_tensor(G["ag_up_dist_grd"])).sum()).backward()
    ta, ba = w.trace_args, w.bwd_args
    # positional layout of trace(): 12 arguments, trace_bwd(): the same 12 + rgba, rgba_grd, dist, dist_grd
    assert len(ta) == rec["n_trace_args"] == 12 and len(ba) == rec["n_bwd_args"] == 16
    assert ta[0] == rec["frame_id"] and ta[1] == rec["n_active_features"] and ba[0] == ta[0] and ba[1] == ta[1]
    assert np.array_equal(ta[2].detach().numpy(), G["ag_particle_density"])     # [pos | dns | rot | scl | 0]
    assert np.array_equal(ta[3].detach().numpy(), G["ag_particle_radiance"])
    assert np.array_equal(ba[2].detach().numpy(), G["ag_particle_density"]) and np.array_equal(ba[3].detach().numpy(), G["ag_particle_radiance"])
    assert np.array_equal(ta[4].detach().numpy(), G["ag_in_ray_ori"]) and np.array_equal(ta[5].detach().numpy(), G["ag_in_ray_dir"])
    # slot 6 is ray_time: the reference passes an int64 [1,H,W,1] tensor no kernel reads; this build passes None (documented)
    assert rec["trace_args"][6]["dtype"] == "int64" and ta[6] is None and ba[6] is None
    assert ta[7] is sensor and ba[7] is sensor and rec["sensor_is_passed_through"]
    assert [int(ta[8]), int(ta[9])] == rec["ts"] and [int(ba[8]), int(ba[9])] == rec["ts"]
    assert np.array_equal(np.asarray(ta[10]), G["ag_pose_start"]) and np.array_equal(np.asarray(ta[11]), G["ag_pose_end"])
    assert np.array_equal(ba[12].detach().numpy(), G["ag_bwd_rgba"]) and np.array_equal(ba[13].detach().numpy(), G["ag_bwd_rgba_grd"])
    assert np.array_equal(ba[14].detach().numpy(), G["ag_bwd_dist"]) and np.array_equal(ba[15].detach().numpy(), G["ag_bwd_dist_grd"])
    # the [3,1,4,3,1] split and the slots that carry gradients (pos, rot, scl, dns, sph); rays get none
    for k, v in leaves.items():
        assert np.array_equal(v.grad.numpy(), G["ag_grad_" + k]), k
    assert rays["ray_ori"].grad is None and rays["ray_dir"].grad is None and rec["ray_grads_none"]


def _batch_from_case(rec, c2w):
    dummy = torch.zeros(1, 2, 2, 3)
    vals = lambda k: np.array(rec[k]["values"], np.float32)
    if rec["case"] == "list":
        return gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None], intrinsics=list(rec["input_intrinsics"]))
    name = rec["case"].split("_", 1)[1]
    if name == "int5":
        st = 5
    else:
        st = cams.ShutterType[name]
    res = np.array(rec["resolution"]["values"], np.int64)
    if rec["fn"].startswith("fromOpenCVPinhole"):
        K = dict(resolution=res, shutter_type=st, principal_point=vals("principal_point"), focal_length=vals("focal_length"),
                 radial_coeffs=vals("radial_coeffs"), tangential_coeffs=vals("tangential_coeffs"),
                 thin_prism_coeffs=vals("thin_prism_coeffs"))
        return gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
    K = dict(resolution=res, shutter_type=st, principal_point=vals("principal_point"), focal_length=vals("focal_length"),
             radial_coeffs=vals("radial_coeffs"), max_angle=rec["max_angle"]["value"])
    return gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None], intrinsics_OpenCVFisheyeCameraModelParameters=K)


def test_shutter_enums_match_the_reference(R):
    assert {m.name: int(m.value) for m in cams.ShutterType} == R["dataset_shutter_type"]        # camera_models.py:29-36 (1..5)
    assert {m.name: int(m.value) for m in tracer_mod.ShutterType} == R["plugin_shutter_type"]   # bindings.cpp:87-92 (0..4)


def test_create_camera_parameters_against_the_reference(R):
    """tracer.py:361-431 with a recording plugin: same camera struct contents and pose for every intrinsics path and
    every dataset shutter type (the round-1 bug: dataset GLOBAL=5 raised, 1 was read as LEFT_TO_RIGHT)."""
    cc = R["create_camera_parameters"]
    c2w = torch.tensor(cc["c2w"], dtype=torch.float32).reshape(4, 4)
    seen = set()
    for rec in cc["cases"]:
        sensor, poses = gut.Tracer.create_camera_parameters(_batch_from_case(rec, c2w))
        cam = sensor.cam
        seen.add(rec["shutter_type"]["name"])
        assert cam.shutter == rec["shutter_type"]["value"], rec["case"]
        assert cam.model == (capi.CAMERA_PINHOLE if "Pinhole" in rec["fn"] else capi.CAMERA_FISHEYE)
        f32 = lambda k: np.array(rec[k]["values"], np.float32)
        assert np.array_equal(np.array(cam.principal_point[:], np.float32), f32("principal_point")), rec["case"]
        assert np.array_equal(np.array(cam.focal_length[:], np.float32), f32("focal_length")), rec["case"]
        nrad = len(rec["radial_coeffs"]["values"])
        assert np.array_equal(np.array(cam.radial_coeffs[:nrad], np.float32), f32("radial_coeffs"))
        if "Pinhole" in rec["fn"]:
            assert np.array_equal(np.array(cam.tangential_coeffs[:], np.float32), f32("tangential_coeffs"))
            assert np.array_equal(np.array(cam.thin_prism_coeffs[:], np.float32), f32("thin_prism_coeffs"))
        else:
            assert cam.max_angle == np.float32(rec["max_angle"]["value"])
        assert poses.timestamps_us == rec["timestamps_us"]
        assert np.array_equal(np.asarray(poses.T_world_sensors[0], np.float32), np.array(rec["pose_start"], np.float32)), rec["case"]
        assert np.array_equal(np.asarray(poses.T_world_sensors[1], np.float32), np.array(rec["pose_end"], np.float32))
    assert seen == set(R["plugin_shutter_type"])


def test_create_camera_parameters_rejects_what_the_reference_rejects(R):
    cc = R["create_camera_parameters"]
    c2w = torch.tensor(cc["c2w"], dtype=torch.float32).reshape(4, 4)
    rec = next(r for r in cc["cases"] if r["case"] == "pinhole_GLOBAL")
    assert cc["rejected"]["0"] == "KeyError" and cc["rejected"]["6"] == "KeyError"
    for bad in (0, 6):
        b = _batch_from_case(rec, c2w)
        b.intrinsics_OpenCVPinholeCameraModelParameters["shutter_type"] = bad
        with pytest.raises(KeyError):
            gut.Tracer.create_camera_parameters(b)
    b = _batch_from_case(rec, c2w)
    b.intrinsics_OpenCVPinholeCameraModelParameters["shutter_type"] = tracer_mod.ShutterType.GLOBAL   # plugin enum: wrong type
    with pytest.raises(KeyError):
        gut.Tracer.create_camera_parameters(b)
    dummy = torch.zeros(1, 2, 2, 3)
    with pytest.raises((ValueError, AttributeError)):   # the reference trips over gpu_batch.keys() (AttributeError)
        gut.Tracer.create_camera_parameters(gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None]))


def test_reference_shaped_dataset_batch_is_accepted():
    """A batch in the shape the reference's COLMAP dataset emits (`params.to_dict()`: numpy arrays turned into lists,
    the dataset enum member under "shutter_type", dataset_colmap.py:138-183)."""
    K = dict(resolution=[1237, 822], shutter_type=cams.ShutterType.GLOBAL, principal_point=[np.float32(618.5), np.float32(411.0)],
             focal_length=[np.float32(1040.5), np.float32(1041.75)], radial_coeffs=[np.float32(0)] * 6,
             tangential_coeffs=[np.float32(0)] * 2, thin_prism_coeffs=[np.float32(0)] * 4)
    dummy = torch.zeros(1, 2, 2, 3)
    sensor, _ = gut.Tracer.create_camera_parameters(
        gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=torch.eye(4)[None], intrinsics_OpenCVPinholeCameraModelParameters=K))
    assert sensor.cam.shutter == int(tracer_mod.ShutterType.GLOBAL) == 4
    K["shutter_type"] = cams.ShutterType.ROLLING_TOP_TO_BOTTOM
    sensor, _ = gut.Tracer.create_camera_parameters(
        gut.Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=torch.eye(4)[None], intrinsics_OpenCVPinholeCameraModelParameters=K))
    assert sensor.cam.shutter == int(tracer_mod.ShutterType.ROLLING_TOP_TO_BOTTOM) == 0
    # the repo's own builders emit the dataset numbering
    assert cams.pinhole_intrinsics_dict(8, 8, 4.0, 4.0)["shutter_type"] == 5
    assert cams.fisheye_intrinsics_dict(8, 8, 4.0, 4.0)["shutter_type"] == 5


def test_render_output_dict_against_the_reference(G, R):
    """Tracer.render (tracer.py:304-351) with the fake native module: same keys, shapes and values."""
    rec = R["render"]
    assert rec["ok"]
    t = lambda k: torch.as_tensor(G["ag_in_" + k])

    class Gaussians:
        num_gaussians = int(G["ag_in_pos"].shape[0])
        n_active_features = 2
        positions = t("pos")
        def get_rotation(self): return t("rot")
        def get_scale(self): return t("scl")
        def get_density(self): return t("dns")
        def get_features(self): return t("sph")
        def background(self, T_to_world, rays_d, rgb, opacity, train):
            self.train = train
            return rgb + 0.25 * (1.0 - opacity), opacity

    tr = gut.Tracer.__new__(gut.Tracer)
    tr.tracer_wrapper = _Raster(G)
    gs = Gaussians()
    c2w = torch.tensor(R["create_camera_parameters"]["c2w"], dtype=torch.float32).reshape(4, 4)
    batch = gut.Batch(rays_ori=t("ray_ori"), rays_dir=t("ray_dir"), T_to_world=c2w[None], intrinsics=[100.0, 110.0, 3.5, 2.5])
    out = tr.render(gs, batch, train=True, frame_id=9)
    assert list(out.keys()) == rec["keys"]
    assert out["frame_time_ms"] == rec["frame_time_ms"] and tr.tracer_wrapper.trace_args[0] == rec["frame_id"]
    assert gs.train is rec["background_train"]
    for k, shp in rec["shapes"].items():
        if shp is not None:
            assert list(out[k].shape) == shp, k
            assert np.array_equal(out[k].detach().numpy(), G["render_" + k]), k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pinhole_rays_against_the_reference(G, tag):
    """datasets/utils.py:39-59 (float64) vs cameras.pinhole_rays (float32 output)."""
    w, h, fx, fy = G[f"rays_{tag}_whff"]
    ro, rd = cams.pinhole_rays(int(w), int(h), fx, fy)
    ref_d = G[f"rays_{tag}_dir"].reshape(1, int(h), int(w), 3)
    assert np.array_equal(rd, ref_d.astype(np.float32))           # the dataset casts to float32 (dataset_colmap.py:151)
    assert np.array_equal(ro, G[f"rays_{tag}_ori"].reshape(1, int(h), int(w), 3).astype(np.float32))


@pytest.mark.parametrize("tag", ["zero", "dist"])
def test_fisheye_rays_against_the_reference(G, tag):
    """camera_models.py:156-248 (float32 torch, 3 Newton steps) vs cameras.fisheye_rays (float64 numpy, cast)."""
    w, h, fx, fy, cx, cy, max_angle = G[f"fisheye_{tag}_params"]
    w, h = int(w), int(h)
    # the reference evaluates everything from float32 focal lengths / principal point
    fx, fy, cx, cy = [float(np.float32(v)) for v in (fx, fy, cx, cy)]
    assert abs(cams.fisheye_max_angle(w, h, fx, fy, cx, cy) - max_angle) <= 1e-6 * max_angle   # dataset_colmap.py:167-172
    ro, rd = cams.fisheye_rays(w, h, fx, fy, cx, cy, radial=G[f"fisheye_{tag}_radial"])
    ref = G[f"fisheye_{tag}_dir"]
    assert rd.shape == ref.shape and not ro.any()
    assert np.abs(rd - ref).max() <= 2e-6     # float32 reference arithmetic vs float64 here: a few ulp of a unit vector


def test_max_radius_rule(G):
    for w, h, cx, cy, expect in G["max_radius_cases"]:
        mx, my = max(cx, w - cx), max(cy, h - cy)
        assert abs(np.hypot(mx, my) - expect) <= 1e-9 * expect
        got = cams.fisheye_max_angle(w, h, 700.0, 710.0, cx, cy)
        assert abs(got - max(2 * expect / 700.0, 2 * expect / 710.0) / 2) <= 1e-9


@pytest.mark.parametrize("fmt", ["bin", "txt"])
def test_colmap_readers_against_the_reference(G, R, fmt):
    """datasets/utils.py:258-566 on tests/golden/colmap_model (written by the generator in COLMAP's documented layouts)."""
    d = os.path.join(GOLD, "colmap_model", fmt)
    rec = R["colmap"][fmt]
    if fmt == "bin":
        cams_ = io_colmap.read_cameras_binary(os.path.join(d, "cameras.bin"))
        imgs = io_colmap.read_images_binary(os.path.join(d, "images.bin"))
        xyz, rgb, err = io_colmap.read_points3D_binary(os.path.join(d, "points3D.bin"))
    else:
        cams_ = io_colmap.read_cameras_text(os.path.join(d, "cameras.txt"))
        imgs = io_colmap.read_images_text(os.path.join(d, "images.txt"))
        xyz, rgb, err = io_colmap.read_points3D_text(os.path.join(d, "points3D.txt"))
    assert len(cams_) == len(rec["cameras"])
    for c in rec["cameras"]:
        m = cams_[c["id"]]
        assert (m.model, m.width, m.height) == (c["model"], c["width"], c["height"])
        assert np.array_equal(np.asarray(m.params, np.float64), np.array(c["params"], np.float64))
    # same ORDER (sorted by image name): the every-8th train/test split indexes it (dataset_colmap.py:81-91)
    assert [i.name for i in imgs] == [i["name"] for i in rec["images"]]
    for got, exp in zip(imgs, rec["images"]):
        assert got.id == exp["id"] and got.camera_id == exp["camera_id"]
        assert np.array_equal(np.asarray(got.qvec, np.float64), np.array(exp["qvec"]))
        assert np.array_equal(np.asarray(got.tvec, np.float64), np.array(exp["tvec"]))
    assert np.array_equal(xyz, G[f"colmap_{fmt}_xyz"])
    assert np.array_equal(rgb.astype(np.float64), G[f"colmap_{fmt}_rgb"])
    assert np.array_equal(err.reshape(-1), G[f"colmap_{fmt}_err"].reshape(-1))


def test_qvec_to_rotation_and_scene_extent(G):
    for q, Rm in zip(G["qvec_in"], G["qvec_so3"]):
        assert np.abs(io_colmap.qvec_to_rotation(q) - Rm).max() <= 1e-15
    cc = G["center_diag_in"]
    centre = cc.mean(0)
    assert np.allclose(centre, G["center_diag_center"], atol=0, rtol=1e-15)
    assert abs(np.linalg.norm(cc - centre, axis=1).max() - float(G["center_diag_diag"])) <= 1e-15


def test_sh_constants_against_the_reference(G):
    """threedgrut/utils/render.py: the constants compiled into the kernels / oracle (gut_project.hip, gut_oracle.c)."""
    prt = importlib.import_module("oracle.per_ray_torch")
    assert prt._C0 == float(G["sh_C0"]) and prt._C1 == float(G["sh_C1"])
    assert np.array_equal(np.array(prt._C2), G["sh_C2"]) and np.array_equal(np.array(prt._C3), G["sh_C3"])
    src = open(os.path.join(os.path.dirname(GOLD), "..", "3dgrut_amd", "csrc", "gut_project.hip")).read()
    osrc = open(os.path.join(os.path.dirname(GOLD), "..", "oracle", "gut_oracle.c")).read()
    for c in [float(G["sh_C0"]), float(G["sh_C1"])] + [abs(float(v)) for v in G["sh_C2"]] + [abs(float(v)) for v in G["sh_C3"]]:
        assert repr(c) + "f" in src and repr(c) + "f" in osrc, c
    x = G["rgb2sh_in"]
    assert np.allclose((x - 0.5) / 0.28209479177387814, G["rgb2sh_out"], rtol=1e-15, atol=0)   # io_colmap.initial_gaussians


def test_position_lr_schedule_against_the_reference(G, R):
    lr_i, lr_f, max_steps = G["sched_args"]
    f = schedule.exponential_scheduler(lr_i, lr_f, int(max_steps))
    got = np.array([f(int(s)) for s in G["sched_steps"]])
    assert np.allclose(got, G["sched_lr"], rtol=1e-12, atol=0)
    for rec in R["check_step_condition"]:
        assert schedule.check_step_condition(*rec["args"]) is rec["result"], rec
    assert R["skip_scheduler_returns_none"]
    assert R["sh_degree_to_num_features"] == {"0": 3, "1": 12, "2": 27, "3": 48}   # the [N,48] radiance layout


def test_train_schedule_follows_the_reference_iteration_order(G):
    """trainer.py:745-765: the rate set after iteration g is sched(g); the SH degree rises after iterations 1000, 2000, 3000."""
    extent = 4.5
    s = schedule.TrainSchedule(scene_extent=extent)
    assert s.n_active_features == 0 and abs(s.position_lr - 0.00016 * extent) < 1e-18
    ref = dict(zip([int(v) for v in G["sched_steps"]], G["sched_lr"]))
    degs = {}
    for g in range(0, 4002):
        lr, deg = s.after_optimizer_step(g)
        if g in ref:
            assert abs(lr - ref[g]) <= 1e-12 * ref[g]
        degs[g] = deg
    assert degs[999] == 0 and degs[1000] == 1 and degs[1999] == 1 and degs[2000] == 2 and degs[3000] == 3 and degs[4001] == 3


# ---------------------------------------------------------------------------------------------------------------------
# N3: strategy.GSStrategy against threedgrut/strategy/gs.py itself (tests/golden/gen_strategy_golden.py drove the
# reference's GSStrategy + BaseStrategy._update_param_with_optimizer on a fake MixtureOfGaussians with a real
# torch.optim.Adam; strategy_golden.npz holds every parameter, both Adam moments and the densification buffers after each
# stage).  Same inputs, same recorded standard-normal draws for the split -> same rows, in the same order.
# ---------------------------------------------------------------------------------------------------------------------
_REF_PARAMS = ("positions", "density", "rotation", "scale", "features_albedo", "features_specular")


class _StrategyStepper:
    """NativeTrainStep's state as strategy.py sees it (tensors only; no GPU)."""
    def __init__(self, S, tag):
        native = importlib.import_module("3dgrut_amd.native")
        t = lambda k: torch.as_tensor(S[f"{tag}/{k}"])
        n = S[f"{tag}/positions"].shape[0]
        self.model = native.NativeGaussianModel.__new__(native.NativeGaussianModel)
        self.model.raw, self.m12, self.v12 = (torch.zeros(n, 12) for _ in range(3))
        for dst, suffix in ((self.model.raw, ""), (self.m12, "/exp_avg"), (self.v12, "/exp_avg_sq")):
            dst[:, 0:3], dst[:, 3:4], dst[:, 4:8], dst[:, 8:11] = (t(k + suffix) for k in ("positions", "density", "rotation", "scale"))
        cat = lambda suffix: torch.cat([t("features_albedo" + suffix), t("features_specular" + suffix)], 1).contiguous()
        self.model.features, self.m48, self.v48 = cat(""), cat("/exp_avg"), cat("/exp_avg_sq")
        self.model.permutation, self.model.spatial_order = None, False
        self.row_listeners, self.post_backward_hook, self.step_id = [], None, 1

    def resize_workspace(self):
        pass


def _assert_stage(S, tag, st, gs, exact=True, skip=()):
    cmp_ = (lambda a, b, what: np.testing.assert_array_equal(a, b, err_msg=what)) if exact else \
        (lambda a, b, what: np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7, err_msg=what))
    for src, suffix in ((st.model.raw, ""), (st.m12, "/exp_avg"), (st.v12, "/exp_avg_sq")):
        for k, sl in (("positions", slice(0, 3)), ("density", slice(3, 4)), ("rotation", slice(4, 8)), ("scale", slice(8, 11))):
            if (k + suffix) not in skip:
                cmp_(src[:, sl].numpy(), S[f"{tag}/{k}{suffix}"], f"{tag}/{k}{suffix}")
    for src, suffix in ((st.model.features, ""), (st.m48, "/exp_avg"), (st.v48, "/exp_avg_sq")):
        cmp_(src[:, 0:3].numpy(), S[f"{tag}/features_albedo{suffix}"], f"{tag}/features_albedo{suffix}")
        cmp_(src[:, 3:].numpy(), S[f"{tag}/features_specular{suffix}"], f"{tag}/features_specular{suffix}")
    np.testing.assert_allclose(gs.grad_norm_accum.numpy(), S[f"{tag}/grad_norm_accum"], rtol=2e-6, atol=1e-12, err_msg=f"{tag}/accum")
    np.testing.assert_array_equal(gs.grad_norm_denom.numpy(), S[f"{tag}/grad_norm_denom"], err_msg=f"{tag}/denom")


def test_gs_strategy_against_the_reference():
    strategy = importlib.import_module("3dgrut_amd.strategy")
    S = np.load(os.path.join(GOLD, "strategy_golden.npz"))
    st = _StrategyStepper(S, "start")
    gs = strategy.GSStrategy(st, seed=0).attach()
    assert st.post_backward_hook is not None
    extent = float(S["scene_extent"])
    # post_backward x 3 views (gs.py:106-115)
    for g, s in zip(S["view_grads"], S["view_sensors"]):
        st.post_backward_hook(torch.as_tensor(g), torch.as_tensor(s))
    _assert_stage(S, "after_buffer", st, gs)
    # densify_gaussians = clone then split (gs.py:117-210) with the recorded draws
    unit, used = torch.as_tensor(S["unit_draws"]), []
    def draws(shape):
        used.append(shape[0])
        return unit[sum(used[:-1]):sum(used)]
    gs.unit_normal_fn = draws
    n0 = st.model.raw.shape[0]
    gs.densify(extent, step=600)
    assert sum(used) == int(S["draws_used"]) and st.model.raw.shape[0] == S["after_densify/positions"].shape[0] > n0
    # every row and both moments bit for bit, except the split children's positions: R(q) @ (unit * std) is evaluated with
    # torch.nn.functional.normalize here and with the reference's own quaternion_to_so3 there (last-ulp differences)
    _assert_stage(S, "after_densify", st, gs, skip=("positions",))
    np.testing.assert_allclose(st.model.raw[:, 0:3].numpy(), S["after_densify/positions"], rtol=1e-5, atol=2e-7)
    st.model.raw[:, 0:3] = torch.as_tensor(S["after_densify/positions"])     # continue from the reference's bits
    assert gs.prune_opacity() == S["after_densify/positions"].shape[0] - S["after_prune/positions"].shape[0] > 0
    _assert_stage(S, "after_prune", st, gs)
    gs.decay_density()
    _assert_stage(S, "after_decay", st, gs)
    gs.reset_density()
    _assert_stage(S, "after_reset", st, gs, exact=False)    # the clamp bound is logit(0.01) in float32 there, in double here
    assert float(st.m12[:, 3].abs().max()) == 0.0 and float(st.v12[:, 3].abs().max()) == 0.0


def test_gs_schedule_against_the_reference():
    """post_optimizer_step / post_backward fire on exactly the iterations the reference's check_step_condition selects for
    configs/strategy/gs.yaml (fixture: the selected iterations 0..16000)."""
    strategy = importlib.import_module("3dgrut_amd.strategy")
    S = np.load(os.path.join(GOLD, "strategy_golden.npz"))
    sc = strategy.GS_SCHEDULE
    for key, name in (("schedule_densify", "densify"), ("schedule_prune", "prune"), ("schedule_reset", "reset_density")):
        mine = [s for s in range(16001) if schedule.check_step_condition(s, *sc[name])]
        assert mine == S[key].tolist(), name
    assert [s for s in range(20) if schedule.check_step_condition(s, 0, sc["densify"][1], 1)] == S["schedule_buffer"].tolist()
    assert not any(schedule.check_step_condition(s, *sc["density_decay"]) for s in range(16001))


# ---------------------------------------------------------------------------------------------------------------------
# N3: strategy.MCMCStrategy against threedgrut/strategy/mcmc.py itself (tests/golden/gen_mcmc_golden.py drove the reference's
# MCMCStrategy on a fake MixtureOfGaussians with a real torch.optim.Adam; the multinomial draws, the relocation kernel's inputs
# AND outputs (closed form in float64 — the CUDA plugin cannot exist here) and the perturbation's normal draws are recorded).
# Same state, same draws, same kernel outputs -> same rows and moments after relocate / add / perturb.
# ---------------------------------------------------------------------------------------------------------------------
def _assert_mcmc_stage(S, tag, st, exact=True, skip=()):
    cmp_ = (lambda a, b, what: np.testing.assert_array_equal(a, b, err_msg=what)) if exact else \
        (lambda a, b, what: np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7, err_msg=what))
    for src, suffix in ((st.model.raw, ""), (st.m12, "/exp_avg"), (st.v12, "/exp_avg_sq")):
        for k, sl in (("positions", slice(0, 3)), ("density", slice(3, 4)), ("rotation", slice(4, 8)), ("scale", slice(8, 11))):
            if (k + suffix) not in skip:
                cmp_(src[:, sl].numpy(), S[f"{tag}/{k}{suffix}"], f"{tag}/{k}{suffix}")
    for src, suffix in ((st.model.features, ""), (st.m48, "/exp_avg"), (st.v48, "/exp_avg_sq")):
        cmp_(src[:, 0:3].numpy(), S[f"{tag}/features_albedo{suffix}"], f"{tag}/features_albedo{suffix}")
        cmp_(src[:, 3:].numpy(), S[f"{tag}/features_specular{suffix}"], f"{tag}/features_specular{suffix}")


def test_mcmc_strategy_against_the_reference():
    strategy = importlib.import_module("3dgrut_amd.strategy")
    S = np.load(os.path.join(GOLD, "mcmc_golden.npz"))
    st = _StrategyStepper(S, "start")
    mc = strategy.MCMCStrategy(st, opacity_threshold=0.005, binom_n_max=51, max_n_gaussians=330, noise_lr=500000.0)
    np.testing.assert_array_equal(mc.binoms.numpy(), S["binoms"])        # the Pascal table handed to the kernel
    calls = dict(sample=0, kernel=0)

    def sample(weights, n, step):
        i = calls["sample"]; calls["sample"] += 1
        np.testing.assert_array_equal(weights.numpy(), S[f"sample{i}/probabilities"], err_msg=f"sample {i}: weights")
        assert n == S[f"sample{i}/indices"].shape[0]
        return torch.as_tensor(S[f"sample{i}/indices"])

    def kernel(dens, scales, ratios):
        i = calls["kernel"]; calls["kernel"] += 1
        np.testing.assert_array_equal(dens.numpy().reshape(-1), S[f"kernel{i}/opacities"].reshape(-1), err_msg=f"kernel {i}: opacities")
        np.testing.assert_array_equal(scales.numpy(), S[f"kernel{i}/scales"], err_msg=f"kernel {i}: scales")
        np.testing.assert_array_equal(ratios.numpy(), S[f"kernel{i}/ratios"], err_msg=f"kernel {i}: ratios")
        assert ratios.dtype == torch.int32 and dens.is_contiguous() and scales.is_contiguous()
        return torch.as_tensor(S[f"kernel{i}/new_opacities"]).reshape(dens.shape), torch.as_tensor(S[f"kernel{i}/new_scales"])

    mc.sample_fn, mc.relocation_fn = sample, kernel
    mc.unit_normal_fn = lambda shape, step: torch.as_tensor(S["perturb/unit_draws"]).reshape(shape)

    assert mc.relocate() == int(S["n_dead"]) > 0
    _assert_mcmc_stage(S, "after_relocate", st)
    assert mc.add_new() == 15 and st.model.raw.shape[0] == 315
    _assert_mcmc_stage(S, "after_add", st)
    mc.perturb(float(S["position_lr"]))
    # positions: covariance @ noise with R from torch.nn.functional.normalize here, the reference's quaternion_to_so3 there
    _assert_mcmc_stage(S, "after_perturb", st, skip=("positions",))
    np.testing.assert_allclose(st.model.raw[:, 0:3].numpy(), S["after_perturb/positions"], rtol=2e-5, atol=1e-7)
    moved = np.abs(S["after_perturb/positions"] - S["after_add/positions"]).max(axis=1)
    assert (moved > 0).sum() > 10                                            # (the noise is gated to near-transparent Gaussians)
    st.model.raw[:, 0:3] = torch.as_tensor(S["after_perturb/positions"])     # continue from the reference's bits
    assert mc.add_new() == 15 and mc.add_new() == 0                          # 330 = the cap: nothing more, and no kernel call
    _assert_mcmc_stage(S, "after_cap", st)
    assert calls["kernel"] == int(S["kernel_calls"]) and calls["sample"] == int(S["sample_calls"])


def test_mcmc_schedule_against_the_reference():
    """post_optimizer_step fires relocate / add / perturb on exactly the iterations the reference's check_step_condition selects for
    configs/strategy/mcmc.yaml (fixture: the selected iterations 0..28000), in the reference's order."""
    strategy = importlib.import_module("3dgrut_amd.strategy")
    S = np.load(os.path.join(GOLD, "mcmc_golden.npz"))
    for name in ("relocate", "add", "perturb"):
        mine = [s for s in range(28001) if schedule.check_step_condition(s, *strategy.MCMC_SCHEDULE[name])]
        assert mine == S[f"schedule_{name}"].tolist(), name
    st = _StrategyStepper(S, "start")
    mc = strategy.MCMCStrategy(st)
    log = []
    mc.relocate = lambda step=0: log.append("relocate")
    mc.add_new = lambda step=0: log.append("add")
    mc.perturb = lambda lr, step=0: log.append("perturb")
    assert mc.post_optimizer_step(600, 1e-4) == ["relocate", "add", "perturb"] == log
    assert mc.post_optimizer_step(601, 1e-4) == ["perturb"] and mc.post_optimizer_step(500, 1e-4) == ["perturb"]
    assert mc.post_optimizer_step(0, 1e-4) == [] and mc.post_optimizer_step(27500, 1e-4) == []


def test_render_hands_over_the_models_two_feature_tensors(G, R):
    """Tracer.render on a model with the reference's get_features_albedo / get_features_specular accessors (model.py:68-72) and a
    wrapper that offers trace_model_fields: the two tensors go to _Autograd as they are (features_albedo in the mog_sph slot,
    features_specular as the thirteenth argument, no get_features() call), the six gradients of trace_bwd_model_fields reach
    the six leaves, and with tracer.split_features = False — or a wrapper without the entry point — the reference's call is made."""
    t = lambda k: torch.as_tensor(G["ag_in_" + k]).clone()
    n = int(G["ag_in_pos"].shape[0])
    canned = {k: torch.as_tensor(G["ag_canned_" + k]) for k in ("rgba", "dist", "hits", "vis")}
    grads = {k: torch.full((n, c), float(i + 1)) for i, (k, c) in enumerate((("pos", 3), ("dns", 1), ("rot", 4), ("scl", 3), ("alb", 3), ("spec", 45)))}

    class Wrapper(_Raster):
        def __init__(self, G):
            super().__init__(G)
            self.model_fields_args = self.model_fields_bwd_args = None

        def trace_model_fields(self, *a):
            self.model_fields_args = a
            return tuple(canned[k].clone() for k in ("rgba", "dist", "hits", "vis"))

        def trace_bwd_model_fields(self, *a):
            self.model_fields_bwd_args = a
            return tuple(grads[k] for k in ("pos", "dns", "rot", "scl", "alb", "spec"))

    class Gaussians:
        num_gaussians, n_active_features = n, 2
        def __init__(self):
            self.positions = t("pos").requires_grad_(True)
            self.rot, self.scl, self.dns = (t(k).requires_grad_(True) for k in ("rot", "scl", "dns"))
            sph = t("sph")
            self.alb, self.spec = sph[:, :3].clone().requires_grad_(True), sph[:, 3:].clone().requires_grad_(True)
            self.cat_calls = 0
        def get_rotation(self): return self.rot
        def get_scale(self): return self.scl
        def get_density(self): return self.dns
        def get_features_albedo(self): return self.alb
        def get_features_specular(self): return self.spec
        def get_features(self):
            self.cat_calls += 1
            return torch.cat((self.alb, self.spec), dim=1)
        def background(self, T_to_world, rays_d, rgb, opacity, train): return rgb, opacity

    c2w = torch.tensor(R["create_camera_parameters"]["c2w"], dtype=torch.float32).reshape(4, 4)
    batch = gut.Batch(rays_ori=t("ray_ori"), rays_dir=t("ray_dir"), T_to_world=c2w[None], intrinsics=[100.0, 110.0, 3.5, 2.5])
    tr = gut.Tracer.__new__(gut.Tracer)
    tr.tracer_wrapper = Wrapper(G)
    gs = Gaussians()
    out = tr.render(gs, batch, train=True, frame_id=4)
    a = tr.tracer_wrapper.model_fields_args
    assert a is not None and tr.tracer_wrapper.trace_args is None and gs.cat_calls == 0
    assert a[0] == 4 and a[1] == 2
    for got, want in zip(a[2:8], (gs.positions, gs.dns, gs.rot, gs.scl, gs.alb, gs.spec)):   # pos, density, rotation, scale, albedo, specular
        assert got.shape == want.shape and torch.equal(got.detach(), want.detach())
    (out["pred_rgb"].sum() + out["pred_opacity"].sum()).backward()
    b = tr.tracer_wrapper.model_fields_bwd_args
    assert b is not None and b[0] == 4 and b[1] == 2 and b[2] == n
    for leaf, k in ((gs.positions, "pos"), (gs.dns, "dns"), (gs.rot, "rot"), (gs.scl, "scl"), (gs.alb, "alb"), (gs.spec, "spec")):
        assert torch.equal(leaf.grad, grads[k]), k
    # the switch, and a wrapper with the pybind surface only: the reference's get_features() + trace()
    for wrapper, split in ((Wrapper(G), False), (_Raster(G), True)):
        tr.tracer_wrapper, tr.split_features = wrapper, split
        gs = Gaussians()
        tr.render(gs, batch, train=True, frame_id=5)
        assert gs.cat_calls == 1 and wrapper.trace_args is not None and getattr(wrapper, "model_fields_args", None) is None
