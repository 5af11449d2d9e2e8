"""Generate golden vectors for the HOST side of the 3DGUT tracer boundary from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); nothing of the reference travels — the outputs are data
(inputs + what the reference's code returned for them):

  tests/golden/host_golden.npz          arrays
  tests/golden/host_golden.json         call records (argument order / names / types), enum values, scalars
  tests/golden/colmap_model/{bin,txt}/  a small synthetic COLMAP sparse model (written by THIS script in COLMAP's
                                        documented layouts, with non-empty 2D-point and track lists)

What is driven (each imported by file path; packages that are not installed here and are not used by the code under test
are replaced by empty module shells: omegaconf, dataclasses_json, torch.utils.tensorboard):

  A. threedgut_tracer/tracer.py  Tracer._Autograd.forward/backward (:159-286) with a recording fake `tracer_wrapper`
     -> pins the [pos|dns|rot|scl|0] packing, the positional argument order of trace / trace_bwd, the [3,1,4,3,1]
        split and the 12-slot backward tuple.
  B. threedgut_tracer/tracer.py  Tracer.__create_camera_parameters (:361-431) with a recording fake `_3dgut_plugin`
     (enum numbering of bindings.cpp:87-92) and batches built from the reference's Batch + dataset ShutterType
     -> pins the dataset->plugin shutter mapping, the keyword arguments handed to fromOpenCV*CameraModelParameters for the
        [fx,fy,cx,cy] / pinhole-dict / fisheye-dict paths, and the pose.
  C. threedgut_tracer/tracer.py  Tracer.render (:304-351) on CPU tensors with the same fakes -> the output dict.
  D. threedgrut/datasets/utils.py  pinhole_camera_rays (:39-59), compute_max_radius (:170-191), get_center_and_diag
     (:130-135), COLMAP readers (:258-566) on the synthetic model.
  E. threedgrut/datasets/camera_models.py  ShutterType values (:29-36), pixels_to_image_points +
     image_points_to_camera_rays (:156-248) for a zero-coefficient and a distorted OpenCV-fisheye camera.
  F. threedgrut/utils/render.py  SH constants, RGB2SH / SH2RGB.
  G. threedgrut/utils/misc.py  exponential_scheduler (:89-96), check_step_condition (:198-202), sh_degree_to_*.
"""
import enum
import importlib.util
import json
import os
import struct
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _shell(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    class _OmegaConf:
        @staticmethod
        def register_new_resolver(*a, **k):
            pass

        @staticmethod
        def to_container(c, *a, **k):
            return c
    _shell("omegaconf", OmegaConf=_OmegaConf, DictConfig=dict)
    _shell("dataclasses_json", DataClassJsonMixin=type("DataClassJsonMixin", (), {}))
    _shell("torch.utils.tensorboard")
    _shell("torch.utils.tensorboard.writer", SummaryWriter=object)
    _shell("threedgrut")
    _shell("threedgrut.datasets")
    _shell("threedgrut.utils")
    protocols = _load("threedgrut.datasets.protocols", "threedgrut/datasets/protocols.py")
    camera_models = _load("threedgrut.datasets.camera_models", "threedgrut/datasets/camera_models.py")
    utils = _load("threedgrut.datasets.utils", "threedgrut/datasets/utils.py")
    render = _load("threedgrut.utils.render", "threedgrut/utils/render.py")
    misc = _load("threedgrut.utils.misc", "threedgrut/utils/misc.py")
    tracer = _load("_ref_tracer", "threedgut_tracer/tracer.py")
    return dict(protocols=protocols, camera_models=camera_models, utils=utils, render=render, misc=misc, tracer=tracer)


# ---------------------------------------------------------------------------------------------- fakes
class PluginShutterType(enum.IntEnum):  # numbering of bindings.cpp:87-92 / sensors/cameraModels.h:39-45
    ROLLING_TOP_TO_BOTTOM = 0
    ROLLING_LEFT_TO_RIGHT = 1
    ROLLING_BOTTOM_TO_TOP = 2
    ROLLING_RIGHT_TO_LEFT = 3
    GLOBAL = 4


def _describe(x):
    if isinstance(x, torch.Tensor):
        return dict(kind="tensor", dtype=str(x.dtype).replace("torch.", ""), shape=list(x.shape))
    if isinstance(x, np.ndarray):
        return dict(kind="ndarray", dtype=str(x.dtype), shape=list(x.shape))
    if isinstance(x, enum.Enum):
        return dict(kind="enum", name=x.name, value=int(x.value))
    if isinstance(x, (bool, int, float, str)) or x is None:
        return dict(kind=type(x).__name__, value=x)
    return dict(kind=type(x).__name__)


class FakePlugin:
    """Stands in for lib3dgut_cc: records what the reference's host code hands to the native module."""
    ShutterType = PluginShutterType

    def __init__(self):
        self.camera_calls = []

    def _rec(self, fn, kw):
        rec = dict(fn=fn, order=list(kw.keys()))
        for k, v in kw.items():
            rec[k] = _describe(v)
            if isinstance(v, np.ndarray):
                rec[k]["values"] = [float(t) for t in np.asarray(v, np.float64).reshape(-1)]
            elif isinstance(v, (list, tuple)):
                rec[k]["values"] = [float(t) for t in v]
        self.camera_calls.append(rec)
        return ("camera_model_parameters", len(self.camera_calls) - 1)

    def fromOpenCVPinholeCameraModelParameters(self, **kw):
        return self._rec("fromOpenCVPinholeCameraModelParameters", kw)

    def fromOpenCVFisheyeCameraModelParameters(self, **kw):
        return self._rec("fromOpenCVFisheyeCameraModelParameters", kw)


class FakeRaster:
    """Recording tracer_wrapper: returns canned outputs so the autograd plumbing can be replayed elsewhere."""

    def __init__(self, canned):
        self.canned = canned
        self.trace_args = None
        self.bwd_args = None

    def trace(self, *args):
        self.trace_args = args
        c = self.canned
        return c["rgba"].clone(), c["dist"].clone(), c["hits"].clone(), c["vis"].clone()

    def trace_bwd(self, *args):
        self.bwd_args = args
        return self.canned["dens_grd"].clone(), self.canned["sph_grd"].clone()

    def collect_times(self):
        return {"forward_render": 1.25}


# ---------------------------------------------------------------------------------------------- COLMAP files
def write_colmap_model(out_bin, out_txt, rng):
    os.makedirs(out_bin, exist_ok=True)
    os.makedirs(out_txt, exist_ok=True)
    cams = [(1, 0, "SIMPLE_PINHOLE", 640, 480, [512.25, 320.0, 240.0]),
            (2, 1, "PINHOLE", 1237, 822, [1040.5, 1041.75, 618.5, 411.0]),
            (5, 5, "OPENCV_FISHEYE", 1752, 1168, [791.5, 792.25, 876.0, 584.0, 0.01, -0.002, 0.0003, -0.00004])]
    images = []
    for i in range(6):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.uniform(-3, 3, size=3)
        npts = int(rng.integers(1, 5))   # >= 1: the reference text reader drops blank lines, so an image without points breaks its pairing
        xys = rng.uniform(0, 600, size=(npts, 2))
        p3d = rng.integers(-1, 40, size=npts)
        images.append((i + 1, q, t, cams[i % 3][0], f"frame_{i:05d}.JPG" if i != 3 else f"sub dir/frame {i}.png", xys, p3d))
    npnt = 40
    xyz = rng.normal(size=(npnt, 3)) * 2.0
    rgb = rng.integers(0, 256, size=(npnt, 3))
    err = rng.uniform(0.1, 2.0, size=npnt)
    tracks = [[(int(rng.integers(1, 7)), int(rng.integers(0, 50))) for _ in range(int(rng.integers(0, 4)))] for _ in range(npnt)]
    pid = [int(3 * k + 1) for k in range(npnt)]
    with open(os.path.join(out_bin, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, mid, _, w, h, p in cams:
            f.write(struct.pack("<iiQQ", cid, mid, w, h) + struct.pack("<%dd" % len(p), *p))
    with open(os.path.join(out_bin, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for iid, q, t, cid, name, xys, p3d in images:
            f.write(struct.pack("<idddddddi", iid, *q, *t, cid) + name.encode() + b"\x00" + struct.pack("<Q", len(xys)))
            for (x, y), p in zip(xys, p3d):
                f.write(struct.pack("<ddq", x, y, int(p)))
    with open(os.path.join(out_bin, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", npnt))
        for k in range(npnt):
            f.write(struct.pack("<QdddBBBd", pid[k], *xyz[k], *[int(c) for c in rgb[k]], err[k]) + struct.pack("<Q", len(tracks[k])))
            for a, b in tracks[k]:
                f.write(struct.pack("<ii", a, b))
    with open(os.path.join(out_txt, "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        f.write(f"# Number of cameras: {len(cams)}\n")
        for cid, _, name, w, h, p in cams:
            f.write(f"{cid} {name} {w} {h} " + " ".join(repr(float(v)) for v in p) + "\n")
    with open(os.path.join(out_txt, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n")
        f.write("#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        for iid, q, t, cid, name, xys, p3d in images:
            if " " in name:   # the text format cannot carry names with blanks; the reference splits on whitespace
                name = name.replace(" ", "_")
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in q) + " " + " ".join(repr(float(v)) for v in t) + f" {cid} {name}\n")
            f.write(" ".join(f"{float(x)!r} {float(y)!r} {int(p)}" for (x, y), p in zip(xys, p3d)) + "\n")
    with open(os.path.join(out_txt, "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
        for k in range(npnt):
            tr = " ".join(f"{a} {b}" for a, b in tracks[k])
            f.write(f"{pid[k]} " + " ".join(repr(float(v)) for v in xyz[k]) + " " + " ".join(str(int(c)) for c in rgb[k]) +
                    f" {float(err[k])!r} {tr}".rstrip() + "\n")


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def main():
    ref = load_reference()
    T = ref["tracer"]
    arrays, records = {}, {}
    rng = np.random.default_rng(20261004)
    g = torch.Generator().manual_seed(7)

    # ---------------- A. _Autograd with a recording wrapper (CPU tensors; no kernels involved)
    N, H, W = 37, 5, 7
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    inp = dict(pos=rnd(N, 3), rot=torch.nn.functional.normalize(rnd(N, 4), dim=1), scl=rnd(N, 3).exp(), dns=torch.sigmoid(rnd(N, 1)),
               sph=rnd(N, 48), ray_ori=rnd(1, H, W, 3), ray_dir=torch.nn.functional.normalize(rnd(1, H, W, 3), dim=3))
    canned = dict(rgba=rnd(H, W, 4), dist=rnd(H, W, 1), hits=rnd(H, W, 1).abs().round(), vis=(rnd(N, 1) > 0).float(),
                  dens_grd=rnd(N, 12), sph_grd=rnd(N, 48))
    up = dict(rgba_grd=rnd(H, W, 4), dist_grd=rnd(H, W, 1))
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pos", "rot", "scl", "dns", "sph")}
    rays = {k: inp[k].clone().requires_grad_(True) for k in ("ray_ori", "ray_dir")}
    wrapper = FakeRaster(canned)
    poses = T.SensorPose3D(T_world_sensors=[torch.arange(7, dtype=torch.float32), torch.arange(7, dtype=torch.float32) + 10],
                           timestamps_us=[11, 29])
    sensor = ("sensor-params-object",)
    out = T.Tracer._Autograd.apply(wrapper, 42, 2, rays["ray_ori"], rays["ray_dir"], leaves["pos"], leaves["rot"], leaves["scl"],
                                   leaves["dns"], leaves["sph"], sensor, poses)
    assert len(out) == 4
    (out[0] * up["rgba_grd"]).sum().add((out[1] * up["dist_grd"]).sum()).backward()
    ta, ba = wrapper.trace_args, wrapper.bwd_args
    records["autograd"] = dict(
        n_trace_args=len(ta), n_bwd_args=len(ba),
        trace_args=[_describe(a) for a in ta], bwd_args=[_describe(a) for a in ba],
        frame_id=int(ta[0]), n_active_features=int(ta[1]), ts=[int(ta[8]), int(ta[9])],
        sensor_is_passed_through=bool(ta[7] is sensor and ba[7] is sensor),
        ray_grads_none=bool(rays["ray_ori"].grad is None and rays["ray_dir"].grad is None),
        ray_time_value=int(ta[6].reshape(-1)[0]),
    )
    for k, v in inp.items():
        arrays["ag_in_" + k] = _np(v)
    for k, v in canned.items():
        arrays["ag_canned_" + k] = _np(v)
    for k, v in up.items():
        arrays["ag_up_" + k] = _np(v)
    arrays["ag_particle_density"] = _np(ta[2])
    arrays["ag_particle_radiance"] = _np(ta[3])
    arrays["ag_pose_start"], arrays["ag_pose_end"] = _np(ta[10]), _np(ta[11])
    arrays["ag_bwd_rgba"], arrays["ag_bwd_rgba_grd"] = _np(ba[12]), _np(ba[13])
    arrays["ag_bwd_dist"], arrays["ag_bwd_dist_grd"] = _np(ba[14]), _np(ba[15])
    for k, v in leaves.items():
        arrays["ag_grad_" + k] = _np(v.grad)
    for k, v in zip(("rgba", "dist", "hits", "vis"), out):
        arrays["ag_out_" + k] = _np(v)

    # ---------------- B. __create_camera_parameters with a recording plugin
    plugin = FakePlugin()
    T._3dgut_plugin = plugin
    Batch = ref["protocols"].Batch
    DS = ref["camera_models"].ShutterType
    records["dataset_shutter_type"] = {m.name: int(m.value) for m in DS}
    records["plugin_shutter_type"] = {m.name: int(m.value) for m in PluginShutterType}
    create = getattr(T.Tracer, "_Tracer__create_camera_parameters")
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.tensor([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]])
    c2w[:3, 3] = torch.tensor([0.5, -1.25, 2.0])
    dummy = torch.zeros(1, 2, 2, 3)
    cases = []
    for fx, fy, cx, cy in ((1111.111, 1111.111, 400.0, 400.0), (100.0, 110.0, 64.0, 48.0), (523.7, 519.2, 320.5, 239.75)):
        cases.append(("list", dict(intrinsics=[fx, fy, cx, cy])))
    for st in DS:
        cases.append(("pinhole_" + st.name, dict(intrinsics_OpenCVPinholeCameraModelParameters=dict(
            resolution=np.array([1237, 822], np.int64), shutter_type=st, principal_point=np.array([618.5, 411.0], np.float32),
            focal_length=np.array([1040.5, 1041.75], np.float32),
            radial_coeffs=np.array([0.01, -0.02, 0.003, 0.0, 0.001, -0.0002], np.float32),
            tangential_coeffs=np.array([1e-3, -2e-3], np.float32), thin_prism_coeffs=np.array([1e-4, 2e-4, -3e-4, 4e-4], np.float32)))))
    for st in (DS.GLOBAL, DS.ROLLING_LEFT_TO_RIGHT):
        cases.append(("fisheye_" + st.name, dict(intrinsics_OpenCVFisheyeCameraModelParameters=dict(
            resolution=np.array([1752, 1168], np.int64), shutter_type=st, principal_point=np.array([876.0, 584.0], np.float32),
            focal_length=np.array([791.5, 792.25], np.float32), radial_coeffs=np.array([0.01, -0.002, 0.0003, -0.00004], np.float32),
            max_angle=1.329))))
    # the int values of the dataset enum hash like the members, so the reference's dict lookup accepts them as well
    cases.append(("pinhole_int5", dict(intrinsics_OpenCVPinholeCameraModelParameters=dict(
        resolution=np.array([64, 48], np.int64), shutter_type=5, principal_point=np.array([32.0, 24.0], np.float32),
        focal_length=np.array([60.0, 61.0], np.float32), radial_coeffs=np.zeros(6, np.float32),
        tangential_coeffs=np.zeros(2, np.float32), thin_prism_coeffs=np.zeros(4, np.float32)))))
    cam_records = []
    for name, kw in cases:
        b = Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None], **kw)
        n0 = len(plugin.camera_calls)
        params, pose = create(b)
        assert len(plugin.camera_calls) == n0 + 1 and params[1] == n0
        rec = dict(plugin.camera_calls[n0])
        rec["case"] = name
        rec["pose_start"] = [float(v) for v in pose.T_world_sensors[0]]
        rec["pose_end"] = [float(v) for v in pose.T_world_sensors[1]]
        rec["timestamps_us"] = list(pose.timestamps_us)
        if "intrinsics" in kw:
            rec["input_intrinsics"] = [float(v) for v in kw["intrinsics"]]
        cam_records.append(rec)
    # values outside the dataset enum are rejected by the reference's mapping
    rejected = {}
    for bad in (0, 6):
        try:
            create(Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None], intrinsics_OpenCVPinholeCameraModelParameters=dict(
                cases[3][1]["intrinsics_OpenCVPinholeCameraModelParameters"], shutter_type=bad)))
            rejected[str(bad)] = "accepted"
        except Exception as e:  # noqa: BLE001
            rejected[str(bad)] = type(e).__name__
    try:
        create(Batch(rays_ori=dummy, rays_dir=dummy, T_to_world=c2w[None]))
        rejected["no_intrinsics"] = "accepted"
    except Exception as e:  # noqa: BLE001
        rejected["no_intrinsics"] = type(e).__name__
    records["create_camera_parameters"] = dict(c2w=[float(v) for v in c2w.reshape(-1)], cases=cam_records, rejected=rejected)

    # ---------------- C. Tracer.render on CPU tensors with the fakes
    class FakeGaussians:
        num_gaussians = N
        n_active_features = 2
        positions = inp["pos"]
        def get_rotation(self): return inp["rot"]
        def get_scale(self): return inp["scl"]
        def get_density(self): return inp["dns"]
        def get_features(self): return inp["sph"]
        def background(self, T_to_world, rays_d, rgb, opacity, train):
            self.bg_args = (T_to_world, rays_d, rgb, opacity, train)
            return rgb + 0.25 * (1.0 - opacity), opacity
    tr = T.Tracer.__new__(T.Tracer)
    tr.tracer_wrapper = FakeRaster(canned)
    gs = FakeGaussians()
    batch = Batch(rays_ori=inp["ray_ori"], rays_dir=inp["ray_dir"], T_to_world=c2w[None], intrinsics=[100.0, 110.0, 3.5, 2.5])
    render_ok = True
    try:
        outd = tr.render(gs, batch, train=True, frame_id=9)
    except Exception as e:  # e.g. torch.cuda.nvtx without a GPU build
        render_ok = False
        records["render"] = dict(ok=False, error=f"{type(e).__name__}: {e}")
    if render_ok:
        records["render"] = dict(ok=True, keys=list(outd.keys()),
                                 shapes={k: (list(v.shape) if isinstance(v, torch.Tensor) else None) for k, v in outd.items()},
                                 frame_time_ms=float(outd["frame_time_ms"]), frame_id=int(tr.tracer_wrapper.trace_args[0]),
                                 background_train=bool(gs.bg_args[4]))
        for k, v in outd.items():
            if isinstance(v, torch.Tensor):
                arrays["render_" + k] = _np(v)

    # ---------------- D. datasets/utils.py
    U = ref["utils"]
    for tag, (w, h, fx, fy) in dict(a=(13, 9, 20.0, 21.5), b=(64, 48, 57.25, 55.5)).items():
        u = np.tile(np.arange(w), h)
        v = np.arange(h).repeat(w)
        ro, rd = U.pinhole_camera_rays(u, v, fx, fy, w, h, None)
        arrays[f"rays_{tag}_whff"] = np.array([w, h, fx, fy], np.float64)
        arrays[f"rays_{tag}_ori"] = np.asarray(ro, np.float64)
        arrays[f"rays_{tag}_dir"] = np.asarray(rd, np.float64)
    mr = []
    for (w, h, cx, cy) in ((1752, 1168, 876.0, 584.0), (1752, 1168, 900.5, 500.25), (640, 480, 100.0, 470.0)):
        mr.append([w, h, cx, cy, U.compute_max_radius(np.array([w, h], np.float64), np.array([cx, cy], np.float32))])
    arrays["max_radius_cases"] = np.array(mr, np.float64)
    cc = rng.normal(size=(9, 3))
    center, diag = U.get_center_and_diag(cc)
    arrays["center_diag_in"], arrays["center_diag_center"], arrays["center_diag_diag"] = cc, np.asarray(center), np.float64(diag)

    out_bin, out_txt = os.path.join(HERE, "colmap_model", "bin"), os.path.join(HERE, "colmap_model", "txt")
    write_colmap_model(out_bin, out_txt, rng)
    col = {}
    for fmt, d, ri, re_, rp in (("bin", out_bin, U.read_colmap_intrinsics_binary, U.read_colmap_extrinsics_binary, U.read_colmap_points3D_binary),
                                ("txt", out_txt, U.read_colmap_intrinsics_text, U.read_colmap_extrinsics_text, U.read_colmap_points3D_text)):
        ext = "bin" if fmt == "bin" else "txt"
        intr = ri(os.path.join(d, "cameras." + ext))
        extr = re_(os.path.join(d, "images." + ext))
        xyz, rgb, err = rp(os.path.join(d, "points3D." + ext))
        intr_l = list(intr.values()) if isinstance(intr, dict) else list(intr)
        extr_l = list(extr.values()) if isinstance(extr, dict) else list(extr)
        col[fmt] = dict(
            cameras=[dict(id=int(c.id), model=str(c.model), width=int(c.width), height=int(c.height),
                          params=[float(p) for p in c.params]) for c in intr_l],
            images=[dict(id=int(i.id), qvec=[float(p) for p in i.qvec], tvec=[float(p) for p in i.tvec], camera_id=int(i.camera_id),
                         name=str(i.name)) for i in extr_l])
        arrays[f"colmap_{fmt}_xyz"] = np.asarray(xyz, np.float64)
        arrays[f"colmap_{fmt}_rgb"] = np.asarray(rgb, np.float64)
        arrays[f"colmap_{fmt}_err"] = np.asarray(err, np.float64)
    records["colmap"] = col
    qs = rng.normal(size=(5, 4))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    arrays["qvec_in"] = qs
    arrays["qvec_so3"] = np.stack([U.qvec_to_so3(q) for q in qs])

    # ---------------- E. camera_models.py
    CM = ref["camera_models"]
    for tag, radial in (("zero", np.zeros(4, np.float32)), ("dist", np.array([0.01, -0.002, 0.0003, -0.00004], np.float32))):
        w, h = 36, 24
        fx, fy, cx, cy = 16.5, 16.25, 18.0, 12.0
        res = np.array([w, h]).astype(np.int64)
        pp = np.array([cx, cy], np.float32)
        fl = np.array([fx, fy], np.float32)
        max_radius = U.compute_max_radius(res.astype(np.float64), pp)
        max_angle = np.max([2.0 * max_radius / fl[0], 2.0 * max_radius / fl[1]]) / 2.0   # dataset_colmap.py:167-172
        params = CM.OpenCVFisheyeCameraModelParameters(principal_point=pp, focal_length=fl, radial_coeffs=radial, resolution=res,
                                                       max_angle=max_angle, shutter_type=CM.ShutterType.GLOBAL)
        u = np.tile(np.arange(w), h)
        v = np.arange(h).repeat(w)
        pix = torch.tensor(np.stack([u, v], axis=1), dtype=torch.int32)
        rd = CM.image_points_to_camera_rays(params, CM.pixels_to_image_points(pix))
        arrays[f"fisheye_{tag}_params"] = np.array([w, h, fx, fy, cx, cy, float(max_angle)], np.float64)
        arrays[f"fisheye_{tag}_radial"] = radial
        arrays[f"fisheye_{tag}_dir"] = _np(rd).reshape(1, h, w, 3)

    # ---------------- F. utils/render.py
    R = ref["render"]
    arrays["sh_C0"], arrays["sh_C1"] = np.float64(R.C0), np.float64(R.C1)
    arrays["sh_C2"], arrays["sh_C3"] = np.array(R.C2, np.float64), np.array(R.C3, np.float64)
    x = np.linspace(-0.3, 1.2, 7)
    arrays["rgb2sh_in"], arrays["rgb2sh_out"], arrays["sh2rgb_out"] = x, R.RGB2SH(x), R.SH2RGB(x)

    # ---------------- G. utils/misc.py
    M = ref["misc"]
    sched = M.get_scheduler("exp")(lr_init=0.00016 * 4.5, lr_final=0.0000016 * 4.5, max_steps=30000)
    steps = np.array([0, 1, 2, 10, 999, 1000, 15000, 29999, 30000, 40000], np.int64)
    arrays["sched_steps"], arrays["sched_lr"] = steps, np.array([sched(int(s)) for s in steps], np.float64)
    arrays["sched_args"] = np.array([0.00016 * 4.5, 0.0000016 * 4.5, 30000], np.float64)
    records["skip_scheduler_returns_none"] = bool(M.get_scheduler("skip")()(5) is None)
    csc = [(s, 0, 1e6, 1000) for s in (0, 1, 999, 1000, 1001, 2000, 3000, 4000)] + [(300, 500, 15000, 300), (600, 500, 15000, 300),
                                                                                    (15000, 500, 15000, 300), (900, 500, -1, 300)]
    records["check_step_condition"] = [dict(args=list(a), result=bool(M.check_step_condition(*a))) for a in csc]
    records["sh_degree_to_num_features"] = {str(d): int(M.sh_degree_to_num_features(d)) for d in range(4)}
    records["sh_degree_to_specular_dim"] = {str(d): int(M.sh_degree_to_specular_dim(d)) for d in range(4)}

    np.savez_compressed(os.path.join(HERE, "host_golden.npz"), **arrays)
    with open(os.path.join(HERE, "host_golden.json"), "w") as f:
        json.dump(records, f, indent=1, sort_keys=True)
    print("wrote host_golden.npz (%d arrays), host_golden.json, colmap_model/" % len(arrays))
    print("render:", records["render"] if not records["render"].get("ok") else "ok")


if __name__ == "__main__":
    main()
