"""MI355X-native 3DGUT tracer behind the reference's plugin surface.

Mirrors, name for name, what the rest of 3dgrut sees of `threedgut_tracer`:

  * `Tracer(conf)`, `.render(gaussians, gpu_batch, train=False, frame_id=0) -> dict`, `.timings`,
    `.build_acc(...)` (no-op) and the inner `Tracer._Autograd` torch.autograd.Function
    (reference: threedgut_tracer/tracer.py:158-351);
  * the native module surface the reference binds with pybind (bindings.cpp:79-113):
    `SplatRaster(config).trace / .trace_bwd / .collect_times`, `ShutterType`,
    `fromOpenCVPinholeCameraModelParameters`, `fromOpenCVFisheyeCameraModelParameters`
    — implemented here over the C ABI of libgut_hip.so (include/gut_hip.h) with ctypes.

Host code is Python on PyTorch-ROCm; PyTorch only provides device memory and the stream.  There is no
CPU fallback: without the HIP library every call raises.
"""
import ctypes as C
import enum
import math
import os

import numpy as np
import torch

from . import _capi
from .cameras import ShutterType as DatasetShutterType
from .pose import SensorPose3D, sensor_pose_from_c2w


# ----------------------------------------------------------------------------------------------------
# native-module surface (lib3dgut_cc equivalent)
# ----------------------------------------------------------------------------------------------------
class ShutterType(enum.IntEnum):  # the PLUGIN enum, bindings.cpp:87-92 / sensors/cameraModels.h:34-40
    ROLLING_TOP_TO_BOTTOM = 0
    ROLLING_LEFT_TO_RIGHT = 1
    ROLLING_BOTTOM_TO_TOP = 2
    ROLLING_RIGHT_TO_LEFT = 3
    GLOBAL = 4


# dataset enum (1..5) -> plugin enum (0..4), by name: threedgut_tracer/tracer.py:365-371 (SHUTTER_TYPE_MAP)
SHUTTER_TYPE_MAP = {d: ShutterType[d.name] for d in DatasetShutterType}


def plugin_shutter_type(value):
    """K["shutter_type"] of a dataset batch -> plugin ShutterType.  Accepts the dataset enum member, its int value
    1..5 (IntEnum members hash like their ints, so the reference's dict lookup accepts both too) or the member name
    (what a JSON round trip of `to_dict()` leaves behind).  Anything else raises KeyError like the reference's lookup."""
    if isinstance(value, ShutterType):
        raise KeyError(f"shutter_type {value!r} is already the plugin enum; a Batch carries the dataset enum (1..5)")
    if isinstance(value, str):
        name = value.split(".")[-1]
        if name not in DatasetShutterType.__members__:
            raise KeyError(f"unknown shutter_type {value!r}")
        return ShutterType[name]
    v = value.item() if hasattr(value, "item") else value
    if isinstance(v, bool) or not isinstance(v, int):
        raise KeyError(f"unknown shutter_type {value!r}")
    try:
        return SHUTTER_TYPE_MAP[DatasetShutterType(v)]
    except ValueError:
        raise KeyError(f"unknown shutter_type {value!r} (dataset enum is 1..5, camera_models.py:29-36)") from None


class CameraModelParameters:
    """POD camera intrinsics (sensors/cameraModels.h:22-58)."""

    def __init__(self):
        self.cam = _capi.GutCamera()
        self.cam.shutter = int(ShutterType.GLOBAL)
        self.cam.model = -1  # EmptyModel


def _arr(x, n, what):
    a = np.asarray(x, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise RuntimeError(f"[3dgut] {what}: expected {n} values, got {a.size}")
    return [float(v) for v in a]


def fromOpenCVPinholeCameraModelParameters(resolution, shutter_type, principal_point, focal_length, radial_coeffs,
                                           tangential_coeffs, thin_prism_coeffs):
    p = CameraModelParameters()
    p.cam.model = _capi.CAMERA_PINHOLE
    p.cam.shutter = int(shutter_type)
    p.cam.principal_point[:] = _arr(principal_point, 2, "principal_point")
    p.cam.focal_length[:] = _arr(focal_length, 2, "focal_length")
    p.cam.radial_coeffs[:] = _arr(radial_coeffs, 6, "radial_coeffs")
    p.cam.tangential_coeffs[:] = _arr(tangential_coeffs, 2, "tangential_coeffs")
    p.cam.thin_prism_coeffs[:] = _arr(thin_prism_coeffs, 4, "thin_prism_coeffs")
    return p


def fromOpenCVFisheyeCameraModelParameters(resolution, shutter_type, principal_point, focal_length, radial_coeffs,
                                           max_angle):
    p = CameraModelParameters()
    p.cam.model = _capi.CAMERA_FISHEYE
    p.cam.shutter = int(shutter_type)
    p.cam.principal_point[:] = _arr(principal_point, 2, "principal_point")
    p.cam.focal_length[:] = _arr(focal_length, 2, "focal_length")
    p.cam.radial_coeffs[:] = _arr(radial_coeffs, 4, "radial_coeffs") + [0.0, 0.0]
    p.cam.max_angle = float(max_angle)
    return p


def _conf_get(conf, path, default=None):
    """conf may be an OmegaConf node, a plain nested dict or any attribute namespace."""
    node = conf
    for key in path.split("."):
        if node is None:
            return default
        nxt = None
        if isinstance(node, dict):
            nxt = node.get(key, None)
        else:
            nxt = getattr(node, key, None)
            if nxt is None and hasattr(node, "__getitem__"):
                try:
                    nxt = node[key]
                except (KeyError, TypeError, IndexError):
                    nxt = None
        node = nxt
    return default if node is None else node


def config_from_conf(conf) -> "_capi.GutConfig":
    """conf.render.* -> GutConfig.  The reference turns these into -D defines (setup_3dgut.py:47-70)."""
    cfg = _capi.GutConfig()
    _capi.load().gut_default_config(C.byref(cfg))
    if conf is None:
        return cfg
    g = lambda p, d: _conf_get(conf, "render." + p, d)
    cfg.enable_kernel_timings = int(bool(g("enable_kernel_timings", False)))
    cfg.particle_radiance_sph_degree = int(g("particle_radiance_sph_degree", 3))
    cfg.particle_kernel_degree = int(g("particle_kernel_degree", 2))
    cfg.particle_kernel_min_response = float(g("particle_kernel_min_response", 0.0113))
    cfg.particle_kernel_min_alpha = float(g("particle_kernel_min_alpha", 1.0 / 255.0))
    cfg.particle_kernel_max_alpha = float(g("particle_kernel_max_alpha", 0.99))
    cfg.min_transmittance = float(g("min_transmittance", 0.0001))
    cfg.enable_hitcounts = int(bool(g("enable_hitcounts", True)))
    cfg.n_rolling_shutter_iterations = int(g("splat.n_rolling_shutter_iterations", 5))
    cfg.k_buffer_size = int(g("splat.k_buffer_size", 0))
    cfg.global_z_order = int(bool(g("splat.global_z_order", True)))
    cfg.ut_alpha = float(g("splat.ut_alpha", 1.0))
    cfg.ut_beta = float(g("splat.ut_beta", 2.0))
    cfg.ut_kappa = float(g("splat.ut_kappa", 0.0))
    cfg.ut_in_image_margin_factor = float(g("splat.ut_in_image_margin_factor", 0.1))
    cfg.ut_require_all_sigma_points = int(bool(g("splat.ut_require_all_sigma_points_valid", False)))
    cfg.rect_bounding = int(bool(g("splat.rect_bounding", True)))
    cfg.tight_opacity_bounding = int(bool(g("splat.tight_opacity_bounding", True)))
    cfg.tile_based_culling = int(bool(g("splat.tile_based_culling", True)))
    return cfg


def _has_reference_activations(gaussians):
    """True for a model built like the reference's MixtureOfGaussians under configs/base_gs.yaml:54-55 — nn.Parameters `rotation`,
    `scale`, `density` [N,4] / [N,3] / [N,1] whose activation callables are exactly torch's normalize / exp / sigmoid
    (threedgrut/model/model.py:163-167, utils/misc.py:45-50) — so that the library may apply them itself."""
    F = torch.nn.functional
    try:
        return (gaussians.rotation_activation is F.normalize and gaussians.scale_activation is torch.exp
                and gaussians.density_activation is torch.sigmoid
                and gaussians.rotation.dim() == 2 and gaussians.rotation.shape[1] == 4 and gaussians.scale.shape[1] == 3
                and gaussians.density.shape[1] == 1 and gaussians.rotation.dtype == torch.float32)
    except AttributeError:
        return False


def _check_f32_cuda(t, name, shape_tail=None):
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise RuntimeError(f"[3dgut] {name}: expected a float32 tensor")  # cf. voidDataPtr, splatRaster.cpp:68-90
    if not t.is_cuda:
        raise RuntimeError(f"[3dgut] {name}: expected a GPU tensor (there is no CPU path)")
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise RuntimeError(f"[3dgut] {name}: expected trailing shape {shape_tail}, got {tuple(t.shape)}")
    return t.contiguous()


class SplatRaster:
    """Drop-in for the pybind class of the same name (splatRaster.h:47-89)."""

    def __init__(self, config=None, device_index=None):
        self._lib = _capi.load()
        self._handle = C.c_void_p()
        cfg = config if isinstance(config, _capi.GutConfig) else config_from_conf(config)
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device_index = int(device_index)
        self.enable_kernel_timings = bool(cfg.enable_kernel_timings)
        # PARTICLE_RADIANCE_NUM_COEFFS = (degree + 1)^2 (setup_3dgut.py:48): the radiance rows are [N, 3 (degree + 1)^2]
        self.sph_degree = int(cfg.particle_radiance_sph_degree)
        self._radiance_width = 3 * (self.sph_degree + 1) ** 2
        _capi.check(self._lib.gut_create(C.byref(cfg), self.device_index, C.byref(self._handle)), "SplatRaster()")
        self._timings = {}
        # extension key (not in the reference's configs): render.splat.sorted_reference_backward, default true = the reference's
        # own backward of the sorted variant; false = the exact derivative of the forward (DESIGN.md §3 deviation 6)
        if not isinstance(config, _capi.GutConfig) and not bool(_conf_get(config, "render.splat.sorted_reference_backward", True)):
            self.set_sorted_reference_backward(False)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and self._handle.value:
                self._lib.gut_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def _camera(sensor_params, ts_start, ts_end, pose_start, pose_end):
        if not isinstance(sensor_params, CameraModelParameters):
            raise RuntimeError("[3dgut] sensor_params must come from fromOpenCV*CameraModelParameters")
        cam = _capi.GutCamera()
        C.memmove(C.byref(cam), C.byref(sensor_params.cam), C.sizeof(cam))
        ps = torch.as_tensor(pose_start).detach().cpu().to(torch.float32).reshape(-1)  # toSensorState, splatRaster.cpp:92-100
        pe = torch.as_tensor(pose_end).detach().cpu().to(torch.float32).reshape(-1)
        if ps.numel() != 7 or pe.numel() != 7:
            raise RuntimeError("[3dgut] sensor poses must be [t(3), q_xyzw(4)]")
        cam.pose_start[:] = ps.tolist()
        cam.pose_end[:] = pe.tolist()
        cam.timestamp_start_us = int(ts_start)
        cam.timestamp_end_us = int(ts_end)
        return cam

    def trace(self, frame_number, num_active_features, particle_density, particle_radiance, ray_ori, ray_dir, ray_time,
              sensor_params, ts_start, ts_end, pose_start, pose_end):
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        if ray_ori.dim() != 4 or ray_ori.shape[0] != 1:
            raise RuntimeError("[3dgut] rays must be [1,H,W,3] (the reference renders one view per call)")
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(particle_density.shape[0])
        dev = ray_ori.device
        if n:
            particle_density = _check_f32_cuda(particle_density, "particleDensity", (12,))
            particle_radiance = _check_f32_cuda(particle_radiance, "particleRadiance", (self._radiance_width,))
        opts = dict(dtype=torch.float32, device=dev)
        rgba = torch.empty((H, W, 4), **opts)   # fully written by the kernels (no zero-fill passes)
        dist = torch.empty((H, W, 1), **opts)
        hits = torch.empty((H, W, 1), **opts)
        vis = torch.empty((n, 1), **opts)
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream  # splatRaster.cpp:186-187
        with torch.cuda.device(dev):
            rc = self._lib.gut_trace(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF,
                                     int(num_active_features), n,
                                     particle_density.data_ptr() if n else None, particle_radiance.data_ptr() if n else None,
                                     W, H, ray_ori.data_ptr(), ray_dir.data_ptr(), C.byref(cam),
                                     rgba.data_ptr(), dist.data_ptr(), hits.data_ptr(), vis.data_ptr() if n else None)
        _capi.check(rc, "trace")
        return rgba, dist, hits, vis

    def trace_bwd(self, frame_number, num_active_features, particle_density, particle_radiance, ray_ori, ray_dir, ray_time,
                  sensor_params, ts_start, ts_end, pose_start, pose_end, ray_radiance_density, ray_radiance_density_grd,
                  ray_hit_distance, ray_hit_distance_grd, raw_parameter_grads=False, compact_radiance_grads=False, out=None,
                  skip_epilogue=False):
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(particle_density.shape[0])
        dev = ray_ori.device
        rgba = _check_f32_cuda(ray_radiance_density, "rayRadianceDensity", (4,))
        rgba_g = _check_f32_cuda(ray_radiance_density_grd, "rayRadianceDensityGradient", (4,))
        dist = _check_f32_cuda(ray_hit_distance, "rayHitDistance")
        dist_g = None if ray_hit_distance_grd is None else _check_f32_cuda(ray_hit_distance_grd, "rayHitDistanceGradient")
        opts = dict(dtype=torch.float32, device=dev)
        if skip_epilogue:   # gradient rows stay in the handle for optimize_after_bwd (native trainer, one view)
            dens_g = sph_g = None
        elif out is not None:
            dens_g, sph_g = out
        else:
            dens_g = torch.empty((n, 12), **opts)  # fully written by the per-Gaussian epilogue kernel
            sph_g = torch.empty((n, 3 if compact_radiance_grads else self._radiance_width), **opts)
        if n:
            particle_density = _check_f32_cuda(particle_density, "particleDensity", (12,))
            particle_radiance = _check_f32_cuda(particle_radiance, "particleRadiance", (self._radiance_width,))
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = self._lib.gut_trace_bwd_ex(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF,
                                         int(num_active_features), n,
                                         particle_density.data_ptr() if n else None,
                                         particle_radiance.data_ptr() if n else None, W, H, ray_ori.data_ptr(),
                                         ray_dir.data_ptr(), C.byref(cam), rgba.data_ptr(), rgba_g.data_ptr(),
                                         dist.data_ptr(), None if dist_g is None else dist_g.data_ptr(),
                                         dens_g.data_ptr() if (n and dens_g is not None) else None,
                                         sph_g.data_ptr() if (n and sph_g is not None) else None,
                                         (_capi.BWD_RAW_PARAMETER_GRADS if raw_parameter_grads else 0) |
                                         (_capi.BWD_COMPACT_RADIANCE_GRADS if compact_radiance_grads else 0) |
                                         (_capi.BWD_SKIP_EPILOGUE if skip_epilogue else 0))
        _capi.check(rc, "trace_bwd")
        return dens_g, sph_g

    def trace_fields(self, frame_number, num_active_features, mog_pos, mog_dns, mog_rot, mog_scl, particle_radiance, ray_ori, ray_dir,
                     sensor_params, ts_start, ts_end, pose_start, pose_end):
        """trace() on the four activated tensors Tracer.render hands to _Autograd (positions [N,3], density [N,1], rotation [N,4],
        scale [N,3]) instead of their [N,12] concatenation: gut_trace_fields packs the rows inside the library (extension of the
        reference's surface; a wrapper without it gets the reference's torch.cat + trace())."""
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        if ray_ori.dim() != 4 or ray_ori.shape[0] != 1:
            raise RuntimeError("[3dgut] rays must be [1,H,W,3] (the reference renders one view per call)")
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(mog_pos.shape[0])
        dev = ray_ori.device
        if n:
            mog_pos = _check_f32_cuda(mog_pos, "positions", (3,)); mog_dns = _check_f32_cuda(mog_dns, "density", (1,))
            mog_rot = _check_f32_cuda(mog_rot, "rotation", (4,)); mog_scl = _check_f32_cuda(mog_scl, "scale", (3,))
            particle_radiance = _check_f32_cuda(particle_radiance, "particleRadiance", (self._radiance_width,))
        opts = dict(dtype=torch.float32, device=dev)
        rgba, dist, hits, vis = torch.empty((H, W, 4), **opts), torch.empty((H, W, 1), **opts), torch.empty((H, W, 1), **opts), torch.empty((n, 1), **opts)
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: t.data_ptr() if n else None
        with torch.cuda.device(dev):
            rc = self._lib.gut_trace_fields(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF, int(num_active_features), n,
                                            ptr(mog_pos), ptr(mog_dns), ptr(mog_rot), ptr(mog_scl), ptr(particle_radiance), W, H,
                                            ray_ori.data_ptr(), ray_dir.data_ptr(), C.byref(cam), rgba.data_ptr(), dist.data_ptr(),
                                            hits.data_ptr(), ptr(vis))
        _capi.check(rc, "trace_fields")
        return rgba, dist, hits, vis

    def trace_bwd_fields(self, frame_number, num_active_features, num_particles, particle_radiance, ray_ori, ray_dir, sensor_params, ts_start,
                         ts_end, pose_start, pose_end, ray_radiance_density, ray_radiance_density_grd, ray_hit_distance,
                         ray_hit_distance_grd):
        """Backward of trace_fields: returns (positions, density, rotation, scale, radiance) gradients as five fresh tensors —
        what _Autograd.backward hands to autograd, with no [N,12] intermediate, split or copies."""
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(num_particles)
        dev = ray_ori.device
        rgba = _check_f32_cuda(ray_radiance_density, "rayRadianceDensity", (4,))
        rgba_g = _check_f32_cuda(ray_radiance_density_grd, "rayRadianceDensityGradient", (4,))
        dist = _check_f32_cuda(ray_hit_distance, "rayHitDistance")
        dist_g = None if ray_hit_distance_grd is None else _check_f32_cuda(ray_hit_distance_grd, "rayHitDistanceGradient")
        opts = dict(dtype=torch.float32, device=dev)
        pos_g, dns_g, rot_g, scl_g = (torch.empty((n, c), **opts) for c in (3, 1, 4, 3))
        sph_g = torch.empty((n, self._radiance_width), **opts)
        if n:
            particle_radiance = _check_f32_cuda(particle_radiance, "particleRadiance", (self._radiance_width,))
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: t.data_ptr() if n else None
        with torch.cuda.device(dev):
            rc = self._lib.gut_trace_bwd_fields(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF, int(num_active_features), n,
                                                ptr(particle_radiance), W, H, ray_ori.data_ptr(), ray_dir.data_ptr(), C.byref(cam),
                                                rgba.data_ptr(), rgba_g.data_ptr(), dist.data_ptr(),
                                                None if dist_g is None else dist_g.data_ptr(), ptr(pos_g), ptr(dns_g), ptr(rot_g),
                                                ptr(scl_g), ptr(sph_g))
        _capi.check(rc, "trace_bwd_fields")
        return pos_g, dns_g, rot_g, scl_g, sph_g

    def trace_raw_model_fields(self, *args):
        """trace_model_fields() on the model's PRE-ACTIVATION density / rotation / scale tensors (gut_trace_raw_model_fields): the
        library applies sigmoid / normalize / exp itself and the following trace_bwd_model_fields returns raw-parameter gradients."""
        return self.trace_model_fields(*args, _raw=True)

    def trace_model_fields(self, frame_number, num_active_features, mog_pos, mog_dns, mog_rot, mog_scl, sph_albedo, sph_specular, ray_ori,
                           ray_dir, sensor_params, ts_start, ts_end, pose_start, pose_end, _raw=False):
        """trace_fields() with the SH coefficients as the model's two tensors (features_albedo [N,3], features_specular [N,45]; model.py:
        68-75) instead of get_features()'s [N,48] torch.cat (gut_trace_model_fields)."""
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        if ray_ori.dim() != 4 or ray_ori.shape[0] != 1:
            raise RuntimeError("[3dgut] rays must be [1,H,W,3] (the reference renders one view per call)")
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(mog_pos.shape[0])
        dev = ray_ori.device
        if n:
            mog_pos = _check_f32_cuda(mog_pos, "positions", (3,)); mog_dns = _check_f32_cuda(mog_dns, "density", (1,))
            mog_rot = _check_f32_cuda(mog_rot, "rotation", (4,)); mog_scl = _check_f32_cuda(mog_scl, "scale", (3,))
            sph_albedo = _check_f32_cuda(sph_albedo, "featuresAlbedo", (3,))
            sph_specular = _check_f32_cuda(sph_specular, "featuresSpecular", (45,))
            if sph_albedo.shape[0] != n or sph_specular.shape[0] != n:
                raise RuntimeError("[3dgut] feature tensors and positions differ in their number of rows")
        opts = dict(dtype=torch.float32, device=dev)
        rgba, dist, hits, vis = torch.empty((H, W, 4), **opts), torch.empty((H, W, 1), **opts), torch.empty((H, W, 1), **opts), torch.empty((n, 1), **opts)
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: t.data_ptr() if n else None
        entry = self._lib.gut_trace_raw_model_fields if _raw else self._lib.gut_trace_model_fields
        with torch.cuda.device(dev):
            rc = entry(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF, int(num_active_features), n,
                                                  ptr(mog_pos), ptr(mog_dns), ptr(mog_rot), ptr(mog_scl), ptr(sph_albedo), ptr(sph_specular),
                                                  W, H, ray_ori.data_ptr(), ray_dir.data_ptr(), C.byref(cam), rgba.data_ptr(),
                                                  dist.data_ptr(), hits.data_ptr(), ptr(vis))
        _capi.check(rc, "trace_model_fields")
        return rgba, dist, hits, vis

    def trace_bwd_model_fields(self, frame_number, num_active_features, num_particles, ray_ori, ray_dir, sensor_params, ts_start, ts_end,
                               pose_start, pose_end, ray_radiance_density, ray_radiance_density_grd, ray_hit_distance, ray_hit_distance_grd,
                               out=None):
        """Backward of trace_model_fields: (positions, density, rotation, scale, features_albedo, features_specular) gradients as six
        fresh tensors (or the six of `out`), each written in full by the kernels."""
        ray_ori = _check_f32_cuda(ray_ori, "rayOrigin", (3,))
        ray_dir = _check_f32_cuda(ray_dir, "rayDirection", (3,))
        H, W = int(ray_ori.shape[1]), int(ray_ori.shape[2])
        n = int(num_particles)
        dev = ray_ori.device
        rgba = _check_f32_cuda(ray_radiance_density, "rayRadianceDensity", (4,))
        rgba_g = _check_f32_cuda(ray_radiance_density_grd, "rayRadianceDensityGradient", (4,))
        dist = _check_f32_cuda(ray_hit_distance, "rayHitDistance")
        dist_g = None if ray_hit_distance_grd is None else _check_f32_cuda(ray_hit_distance_grd, "rayHitDistanceGradient")
        opts = dict(dtype=torch.float32, device=dev)
        if out is not None:
            pos_g, dns_g, rot_g, scl_g, alb_g, spec_g = (_check_f32_cuda(t, "gradient output", (c,)) for t, c in zip(out, (3, 1, 4, 3, 3, 45)))
            if any(t.shape[0] != n or not o.is_contiguous() for t, o in zip((pos_g, dns_g, rot_g, scl_g, alb_g, spec_g), out)):
                raise RuntimeError("[3dgut] gradient outputs must be contiguous [N,c] tensors with the forward's number of rows")
        else:
            pos_g, dns_g, rot_g, scl_g, alb_g, spec_g = (torch.empty((n, c), **opts) for c in (3, 1, 4, 3, 3, 45))
        cam = self._camera(sensor_params, ts_start, ts_end, pose_start, pose_end)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: t.data_ptr() if n else None
        with torch.cuda.device(dev):
            rc = self._lib.gut_trace_bwd_model_fields(self._handle, C.c_void_p(stream), int(frame_number) & 0xFFFFFFFF, int(num_active_features),
                                                      n, W, H, ray_ori.data_ptr(), ray_dir.data_ptr(), C.byref(cam), rgba.data_ptr(),
                                                      rgba_g.data_ptr(), dist.data_ptr(), None if dist_g is None else dist_g.data_ptr(),
                                                      ptr(pos_g), ptr(dns_g), ptr(rot_g), ptr(scl_g), ptr(alb_g), ptr(spec_g))
        _capi.check(rc, "trace_bwd_model_fields")
        return pos_g, dns_g, rot_g, scl_g, alb_g, spec_g

    def set_position_gradient_statistics(self, norm_accum, norm_denom):
        """The NEXT optimize_after_bwd also accumulates GSStrategy.update_gradient_buffer's statistics (gs.py:106-115) into these two
        [N,1] tensors (float32 / int32), from the gradient rows and pre-update positions it holds anyway."""
        if norm_accum is None:
            _capi.check(self._lib.gut_set_position_gradient_statistics(self._handle, None, None), "set_position_gradient_statistics")
            return
        if not (norm_accum.is_cuda and norm_accum.dtype == torch.float32 and norm_accum.is_contiguous()
                and norm_denom.is_cuda and norm_denom.dtype == torch.int32 and norm_denom.is_contiguous()
                and norm_accum.numel() == norm_denom.numel()):
            raise RuntimeError("[3dgut] statistics buffers: contiguous GPU tensors, float32 accum and int32 denom of equal length")
        self._stat_rows = int(norm_accum.numel())   # (checked against N by the caller: the handle's N is the forward's)
        _capi.check(self._lib.gut_set_position_gradient_statistics(self._handle, norm_accum.data_ptr(), norm_denom.data_ptr()),
                    "set_position_gradient_statistics")

    def optimize_after_bwd(self, num_active_features, camera_position, raw12, raw_m, raw_v, sh48, sh_m, sh_v, lr12, lr48, betas, eps,
                           step, visibility=None, act_out=None, lazy=None):
        """Per-Gaussian backward epilogue + SH-gradient rebuild + Adam in one pass (gut_optimize_after_bwd); follows a
        trace_bwd(..., skip_epilogue=True) on the same stream.  lr12 / lr48: float32 numpy arrays.  camera_position None: the sensor
        position of the cached forward, which the library keeps on the device (no host-to-device copy in the step)."""
        f32p = C.POINTER(C.c_float)
        stream = torch.cuda.current_stream(raw12.device).cuda_stream
        with torch.cuda.device(raw12.device):
            rc = self._lib.gut_optimize_after_bwd(self._handle, C.c_void_p(stream), int(num_active_features),
                                                  None if camera_position is None else camera_position.data_ptr(),
                                                  raw12.data_ptr(), raw_m.data_ptr(), raw_v.data_ptr(), sh48.data_ptr(), sh_m.data_ptr(),
                                                  sh_v.data_ptr(), lr12.ctypes.data_as(f32p), lr48.ctypes.data_as(f32p), betas[0],
                                                  betas[1], eps, int(step), None if visibility is None else visibility.data_ptr(),
                                                  None if act_out is None else act_out.data_ptr(), None if lazy is None else C.byref(lazy))
        _capi.check(rc, "optimize_after_bwd")

    def optimize_rows_without_gradient(self, raw12, raw_m, raw_v, sh48, sh_m, sh_v, lr12, lr48, betas, eps, step, act_out=None, lazy=None):
        """Adam step of the rows the projection gave no tile, on the handle's low-priority side stream, between trace and
        trace_bwd(..., skip_epilogue=True) (gut_optimize_rows_without_gradient); optimize_after_bwd must follow."""
        f32p = C.POINTER(C.c_float)
        stream = torch.cuda.current_stream(raw12.device).cuda_stream
        with torch.cuda.device(raw12.device):
            rc = self._lib.gut_optimize_rows_without_gradient(
                self._handle, C.c_void_p(stream), raw12.data_ptr(), raw_m.data_ptr(), raw_v.data_ptr(), sh48.data_ptr(),
                sh_m.data_ptr(), sh_v.data_ptr(), lr12.ctypes.data_as(f32p), lr48.ctypes.data_as(f32p), betas[0], betas[1], eps,
                int(step), None if act_out is None else act_out.data_ptr(), None if lazy is None else C.byref(lazy))
        _capi.check(rc, "optimize_rows_without_gradient")

    def finish_optimizer_step_without_gradient(self):
        """Ends a step that optimize_rows_without_gradient began and optimize_after_bwd cannot finish (something raised in
        between): the remaining waves take the same Adam step with a zero gradient and the handle accepts trace() again
        (gut_optimize_finish_without_gradient).  No-op when no step is half applied."""
        dev = torch.device("cuda", self.device_index)
        with torch.cuda.device(dev):
            rc = self._lib.gut_optimize_finish_without_gradient(self._handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "optimize_finish_without_gradient")

    def compact_gradient_rows(self, act12, records, count):
        """Per-Gaussian epilogue of trace_bwd(..., skip_epilogue=True) as a LIST: one 64-byte record per Gaussian with a
        non-zero gradient row (gut_compact_gradient_rows; layout in gut_hip.h).  records: float32 [>= N, 16] on the device,
        count: int32/uint32 device tensor with one element (receives the number of records)."""
        if records.shape[0] < act12.shape[0] or records.shape[1] != _capi.GRADIENT_RECORD_FLOATS or not records.is_contiguous():
            raise ValueError("records must be a contiguous [>= num_particles, 16] float32 tensor")
        stream = torch.cuda.current_stream(act12.device).cuda_stream
        with torch.cuda.device(act12.device):
            rc = self._lib.gut_compact_gradient_rows(self._handle, C.c_void_p(stream), act12.data_ptr(), records.data_ptr(),
                                                     int(records.shape[0]), count.data_ptr())
        _capi.check(rc, "compact_gradient_rows")

    def collect_times(self):
        f, b = C.c_float(-1.0), C.c_float(-1.0)
        _capi.check(self._lib.gut_collect_times(self._handle, C.byref(f), C.byref(b)), "collect_times")
        if f.value >= 0:
            self._timings["forward_render"] = f.value
        if b.value >= 0:
            self._timings["backward_render"] = b.value
        return dict(self._timings)

    # ---- extensions used by tests / bench (not part of the reference surface) ----
    def set_lazy_tile_order(self, on=True):
        """Tile-only radix grouping + per-tile lazy depth order in the forward compositor (default on; gut_hip.h)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_LAZY_TILE_ORDER, 1 if on else 0), "set_option")

    def set_early_extra_percent(self, percent):
        """Share of the row blocks in which the second side-stream optimiser launch also takes the waves that have tiles but
        hold no Gaussian the forward walked (GUT_OPT_EARLY_EXTRA_PERCENT, default 100; 0 = waves without tiles only)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_EARLY_EXTRA_PERCENT, int(percent)), "set_option")

    def set_forward_tile_order(self, mode):
        """Launch order of the forward compositor's tiles: 1 longest lists first, 0 image order, -1 (default) decided by the library from
        the share of their lists the last frames walked (GUT_OPT_FORWARD_TILE_ORDER, gut_hip.h)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_FORWARD_TILE_ORDER, int(mode)), "set_option")

    def set_kernel_timing_set(self, which):
        """With enable_kernel_timings: 0 = every kernel boundary of the following frames is bracketed by events, 2 = none, 1 = only the
        side-stream optimiser launches (GUT_OPT_KERNEL_TIMING_SET, gut_hip.h)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_KERNEL_TIMING_SET, int(which)), "set_option")

    def debug_replace_scratch(self, index):
        """Developer probe: move one of the handle's scratch buffers to a fresh allocation (GUT_OPT_DEBUG_REPLACE_SCRATCH, gut_hip.h)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_DEBUG_REPLACE_SCRATCH, int(index)), "set_option")

    def set_sorted_reference_backward(self, on=True):
        """Sorted variant: the reference's own (unclamped-colour undo) form of the alpha gradient (the default; off = the exact derivative of the forward; gut_hip.h)."""
        _capi.check(self._lib.gut_set_option(self._handle, _capi.OPT_SORTED_REFERENCE_BACKWARD, 1 if on else 0), "set_option")

    def stats(self):
        s = _capi.GutStats()
        _capi.check(self._lib.gut_get_stats(self._handle, C.byref(s)), "stats")
        return {k: int(getattr(s, k)) for k, _ in s._fields_ if k != "reserved"}

    def kernel_times(self):
        arr = (C.c_float * _capi.GUT_NUM_KERNEL_TIMERS)()
        _capi.check(self._lib.gut_kernel_times(self._handle, arr), "kernel_times")
        return dict(zip(_capi.KERNEL_TIMER_NAMES, [float(v) for v in arr]))

    def kernel_times_mean(self):
        arr = (C.c_float * _capi.GUT_NUM_KERNEL_TIMERS)()
        cnt = C.c_int32(0)
        _capi.check(self._lib.gut_kernel_times_mean(self._handle, arr, C.byref(cnt)), "kernel_times_mean")
        return dict(zip(_capi.KERNEL_TIMER_NAMES, [float(v) for v in arr])), int(cnt.value)

    def debug_buffer(self, name, device=None):
        """Copy of an intermediate buffer as a torch tensor (parity tests)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.gut_debug_buffer(self._handle, _capi.BUF[name], C.byref(ptr), C.byref(nbytes)), "debug_buffer")
        dt = {"tiles_count": torch.int32, "tiles_offset": torch.int32, "unsorted_ids": torch.int32,
              "sorted_ids": torch.int32, "ordered_ids": torch.int32, "tile_ranges": torch.int32, "tile_traversed_fwd": torch.int32,
              "tile_traversed_bwd": torch.int32, "unsorted_keys": torch.int64,
              "sorted_keys": torch.int64}.get(name, torch.float32)
        n = nbytes.value // (8 if dt == torch.int64 else 4)
        dev = torch.device("cuda", self.device_index) if device is None else device
        out = torch.empty(n, dtype=dt, device=dev)
        if n:
            _capi.check(self._lib.gut_debug_copy(self._handle, _capi.BUF[name], C.c_void_p(out.data_ptr()),
                                                 C.c_size_t(nbytes.value)), "debug_copy")
        return out


# ----------------------------------------------------------------------------------------------------
# Tracer (threedgut_tracer/tracer.py:158-431)
# ----------------------------------------------------------------------------------------------------
class Tracer:
    class _Autograd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, tracer_wrapper, frame_id, n_active_features, ray_ori, ray_dir, mog_pos, mog_rot, mog_scl,
                    mog_dns, mog_sph, sensor_params, sensor_poses, mog_sph_specular=None, raw_parameters=False):
            """The reference's twelve arguments (tracer.py:161-174); a thirteenth, mog_sph_specular, says that mog_sph is the model's
            features_albedo [N,3] and this its features_specular [N,45] (Tracer.render below, for models that expose the two); a
            fourteenth, raw_parameters, that mog_rot / mog_scl / mog_dns are the model's PRE-ACTIVATION tensors (un-normalised
            quaternion, log-scale, density logit) and the library applies normalize / exp / sigmoid itself."""
            ctx.frame_id = frame_id
            ctx.n_active_features = n_active_features
            ctx.sensor_params = sensor_params
            ctx.sensor_poses = sensor_poses
            ctx.tracer_wrapper = tracer_wrapper
            ctx.set_materialize_grads(False)  # an unused pred_dist arrives as None -> backward variant without dist terms
            ctx.model_fields = mog_sph_specular is not None
            if ctx.model_fields:
                # all six model tensors go to the library as they are (gut_trace_model_fields): no torch.cat of the features
                # either (1.0 ms per render at 6 M Gaussians), and six gradient tensors come back, nothing for autograd to split
                trace = tracer_wrapper.trace_raw_model_fields if raw_parameters else tracer_wrapper.trace_model_fields
                rgba, dist, hits, vis = trace(
                    frame_id, n_active_features, mog_pos, mog_dns, mog_rot, mog_scl, mog_sph.contiguous(), mog_sph_specular.contiguous(),
                    ray_ori.contiguous(), ray_dir.contiguous(), sensor_params, sensor_poses.timestamps_us[0],
                    sensor_poses.timestamps_us[1], sensor_poses.T_world_sensors[0], sensor_poses.T_world_sensors[1])
                ctx.fields = True
                ctx.num_particles = int(mog_pos.shape[0])
                ctx.save_for_backward(ray_ori, ray_dir, rgba, dist)
                ctx.mark_non_differentiable(hits, vis)
                return rgba, dist, hits, vis
            particle_radiance = mog_sph.contiguous()
            ctx.fields = hasattr(tracer_wrapper, "trace_fields")
            if ctx.fields:
                # this library's wrapper packs the [pos | density | quat | scale | 0] rows itself (gut_trace_fields) and returns the
                # four gradients as four tensors: no torch.cat here (0.43 ms at 6 M Gaussians), no split + four copies in backward
                rgba, dist, hits, vis = tracer_wrapper.trace_fields(
                    frame_id, n_active_features, mog_pos, mog_dns, mog_rot, mog_scl, particle_radiance, ray_ori.contiguous(),
                    ray_dir.contiguous(), sensor_params, sensor_poses.timestamps_us[0], sensor_poses.timestamps_us[1],
                    sensor_poses.T_world_sensors[0], sensor_poses.T_world_sensors[1])
                ctx.num_particles = int(mog_pos.shape[0])
                ctx.save_for_backward(ray_ori, ray_dir, rgba, dist, particle_radiance)
                ctx.mark_non_differentiable(hits, vis)
                return rgba, dist, hits, vis
            # the reference's own form, for any wrapper with the pybind module's surface (trace / trace_bwd on packed rows):
            # [pos(3) | density(1) | quat wxyz(4) | scale(3) | 0] rows, tracer.py:176-178
            particle_density = torch.cat([mog_pos, mog_dns, mog_rot, mog_scl, torch.zeros_like(mog_dns)], dim=1).contiguous()
            ray_time = None  # the reference allocates an int64 [1,H,W,1] tensor that no kernel reads (tracer.py:181-186)
            rgba, dist, hits, vis = tracer_wrapper.trace(
                frame_id, n_active_features, particle_density, particle_radiance, ray_ori.contiguous(),
                ray_dir.contiguous(), ray_time, sensor_params, sensor_poses.timestamps_us[0], sensor_poses.timestamps_us[1],
                sensor_poses.T_world_sensors[0], sensor_poses.T_world_sensors[1])
            ctx.save_for_backward(ray_ori, ray_dir, rgba, dist, particle_density, particle_radiance)
            ctx.mark_non_differentiable(hits, vis)
            return rgba, dist, hits, vis

        @staticmethod
        def backward(ctx, rgba_grd, dist_grd, hits_grd_unused, vis_grd_unused):
            poses = ctx.sensor_poses
            if ctx.model_fields:
                ray_ori, ray_dir, rgba, dist = ctx.saved_tensors
                if rgba_grd is None:
                    rgba_grd = torch.zeros_like(rgba)
                pos_g, dns_g, rot_g, scl_g, alb_g, spec_g = ctx.tracer_wrapper.trace_bwd_model_fields(
                    ctx.frame_id, ctx.n_active_features, ctx.num_particles, ray_ori, ray_dir, ctx.sensor_params, poses.timestamps_us[0],
                    poses.timestamps_us[1], poses.T_world_sensors[0], poses.T_world_sensors[1], rgba, rgba_grd.contiguous(), dist,
                    None if dist_grd is None else dist_grd.contiguous())
                return (None, None, None, None, None, pos_g, rot_g, scl_g, dns_g, alb_g, None, None, spec_g, None)
            if ctx.fields:
                ray_ori, ray_dir, rgba, dist, particle_radiance = ctx.saved_tensors
                if rgba_grd is None:
                    rgba_grd = torch.zeros_like(rgba)
                pos_g, dns_g, rot_g, scl_g, sph_grd = ctx.tracer_wrapper.trace_bwd_fields(
                    ctx.frame_id, ctx.n_active_features, ctx.num_particles, particle_radiance, ray_ori, ray_dir, ctx.sensor_params,
                    poses.timestamps_us[0], poses.timestamps_us[1], poses.T_world_sensors[0], poses.T_world_sensors[1], rgba,
                    rgba_grd.contiguous(), dist, None if dist_grd is None else dist_grd.contiguous())
                return (None, None, None, None, None, pos_g, rot_g, scl_g, dns_g, sph_grd, None, None, None, None)
            ray_ori, ray_dir, rgba, dist, particle_density, particle_radiance = ctx.saved_tensors
            if rgba_grd is None:
                rgba_grd = torch.zeros_like(rgba)
            dens_grd, sph_grd = ctx.tracer_wrapper.trace_bwd(
                ctx.frame_id, ctx.n_active_features, particle_density, particle_radiance, ray_ori, ray_dir, None,
                ctx.sensor_params, poses.timestamps_us[0], poses.timestamps_us[1], poses.T_world_sensors[0],
                poses.T_world_sensors[1], rgba, rgba_grd.contiguous(), dist, None if dist_grd is None else dist_grd.contiguous())
            pos_g, dns_g, rot_g, scl_g, _ = torch.split(dens_grd, [3, 1, 4, 3, 1], dim=1)  # tracer.py:268-270
            return (None, None, None, None, None, pos_g.contiguous(), rot_g.contiguous(), scl_g.contiguous(),
                    dns_g.contiguous(), sph_grd, None, None, None, None)

    def __init__(self, conf):
        self.device = "cuda"
        self.conf = conf
        torch.zeros(1, device=self.device)  # force context creation (tracer.py:292)
        self.tracer_wrapper = SplatRaster(conf)
        self.split_features = True   # False: always go through gaussians.get_features() (the reference's call, kept for A/B tests)
        # True: a model whose activations ARE torch.sigmoid / F.normalize / torch.exp (the reference's MixtureOfGaussians under
        # configs/base_gs.yaml) hands its pre-activation tensors over and the library activates them (gut_trace_raw_model_fields)
        # (GUT_TRACER_RAW_PARAMETERS=0 in the environment: default off — the parity tests, whose oracle is fed torch's own activations
        #  and compares bit for bit, set it in tests/conftest.py and switch the raw path on where they test it)
        self.raw_parameters = os.environ.get("GUT_TRACER_RAW_PARAMETERS", "1") != "0"

    @property
    def timings(self):
        return self.tracer_wrapper.collect_times()

    def build_acc(self, gaussians, rebuild=True):
        pass  # no acceleration structure in 3DGUT (tracer.py:301-302)

    def render(self, gaussians, gpu_batch, train=False, frame_id=0):
        rays_o = gpu_batch.rays_ori
        rays_d = gpu_batch.rays_dir
        sensor, poses = Tracer.create_camera_parameters(gpu_batch)
        if (getattr(self, "split_features", True) and hasattr(self.tracer_wrapper, "trace_model_fields")
                and getattr(self.tracer_wrapper, "sph_degree", 3) == 3   # the two-tensor hand-over is laid out for [N,3] + [N,45]
                and hasattr(gaussians, "get_features_albedo") and hasattr(gaussians, "get_features_specular")):
            # the reference's model keeps the SH coefficients as two tensors and concatenates them for every render (model.py:74-75):
            # hand the two over as they are (the only difference from tracer.py:317-327)
            features = (gaussians.get_features_albedo().contiguous(), sensor, poses, gaussians.get_features_specular().contiguous())
            if getattr(self, "raw_parameters", True) and hasattr(self.tracer_wrapper, "trace_raw_model_fields") and _has_reference_activations(gaussians):
                # the model's own nn.Parameters, activated inside the library: no normalize / exp / sigmoid kernels, none of their
                # backward kernels (threedgrut/model/model.py:77-93; the result equals those calls up to the last bits of expf)
                pred_rgba, pred_dist, hits_count, mog_visibility = Tracer._Autograd.apply(
                    self.tracer_wrapper, frame_id, gaussians.n_active_features, rays_o.contiguous(), rays_d.contiguous(),
                    gaussians.positions.contiguous(), gaussians.rotation.contiguous(), gaussians.scale.contiguous(),
                    gaussians.density.contiguous(), *features, True)
                return self._outputs(gaussians, gpu_batch, rays_d, pred_rgba, pred_dist, hits_count, mog_visibility, train)
        else:
            features = (gaussians.get_features().contiguous(), sensor, poses)
        pred_rgba, pred_dist, hits_count, mog_visibility = Tracer._Autograd.apply(
            self.tracer_wrapper, frame_id, gaussians.n_active_features, rays_o.contiguous(), rays_d.contiguous(),
            gaussians.positions.contiguous(), gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous(),
            gaussians.get_density().contiguous(), *features)
        return self._outputs(gaussians, gpu_batch, rays_d, pred_rgba, pred_dist, hits_count, mog_visibility, train)

    def _outputs(self, gaussians, gpu_batch, rays_d, pred_rgba, pred_dist, hits_count, mog_visibility, train):
        """The seven-key dictionary of tracer.py:329-351."""
        pred_rgb = pred_rgba[..., :3].unsqueeze(0).contiguous()
        pred_opacity = pred_rgba[..., 3:].unsqueeze(0).contiguous()
        pred_dist = pred_dist.unsqueeze(0).contiguous()
        hits_count = hits_count.unsqueeze(0).contiguous()
        pred_rgb, pred_opacity = gaussians.background(gpu_batch.T_to_world.contiguous(), rays_d, pred_rgb, pred_opacity, train)
        timings = self.tracer_wrapper.collect_times()
        return {
            "pred_rgb": pred_rgb,
            "pred_opacity": pred_opacity,
            "pred_dist": pred_dist,
            "pred_normals": torch.nn.functional.normalize(torch.ones_like(pred_rgb), dim=3),
            "hits_count": hits_count,
            "frame_time_ms": timings["forward_render"] if "forward_render" in timings else 0.0,
            "mog_visibility": mog_visibility,
        }

    @staticmethod
    def create_camera_parameters(gpu_batch):
        """Batch -> (CameraModelParameters, SensorPose3D); tracer.py:361-431."""
        pose = gpu_batch.T_to_world.squeeze()
        assert pose.ndim == 2
        poses = sensor_pose_from_c2w(pose.detach().cpu().numpy())
        if (K := getattr(gpu_batch, "intrinsics", None)) is not None:
            fx, fy, cx, cy = (float(K[0]), float(K[1]), float(K[2]), float(K[3]))
            w, h = int(2 * cx), int(2 * cy)
            # focal -> fov -> focal round trip in Python floats, exactly as tracer.py:386-403
            fov_x, fov_y = 2 * math.atan(w / (2 * fx)), 2 * math.atan(h / (2 * fy))
            return fromOpenCVPinholeCameraModelParameters(
                resolution=np.array([w, h], dtype=np.uint64), shutter_type=ShutterType.GLOBAL,
                principal_point=np.array([w, h], dtype=np.float32) / 2,
                focal_length=np.array([w / (2.0 * math.tan(fov_x * 0.5)), h / (2.0 * math.tan(fov_y * 0.5))], dtype=np.float32),
                radial_coeffs=np.zeros((6,), dtype=np.float32), tangential_coeffs=np.zeros((2,), dtype=np.float32),
                thin_prism_coeffs=np.zeros((4,), dtype=np.float32)), poses
        if (K := getattr(gpu_batch, "intrinsics_OpenCVPinholeCameraModelParameters", None)) is not None:
            return fromOpenCVPinholeCameraModelParameters(
                resolution=K["resolution"], shutter_type=plugin_shutter_type(K["shutter_type"]),
                principal_point=K["principal_point"], focal_length=K["focal_length"], radial_coeffs=K["radial_coeffs"],
                tangential_coeffs=K["tangential_coeffs"], thin_prism_coeffs=K["thin_prism_coeffs"]), poses
        if (K := getattr(gpu_batch, "intrinsics_OpenCVFisheyeCameraModelParameters", None)) is not None:
            return fromOpenCVFisheyeCameraModelParameters(
                resolution=K["resolution"], shutter_type=plugin_shutter_type(K["shutter_type"]),
                principal_point=K["principal_point"], focal_length=K["focal_length"], radial_coeffs=K["radial_coeffs"],
                max_angle=K["max_angle"]), poses
        raise ValueError("Camera intrinsics unavailable or unsupported")
