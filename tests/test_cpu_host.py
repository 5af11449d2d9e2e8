"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/gut_hip.h declares,
config mapping, camera-parameter construction, and the product path refuses to run without a GPU (no CPU
fallback).  No compute calls are made here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest
import torch

from tests.common import cams, make_view

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
capi = importlib.import_module("3dgrut_amd._capi")
tracer_mod = importlib.import_module("3dgrut_amd.tracer")
gut = importlib.import_module("3dgrut_amd")


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "gut_hip.h")).read()
    declared = set(re.findall(r"\b(gut_[a-z0-9_]+)\s*\(", header))
    declared -= {"gut_context"}
    assert len(declared) >= 15
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libgut_hip.so does not export {sym}"
    for sym in capi.EXPORTS:
        assert sym in declared, f"{sym} bound in _capi.py but not declared in include/gut_hip.h"
    assert lib.gut_abi_version() == capi.GUT_ABI_VERSION


def test_struct_layouts_match_the_header():
    # sizes implied by include/gut_hip.h (natural alignment): GutCamera 2*4+2*8+6*4+2*4+4*4+4+2*28+(pad 4)+2*8
    assert C.sizeof(capi.GutCamera) == 152
    assert C.sizeof(capi.GutConfig) == 12 * 4 + 8 * 4
    assert C.sizeof(capi.GutStats) == 7 * 8 + 8 + 16


def test_ctypes_mirrors_agree_with_the_compiled_header(tmp_path):
    """The ctypes mirrors of _capi.py against include/gut_hip.h as a C compiler sees it: struct sizes, the ABI version and the
    array lengths a caller sizes its buffers by (round 2 grew GutStats and GUT_NUM_KERNEL_TIMERS without bumping the version)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "gut_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %d %d %d\\n", sizeof(GutCamera), '
                   'sizeof(GutConfig), sizeof(GutStats), sizeof(GutLazyMoments), GUT_ABI_VERSION, GUT_NUM_KERNEL_TIMERS, '
                   'GUT_GRADIENT_RECORD_FLOATS); return 0; }\n')
    exe = tmp_path / "sizes"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(capi.GutCamera), C.sizeof(capi.GutConfig), C.sizeof(capi.GutStats), C.sizeof(capi.GutLazyMoments),
                   capi.GUT_ABI_VERSION, capi.GUT_NUM_KERNEL_TIMERS, capi.GRADIENT_RECORD_FLOATS], got
    assert len(capi.KERNEL_TIMER_NAMES) == capi.GUT_NUM_KERNEL_TIMERS


def test_default_config_and_conf_mapping():
    cfg = tracer_mod.config_from_conf(None)
    assert cfg.k_buffer_size == 0 and cfg.particle_kernel_degree == 2 and cfg.particle_radiance_sph_degree == 3
    assert abs(cfg.particle_kernel_min_alpha - 1 / 255) < 1e-9 and abs(cfg.min_transmittance - 1e-4) < 1e-9
    conf = {"render": {"enable_kernel_timings": True, "min_transmittance": 0.001,
                       "splat": {"k_buffer_size": 16, "ut_alpha": 1.5, "tile_based_culling": False}}}
    cfg = tracer_mod.config_from_conf(conf)
    assert cfg.enable_kernel_timings == 1 and cfg.k_buffer_size == 16 and abs(cfg.ut_alpha - 1.5) < 1e-7
    assert cfg.tile_based_culling == 0 and abs(cfg.min_transmittance - 0.001) < 1e-9

    class NS:  # attribute-style conf (OmegaConf-like)
        pass
    c = NS(); c.render = NS(); c.render.splat = NS(); c.render.splat.global_z_order = False
    assert tracer_mod.config_from_conf(c).global_z_order == 0


def test_unsupported_variant_is_rejected_without_gpu_work():
    lib = capi.load()
    cfg = tracer_mod.config_from_conf({"render": {"particle_kernel_degree": 7}})   # not one of the reference's kernels (0,1,2,3,4,5,8)
    h = C.c_void_p()
    rc = lib.gut_create(C.byref(cfg), 0, C.byref(h))
    assert rc != 0 and b"particle_kernel_degree" in lib.gut_last_error()


def test_camera_parameter_construction():
    view = make_view("pinhole_list", 128, 96, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=100.0, fy=110.0)
    batch = gut.Batch(rays_ori=torch.as_tensor(view["ro"]), rays_dir=torch.as_tensor(view["rd"]),
                      T_to_world=torch.as_tensor(view["c2w"])[None], **view["intrinsics_kw"])
    sensor, poses = gut.Tracer.create_camera_parameters(batch)
    assert sensor.cam.model == capi.CAMERA_PINHOLE and sensor.cam.shutter == 4
    # [fx,fy,cx,cy] path: focal survives the focal->fov->focal round trip of tracer.py:386-403
    assert abs(sensor.cam.focal_length[0] - 100.0) < 1e-3 and abs(sensor.cam.focal_length[1] - 110.0) < 1e-3
    assert list(sensor.cam.principal_point) == [64.0, 48.0]
    assert poses.timestamps_us == [0, 1] and len(poses.T_world_sensors[0]) == 7
    fe = make_view("fisheye", 120, 80, cams.look_at_c2w((0, 0, -2), (0, 0, 0)))
    b2 = gut.Batch(rays_ori=torch.as_tensor(fe["ro"]), rays_dir=torch.as_tensor(fe["rd"]),
                   T_to_world=torch.as_tensor(fe["c2w"])[None], **fe["intrinsics_kw"])
    s2, _ = gut.Tracer.create_camera_parameters(b2)
    assert s2.cam.model == capi.CAMERA_FISHEYE and s2.cam.max_angle > 0
    with pytest.raises(ValueError):
        gut.Tracer.create_camera_parameters(gut.Batch(rays_ori=batch.rays_ori, rays_dir=batch.rays_dir, T_to_world=batch.T_to_world))


def test_batch_shape_checks():
    with pytest.raises(AssertionError):
        gut.Batch(rays_ori=torch.zeros(2, 4, 4, 3), rays_dir=torch.zeros(1, 4, 4, 3), T_to_world=torch.eye(4)[None])
    with pytest.raises(AssertionError):
        gut.Batch(rays_ori=torch.zeros(1, 4, 4, 3), rays_dir=torch.zeros(1, 4, 4, 3), T_to_world=torch.eye(4)[None], intrinsics=[1, 2, 3])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU: it never routes through the oracle or any CPU code."""
    raster_cls = gut.SplatRaster
    with pytest.raises(Exception):
        raster_cls({"render": {}}, device_index=0)
    src = "".join(open(os.path.join(ROOT, "3dgrut_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "3dgrut_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src and "oracle." not in src.replace("oracle.py", "")


def test_fisheye_rays_are_unit_and_match_equidistant_model():
    ro, rd = cams.fisheye_rays(64, 48, 30.0, 30.0)
    assert np.abs(np.linalg.norm(rd, axis=-1) - 1).max() < 1e-6 and np.abs(ro).max() == 0
    # theta = |(p - pp)/f| for zero radial coefficients (camera_models.py:201-235)
    th = np.arccos(rd[0, 10, 50, 2]); px = np.array([50.5 - 32, 10.5 - 24]) / 30.0
    assert abs(th - np.linalg.norm(px)) < 1e-5
