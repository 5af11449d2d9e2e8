"""Dev probe: is the compositors' 5 % wander the CLOCK?  Runs the bench's train step for a few hundred steps on
a diagnostic build of the library whose backward compositor stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at the start
and end of every workgroup (gut_render.hip, GUT_CLOCK_STAMPS), and prints, per window of steps, K7's duration next to the clock the
chip held during the LAST launch of the window (median over its workgroups of delta cycles / delta real time x 100 MHz) and the
cycles its median workgroup took.  Duration x clock ~ constant => the chip slowed its clock (power management); duration moving at a
constant clock => the waves were starved of issue slots (co-runner).

    python tools/clock_probe.py build          (build container: hipcc cross-compiles the stamped library into tools/bin/)
    GUT_HIP_LIBRARY=tools/bin/libgut_hip_stamps.so python tools/clock_probe.py [windows] [steps_per_window] [workload] [--no-overlap]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "bin", "libgut_hip_stamps.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    b = importlib.import_module("3dgrut_amd.build")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    print(b.build_diagnostic(OUT, ["GUT_CLOCK_STAMPS"], verbose=False))
    raise SystemExit(0)

import numpy as np
import torch
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes"); cams = importlib.import_module("3dgrut_amd.cameras")
native = importlib.import_module("3dgrut_amd.native"); capi = importlib.import_module("3dgrut_amd._capi")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
windows = int(args[0]) if len(args) > 0 else 30
per = int(args[1]) if len(args) > 1 else 10
workload = args[2] if len(args) > 2 else "bicycle_like_6M_1237x822"
overlap = None if "--no-overlap" not in sys.argv else False
lib = capi.load()
assert hasattr(lib, "gut_debug_clock_stamps"), "not the stamped library: set GUT_HIP_LIBRARY=tools/bin/libgut_hip_stamps.so"
dev = torch.device("cuda", 0)
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS[workload]
scene = getattr(scenes, fn)(**kw)
model = native.NativeGaussianModel(scene, device=dev, spatial_order=True)
st = native.NativeTrainStep(model, gut.Tracer({"render": {"enable_kernel_timings": True}}), scene_extent=extent, overlap_optimizer=overlap)
bench.synthetic_optimizer_state(st)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev)
ro_t, rd_t, gt = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev), torch.rand((1, H, W, 3), device=dev)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
def batch(i):
    return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=torch.as_tensor(c2ws[i % 8])[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K)
for i in range(14): st.step(batch(i))
st.raster.kernel_times_mean()
stamps = np.zeros((8192, 4), np.uint64)
print(f"# {workload}, overlap_optimizer={st.overlap_optimizer}; per window of {per} steps: K6 ms, K7 ms (library timers, mean), clock GHz during the window's last K7 "
      f"(median / p10 / p90 over workgroups), median workgroup cycles, K7 ms x GHz")
for w in range(windows):
    for i in range(per): st.step(batch(w * per + i))
    t, _ = st.raster.kernel_times_mean()
    torch.cuda.synchronize()
    assert lib.gut_debug_clock_stamps(stamps.ctypes.data_as(C.c_void_p)) == 0
    ok = (stamps[:, 3] > stamps[:, 1]) & (stamps[:, 2] > stamps[:, 0])
    cyc = (stamps[ok, 2] - stamps[ok, 0]).astype(np.float64); rt = (stamps[ok, 3] - stamps[ok, 1]).astype(np.float64)
    long_ = rt > np.quantile(rt, 0.5)          # the longer half: short workgroups quantise badly against the 10 ns counter
    ghz = cyc[long_] / rt[long_] * 0.1
    # workgroups resident at once, averaged over the launch: sum of the workgroups' lifetimes / span of the launch (100 MHz ticks)
    span = float(stamps[ok, 3].max() - stamps[ok, 1].min())
    resident = rt.sum() / max(span, 1.0) / 256.0
    if hasattr(lib, "gut_debug_k6_phases"):
        ph = np.zeros((8192, 4), np.uint64)
        assert lib.gut_debug_k6_phases(ph.ctypes.data_as(C.c_void_p)) == 0
        ph = ph[ph[:, 0] > 0].astype(np.float64)
        tot = ph[:, 0].sum()
        k6 = f"  K6 workgroup time: select {ph[:, 1].sum() / tot:.2f} stage {ph[:, 2].sum() / tot:.2f} walk {ph[:, 3].sum() / tot:.2f} (mean wg cycles {ph[:, 0].mean():.0f})"
    else:
        k6 = ""
    if hasattr(lib, "gut_debug_k7_phases"):
        ph = np.zeros((8192, 4), np.uint64)
        assert lib.gut_debug_k7_phases(ph.ctypes.data_as(C.c_void_p)) == 0
        ph = ph[ph[:, 3] > 0].astype(np.float64)
        tot = ph[:, 3].sum()
        k6 += f"  K7 workgroup time: stage {ph[:, 0].sum() / tot:.2f} walk {ph[:, 1].sum() / tot:.2f} epilogue+flush {ph[:, 2].sum() / tot:.2f}"
    print(f"{w:3d}  K6 {t['render']:.3f}  K7 {t['render_bwd']:.3f}  clock {np.median(ghz):.3f} / {np.quantile(ghz, 0.1):.3f} / {np.quantile(ghz, 0.9):.3f} GHz  "
          f"wg cycles median {np.median(cyc):.0f} mean {cyc.mean():.0f}  K7 x GHz {t['render_bwd'] * np.median(ghz):.3f}  resident per CU {resident:.2f}  "
          f"stamped span {span * 1e-5:.3f} ms{k6}", flush=True)
