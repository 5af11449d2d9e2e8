"""Dev probe (GPU): do the compositing kernels' two speeds switch inside ONE undisturbed process?
400 train steps of the bench scene, the library's K6 / K7 / side-stream / late-pass timers averaged over windows of ten steps."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes"); cams = importlib.import_module("3dgrut_amd.cameras")
native = importlib.import_module("3dgrut_amd.native")
dev = torch.device("cuda", 0)
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
scene = getattr(scenes, fn)(**kw)
model = native.NativeGaussianModel(scene, device=dev, spatial_order=True)
st = native.NativeTrainStep(model, gut.Tracer({"render": {"enable_kernel_timings": True}}), scene_extent=extent)
bench.synthetic_optimizer_state(st)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev)
ro_t, rd_t, gt = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev), torch.rand((1, H, W, 3), device=dev)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
def batch(i):
    return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=torch.as_tensor(c2ws[i % 8])[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K)
for i in range(2): st.step(batch(i))
st.tune_placement()
for i in range(2, 14): st.step(batch(i))
st.raster.kernel_times_mean()
line = []
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for w in range(windows):
    for i in range(10): st.step(batch(i))
    t, _ = st.raster.kernel_times_mean()
    line.append(f"{t['render']:.3f}/{t['render_bwd']:.3f}/{t['optimizer_early_2']:.2f}/{t['optimizer']:.3f}")
    if len(line) == 5:
        print("  ".join(line), flush=True); line = []
